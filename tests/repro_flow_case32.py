# development: the flow case the round-5 sweep found over its bar (tests/fuzz_flow_lncc.py 100 65, case 32), printed element by element against the fp64 / fp32 oracle.
# test infrastructure like tests/: uses the oracle as the checker.   python tests/repro_flow_case32.py
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle, phantoms as ph
import torchregister_amd._engine as eng
from fuzz_flow_lncc import smooth_nd
rng = np.random.default_rng(65)
for it in range(33):
    nd = 3 if rng.random() < 0.7 else 2
    shape = tuple(int(v) for v in rng.integers(3, 40 if nd == 3 else 90, nd))
    tgt = ph.blobs(shape, 300 + it) + 0.05 * smooth_nd(shape, 0.31)
    mov = ph.blobs(shape, 700 + it) + 0.1 * smooth_nd(shape, 0.23)
    amp = float(rng.choice([0.3, 1.5, 6.0]))
    flow = torch.tensor(amp * rng.standard_normal((1, nd) + shape), dtype=torch.float32) + 0.37
    kw = dict(w_ncc=float(rng.uniform(0, 1)), w_mse=float(rng.uniform(0, 1)))
    win = int(rng.choice([3, 5, 7, 9])); B = int(rng.integers(1, 3))
print(shape, amp, kw)
terms, dfl = eng.flow_loss_grad(mov.cuda(), tgt.cuda(), flow.cuda(), eng.LossSpec(**kw))
args = lambda dt: (mov[0, 0].numpy().astype(dt), tgt[0, 0].numpy().astype(dt), flow[0].numpy().astype(dt), oracle.wts(**kw))
t64, parts64, d64, _ = oracle.c_flow_loss_grad(*args(np.float64))
t32, parts32, d32, _ = oracle.c_flow_loss_grad(*args(np.float32))
print('loss', terms.cpu().numpy(), t64, t32, parts64)
g = dfl[0].cpu().numpy()
np.set_printoptions(precision=5, linewidth=200, suppress=False)
print('gmax', np.abs(d64).max())
print('hip - d64:\n', g - d64)
print('d32 - d64:\n', d32 - d64)
print('d64:\n', d64)
# positions
print('flow:\n', flow[0].numpy())
