import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle, phantoms as ph
import torchregister_amd._engine as eng
from torchregister_amd import _lib
import test_gpu_one_kernel as T
mov, tgt = T.batch(300, rough=False)
mats = [T.rot(0.5, 0.4, 0.3), T.rot(0, 0, 0.6) * 1.05, np.diag([1.3, 0.8, 1.1]), T.rot(0.7, 0.8, 0.6), np.diag([-1.0, 1.0, 1.0]), T.rot(0.2, 0.0, 0.0)]
ths = []
for i in range(T.B):
    if i % 3 == 2:
        m = mats[(i // 3) % len(mats)]
        ths.append(np.concatenate([m, [[0.011], [-0.017], [0.013]]], axis=1) + 1e-3 * np.sin(np.arange(12.0).reshape(3, 4) + i))
    else:
        ths.append(T.near_identity(i, 5e-3))
th = torch.tensor(np.stack(ths), dtype=torch.float32)
kw = dict(w_ncc=1.0)
res = {}
for name, ok, fl in (("one", True, 0), ("three", False, 0), ("three-noeft", False, _lib.FLAG_NO_EFT), ("three-noeft-nord", False, _lib.FLAG_NO_EFT | _lib.FLAG_NO_ROT_DEEP_TILE)):
    s = eng.AffineSolver(mov, tgt, mode="affine", loss=eng.LossSpec(**kw), lr=0.0, init=th, capacity=1, one_kernel=ok, flags=fl)
    s.run(1); torch.cuda.synchronize()
    res[name] = (s.losses[:, 0].cpu(), s.grad[:, :12].cpu(), s.bodies())
for i in (2, 5, 8, 11, 14):
    total, _, dth, _ = oracle.c_affine_loss_grad(mov[i, 0].double().cpu().numpy(), tgt[i, 0].double().cpu().numpy(), th[i].double().numpy(), oracle.wts(**kw), oracle.base_tables(T.SHAPE, np.float64))
    gmax = np.max(np.abs(dth))
    print("pair", i, "oracle loss", total)
    for name, (l, g, bd) in res.items():
        print(f"   {name:18s} body {bd[i]:12s} loss err {abs(l[i].item() - total):.2e}  grad err / max {np.max(np.abs(g[i].numpy().reshape(3, 4) - dth)) / gmax:.2e}")
