#!/usr/bin/env python3
"""3-D U-Net flow mode: steady-state seconds per iteration under different torch / MIOpen settings.
   python tools/prof_unet.py [size] [TRX_MIOPEN_BENCHMARK 0|1]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import TorchRegister as tr
import phantoms as ph
S = int(sys.argv[1]) if len(sys.argv) > 1 else 156
os.environ["TRX_MIOPEN_BENCHMARK"] = sys.argv[2] if len(sys.argv) > 2 else "1"
shape = (S, S, S)
tgt = ph.blobs(shape, 1000).cuda(); mov = ph.blobs(shape, 1001).cuda()
for iters in (2, 5):
    torch.manual_seed(0)
    reg = tr.Register("flow", device="cuda", criterion=[tr.NCCLoss()], weight=[1.0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg.optim(mov, tgt, lr=1e-4, max_epochs=iters, n=32)
    torch.cuda.synchronize()
    print(f"{shape} TRX_MIOPEN_BENCHMARK={os.environ['TRX_MIOPEN_BENCHMARK']}: {iters} iterations in {time.perf_counter() - t0:.2f} s", flush=True)
