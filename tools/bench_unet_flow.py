#!/usr/bin/env python3
"""mode='flow' as the reference runs it (the attention U-Net generates the flow): time per iteration and where it goes.
   python tools/bench_unet_flow.py            -> 2-D 160^2 and 3-D 156^3 (sizes the valid convolutions accept), n = 32
The convolutions run in MIOpen through torch; the warp + loss + backward wrt the flow are the fused HIP kernels (one autograd.Function)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import TorchRegister as tr
import phantoms as ph
dev = "cuda"
for shape in ((160, 160), (156, 156, 156)):
    try:
        tgt = ph.blobs(shape, 1000).to(dev); mov = ph.blobs(shape, 1001).to(dev)
        for iters in (3, 23):
            torch.manual_seed(0)
            reg = tr.Register("flow", device=dev, criterion=[tr.NCCLoss()], weight=[1.0])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reg.optim(mov, tgt, lr=1e-4, max_epochs=iters, n=32)
            torch.cuda.synchronize(); t = time.perf_counter() - t0
            print(f"{shape} U-Net flow, {iters} iterations: {t * 1e3:.1f} ms")
    except Exception as e:
        print(shape, "failed:", type(e).__name__, str(e)[:200])
