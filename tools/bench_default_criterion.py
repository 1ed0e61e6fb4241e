#!/usr/bin/env python3
"""Default criterion (criterion=None: MSE + NCC + NMI, ref:torchregister.py / README usage) in 3-D: time per iteration of the
generic path (HIP warp + torch losses; the NMI's Parzen PDFs through trx_kde_pdf).  The reference cannot run this setting:
its PDF materialises an [8, 10^6, 256] fp32 tensor (8 GB) three times per evaluation."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import TorchRegister as tr
from bench import blobs_gpu, THETA_STAR
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
tgt = blobs_gpu((S,) * 3, 1000, dev)
mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None], tgt)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reg = tr.Register(mode="affine", device=dev)
    reg.optim(mov, tgt, lr=1e-3, max_epochs=n)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(3)                                   # first call: kernels compiled / loaded, allocator warmed
t10, t60 = min(run(10) for _ in range(3)), min(run(60) for _ in range(3))
print(f"{S}^3 default criterion (MSE + NCC + NMI): {t10 * 1e3:.1f} ms for 10 iterations, {t60 * 1e3:.1f} ms for 60 -> "
      f"{(t60 - t10) / 50 * 1e3:.3f} ms per iteration  (peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB)")
if len(sys.argv) > 2:                    # kernel / launch census of 10 iterations
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        run(10)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
