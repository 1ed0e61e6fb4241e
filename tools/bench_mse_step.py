import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torchregister_amd as tr
from bench import blobs_gpu, THETA_STAR
dev = torch.device("cuda"); S, B = 256, 8
tgt = torch.cat([blobs_gpu((S,)*3, 1000+b, dev) for b in range(B)])
mov = tr.get_affine_warp(torch.tensor(THETA_STAR, device=dev)[None].expand(B,3,4).contiguous(), tgt)
for name, spec in (("NCC", tr.LossSpec(w_ncc=1.0)), ("MSE", tr.LossSpec(w_mse=1.0)), ("MSE+SSD", tr.LossSpec(w_mse=0.5, w_ssd=0.1))):
    s = tr.AffineSolver(mov, tgt, mode="affine", loss=spec, optimizer="adam", lr=1e-4, capacity=600)
    s.run(150); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s.run(200); e1.record(); torch.cuda.synchronize()
    print(f"{name:8s} {e0.elapsed_time(e1)*1e3/200:7.1f} us per 8-pair step; loss {s.losses[0,0].item():.5f} -> {s.losses[0,349].item():.5f}")
