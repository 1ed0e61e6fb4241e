"""The local-window NCC kernels alone, for profilers: python3 tools/lncc_prof.py [pairs] [window] - five loss + gradient evaluations of `pairs` x 256^3."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torchregister_amd import _engine as eng
from bench import blobs_gpu
dev = torch.device("cuda")
shape = (256,) * 3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = int(sys.argv[2]) if len(sys.argv) > 2 else 9
tgt = torch.cat([blobs_gpu(shape, 1000 + i, dev) for i in range(B)])
wrp = torch.cat([blobs_gpu(shape, 2000 + i, dev) for i in range(B)])
for _ in range(5):
    eng.local_ncc_loss_grad(tgt, wrp, w)
torch.cuda.synchronize()
