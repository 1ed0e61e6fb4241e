import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torchregister_amd import _engine as eng
from bench import blobs_gpu
dev = torch.device("cuda")
shape = (256,) * 3
tgt = torch.cat([blobs_gpu(shape, 1000 + i, dev) for i in range(8)])
wrp = torch.cat([blobs_gpu(shape, 2000 + i, dev) for i in range(8)])
for _ in range(5):
    eng.local_ncc_loss_grad(tgt, wrp, 9)
torch.cuda.synchronize()
