import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torchregister_amd._engine as eng, torchregister_amd._lib as lib
import phantoms as ph, oracle
from test_gpu_zstream import near_identity
shape = (16, 32, 64)
th = torch.tensor(near_identity(sum(shape), 3e-3, (0, 0, 0)), dtype=torch.float32)[None]
print('theta', th[0].numpy().tolist())
D, H, W = shape
zz = torch.arange(D, dtype=torch.float32)[:, None, None].expand(D, H, W)
for name, mov in (("zramp", (zz + 1.0).clone()), ("plane5", (zz == 5).float()), ("plane0", (zz == 0).float()), ("plane15", (zz == 15).float()), ("blobs", ph.blobs(shape, 42)[0, 0])):
    mov = mov.contiguous().view(1, 1, D, H, W); tgt = torch.zeros_like(mov)
    out = {}
    for nm, fl in (("up", lib.FLAG_ZSTREAM), ("down", lib.FLAG_ZSTREAM | lib.FLAG_WALK_DOWN), ("tile", lib.FLAG_NO_ZSTREAM)):
        s = eng.AffineSolver(mov.cuda(), tgt.cuda(), mode="affine", loss=eng.LossSpec(w_mse=1.0), lr=0.0, init=th, capacity=1, flags=fl)
        s.run(1); torch.cuda.synchronize()
        out[nm] = (s.losses[0, 0].item(), s.grad[0, :12].cpu().numpy().reshape(3, 4))
    print(name, 'loss', [out[k][0] for k in out])
    print('  up-tile  ', np.abs(out['up'][1] - out['tile'][1]).max(axis=1), ' down-tile', np.abs(out['down'][1] - out['tile'][1]).max(axis=1), ' scale', np.abs(out['tile'][1]).max())
