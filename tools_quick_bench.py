# scratch timing helper (first GPU contact); superseded by bench.py
import sys, time, torch
sys.path.insert(0, 'tests')
import phantoms as ph
from torchregister_amd import AffineSolver, LossSpec
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 256
shape = (S, S, S)
g = torch.Generator().manual_seed(0)
tgt = torch.rand((B, 1) + shape, generator=g).cuda()
mov = torch.rand((B, 1) + shape, generator=g).cuda()
s = AffineSolver(mov, tgt, mode='affine', loss=LossSpec(w_ncc=1.0), lr=1e-6, capacity=64)
s.run(3); torch.cuda.synchronize()
for iters in (10, 30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.step.zero_()
    e0.record(); s.run(iters); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    per = ms / iters / B * 1e3
    print(f"B={B} S={S} iters={iters}: {ms/iters:.3f} ms/iter, {per:.1f} us/pair-iter, {B*iters/ms*1e3:.0f} pair-it/s, algBW {8*S**3/per/1e6:.2f} TB/s")
print(s.losses[0, :5])
