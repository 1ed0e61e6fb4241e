"""CPU-only checks of the host side: C-ABI surface, criterion mapping, torch-side helpers."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "trx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trx_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from torchregister_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/trx.h but not exported by libtrx.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.trx_version() == 100
    assert b"workspace" in lib.trx_status_string(-3)


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the GPU (no HIP call is made)."""
    from torchregister_amd import _lib
    lib = _lib.load()
    v = _lib.Volumes()
    assert lib.trx_affine_workspace_bytes(ctypes.byref(v)) == 0          # null moving
    v.moving, v.ndim, v.B, v.D, v.H, v.W = 16, 4, 1, 8, 8, 8
    assert lib.trx_affine_workspace_bytes(ctypes.byref(v)) == 0          # ndim 4
    v.ndim = 3
    assert lib.trx_affine_workspace_bytes(ctypes.byref(v)) > 0
    assert lib.trx_affine_warp(ctypes.byref(v), None, 1, None, None) == -1
    v.ndim, v.D = 2, 3
    assert lib.trx_affine_warp(ctypes.byref(v), ctypes.c_void_p(16), 1, ctypes.c_void_p(16), None) == -2
    lc, oc, st = _lib.LossCfg(), _lib.OptCfg(), _lib.AffineState()
    v.ndim, v.D, v.target = 3, 8, 16
    assert lib.trx_affine_step(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), ctypes.c_void_p(16), 1 << 30, None) == -1


def test_cpu_tensors_fail_loudly():
    import torchregister_amd as tr
    reg = tr.Register("affine")
    with pytest.raises(tr._lib.TrxError if hasattr(tr, "_lib") else Exception, match="no CPU fallback"):
        reg.optim(torch.rand(1, 1, 8, 8, 8), torch.rand(1, 1, 8, 8, 8), max_epochs=1)
    with pytest.raises(Exception, match="no CPU fallback"):
        tr.get_affine_warp(torch.eye(3, 4)[None], torch.rand(1, 1, 4, 4, 4))
    with pytest.raises(NotImplementedError):
        tr.affine_register(torch.rand(1, 1, 8, 8, 8), torch.rand(1, 1, 8, 8, 8))   # grad_edges defaults True (Q6)


def test_criterion_mapping_follows_reference_branches():
    from torchregister_amd import warpings as w
    import torchregister_amd as tr
    # user list is discarded (Q2) unless honor_criterion
    c, wt = w._resolve_criterions([tr.NCCLoss()], [1.0], False, "cpu")
    assert len(c) == 1 and type(c[0]) is nn.MSELoss and wt == [1.0]
    c, wt = w._resolve_criterions([tr.NCCLoss()], [2.0], True, "cpu")
    assert type(c[0]) is tr.NCCLoss and wt == [2.0]
    c, wt = w._resolve_criterions(None, [0.0, 1.0, 0.0], False, "cpu")
    assert [type(x) for x in c] == [nn.MSELoss, tr.NCCLoss, tr.NMILoss]
    s = w.loss_spec_from(c, wt)
    assert s is not None and (s.w_mse, s.w_ncc, s.ncc_alpha, s.w_ssd) == (0.0, 1.0, 100.0, 0.0)
    assert w.loss_spec_from(c, [0.33, 0.33, 0.33]) is None               # NMI with weight -> generic path
    s = w.loss_spec_from([nn.MSELoss(), tr.SSDLoss(alpha=2), tr.NCCLoss(alpha=10)], [0.5, 0.1, 0.2])
    assert (s.w_mse, s.w_ssd, s.ssd_alpha, s.w_ncc, s.ncc_alpha) == (0.5, 0.1, 2.0, 0.2, 10.0)
    assert w.loss_spec_from([nn.L1Loss()], [1.0]) is None
    assert w.loss_spec_from([nn.MSELoss(reduction="sum")], [1.0]) is None


def test_torch_side_helpers_match_golden(single_step):
    import torchregister_amd as tr
    import phantoms as ph
    g = single_step
    x = ph.vol((5, 6, 7), 0.37)
    assert abs(tr.NCCLoss()(x, x).item() - float(g["ncc_self"])) < 1e-5
    assert np.allclose(tr.norm(torch.tensor([1.0, 2.0, 4.0])).numpy(), g["norm124"])
    mov, tgt = ph.vol((5, 6, 7), 0.37, "sin"), ph.vol((5, 6, 7), 0.23, "cos")
    w = torch.from_numpy(g["A3/warped32"])
    assert abs(tr.NCCLoss()(tgt, w).item() - float(g["A3/ncc32"])) < 1e-4
    assert abs(tr.SSDLoss()(tgt, w).item() - float(g["A3/ssd32"])) < 1e-3
    th = tr.Theta()
    for name in ("theta3", "theta2"):
        out = th(torch.tensor(g[f"{name}/x"], dtype=torch.float32))
        assert np.allclose(out.numpy(), g[f"{name}/out32"], atol=1e-7)
    assert abs(tr.NMILoss()(ph.blobs((32, 32), 1), ph.blobs((32, 32), 2)).item() - float(g["nmi2d"])) <= 1e-4 * abs(float(g["nmi2d"]))
    torch.manual_seed(0)
    r = tr.Regressor(torch.zeros(1, 1, 4, 4, 4), "cpu")
    assert np.allclose(r.reg.detach().numpy(), [0.49625659, 0.76822180, 0.08847743, 0.13203049, 0.30742282, 0.63407868], atol=1e-7)
    assert r().shape == (1, 3, 4)
    a = torch.zeros(1, 2, 5, 7)
    b = torch.zeros(1, 2, 8, 8)
    p = tr.padNd(a, b)
    assert p.shape == b.shape
