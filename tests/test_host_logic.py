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
    assert lib.trx_version() == 240
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


def test_more_argument_validation_without_gpu():
    """Workspace-size, optimiser-kind and extension entry points: every error path returns its status before any launch."""
    from torchregister_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    v = _lib.Volumes()
    v.moving, v.target, v.ndim, v.B, v.D, v.H, v.W = 16, 16, 3, 1, 8, 8, 8
    need = lib.trx_affine_workspace_bytes(ctypes.byref(v))
    lc, oc, st = _lib.LossCfg(), _lib.OptCfg(), _lib.AffineState()
    for f in ("param", "theta", "best_theta", "best_loss", "best_idx", "step"):
        setattr(st, f, 16)
    oc.kind = 7                                                           # neither SGD nor Adam
    assert lib.trx_affine_step(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), P(16), need, None) == -1
    oc.kind = 1                                                           # Adam without moment buffers
    assert lib.trx_affine_step(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), P(16), need, None) == -1
    oc.kind = 0
    assert lib.trx_affine_step(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), P(16), need - 1, None) == -3
    assert lib.trx_affine_run(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), -1, P(16), need, None) == -1
    st.losses, st.losses_capacity = 16, 5                                 # more iterations than the loss curve holds
    assert lib.trx_affine_run(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(st), 6, P(16), need, None) == -5
    assert lib.trx_affine_accumulate(ctypes.byref(v), P(16), P(16), 0, None) == -3
    assert lib.trx_affine_loss(ctypes.byref(v), ctypes.byref(lc), P(16), None, P(16), need, None) == -1
    # a volume of 2^31 voxels is refused (32-bit voxel indices inside one volume)
    big = _lib.Volumes()
    big.moving, big.target, big.ndim, big.B, big.D, big.H, big.W = 16, 16, 3, 1, 2048, 1024, 1024
    assert lib.trx_affine_workspace_bytes(ctypes.byref(big)) == 0
    assert lib.trx_flow_workspace_bytes(ctypes.byref(big)) == 0
    # flow entry points
    fs = _lib.FlowState()
    assert lib.trx_flow_workspace_bytes(ctypes.byref(v)) > 0
    assert lib.trx_flow_step(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(fs), P(16), 1 << 30, None) == -1   # null flow
    assert lib.trx_flow_warp(ctypes.byref(v), None, 1, P(16), None) == -1
    fs.flow, fs.losses, fs.losses_capacity = 16, 16, 3
    assert lib.trx_flow_run(ctypes.byref(v), ctypes.byref(lc), ctypes.byref(oc), ctypes.byref(fs), 4, P(16), 1 << 30, None) == -5
    # local-window NCC extension
    assert lib.trx_lncc_workspace_bytes(4, 1, 8, 8, 8) == 0
    assert lib.trx_lncc_workspace_bytes(2, 1, 3, 8, 8) == 0               # 2-D needs D == 1
    n = lib.trx_lncc_workspace_bytes(3, 2, 8, 8, 8)
    assert n >= 3 * 2 * 512 * 4                                            # three intermediate fields of fp32 per voxel
    args = (P(16), P(16), 3, 2, 8, 8, 8)
    assert lib.trx_lncc_loss_grad(*args, 9, 1.0, 1e-5, P(16), P(16), P(16), n - 1, None) == -3
    assert lib.trx_lncc_loss_grad(*args, 4, 1.0, 1e-5, P(16), P(16), P(16), n, None) == -1     # even window
    assert lib.trx_lncc_loss_grad(*args, 9, 1.0, 1e-5, None, None, P(16), n, None) == -1        # nothing to compute
    assert lib.trx_lncc_loss_grad(P(16), P(16), 5, 2, 8, 8, 8, 9, 1.0, 1e-5, P(16), P(16), P(16), n, None) == -2
    # Parzen-window PDF kernels
    assert lib.trx_kde_workspace_bytes(1, 1000, 2048) == 0                # more than 1024 bins
    kn = lib.trx_kde_workspace_bytes(2, 10000, 256)
    assert kn > 0
    assert lib.trx_kde_pdf(P(16), P(16), 2, 10000, 256, 3.0, P(16), P(16), kn - 1, None) == -3
    assert lib.trx_kde_pdf(P(16), P(16), 2, 10000, 256, 0.0, P(16), P(16), kn, None) == -1     # bandwidth must be positive
    assert lib.trx_kde_pdf_backward(P(16), P(16), None, 2, 10000, 256, 3.0, P(16), None) == -1
    for code, word in ((-1, b"arg"), (-2, b"dim"), (-4, b"HIP"), (-5, b"loss-curve")):
        assert word.lower() in lib.trx_status_string(code).lower()


def test_ctypes_structs_match_the_header_layout():
    """The ctypes mirrors of the C structs are checked against sizes computed by the C compiler from include/trx.h itself."""
    import subprocess
    import tempfile
    from torchregister_amd import _lib
    src = '#include <stdio.h>\n#include "trx.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(trx_volumes), sizeof(trx_loss_cfg),' \
          ' sizeof(trx_opt_cfg), sizeof(trx_affine_state), sizeof(trx_flow_state), offsetof(trx_volumes, flags), offsetof(trx_flow_state, stopped));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    want = [ctypes.sizeof(_lib.Volumes), ctypes.sizeof(_lib.LossCfg), ctypes.sizeof(_lib.OptCfg), ctypes.sizeof(_lib.AffineState),
            ctypes.sizeof(_lib.FlowState), _lib.Volumes.flags.offset, _lib.FlowState.stopped.offset]
    assert got == want


def test_library_reads_no_environment_variable():
    """include/trx.h promises no global state: the product library must not import getenv (dev knobs are compiled out)."""
    import subprocess
    from torchregister_amd import _lib
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "getenv" not in syms


def test_flow_register_host_checks():
    import torchregister_amd as tr
    from torchregister_amd.warpings import smooth_regulariser
    with pytest.raises(IndexError):                                       # the reference indexes weights[i]
        tr.flow_register((8, 8), criterions=[nn.MSELoss(), tr.NCCLoss()], weights=[1.0], flow_model="direct")
    fl = torch.randn(2, 3, 5, 6, 7, dtype=torch.float64)
    want = 0.0
    for b in range(2):
        for d in range(3):
            df = fl[b].diff(dim=1 + d)
            want = want + 2.5 / 3 * (df * df).mean()
    assert abs(smooth_regulariser(fl, 2.5).item() - want.item()) < 1e-12


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
    # mixed lists: the terms with a fused form go to one F1 launch, the others stay with torch (warpings._generic_loop)
    spec, rest = w.split_fusable(c, [0.3, 0.5, 0.2])                  # the default criterion: MSE + NCC fused, NMI generic
    assert (spec.w_mse, spec.w_ncc, spec.ncc_alpha) == (0.3, 0.5, 100.0) and len(rest) == 1 and type(rest[0][0]) is tr.NMILoss and rest[0][1] == 0.2
    l1 = nn.L1Loss()
    spec, rest = w.split_fusable([l1, tr.SSDLoss(alpha=2)], [1.0, 0.25])
    assert (spec.w_ssd, spec.ssd_alpha, spec.w_mse, spec.w_ncc) == (0.25, 2.0, 0.0, 0.0) and rest == [(l1, 1.0)]
    spec, rest = w.split_fusable([l1], [1.0])
    assert spec is None and rest == [(l1, 1.0)]
    two = [tr.NCCLoss(alpha=10), tr.NCCLoss(alpha=100), l1]          # two NCC terms with different alpha have no single fused form
    spec, rest = w.split_fusable(two, [1.0, 1.0, 1.0])
    assert spec is None and [c for c, _ in rest] == two


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


def test_compose_theta_is_the_chain_of_the_two_grids():
    """compose_theta(first, second): sampling coordinates of the one-warp pipeline = first applied to the coordinates
    second produces, checked with F.affine_grid itself (CPU, fp64), 2-D and 3-D, batched and broadcast."""
    import torch.nn.functional as F
    import torchregister_amd as tr
    g = torch.Generator().manual_seed(3)
    for nd, shape in ((2, (1, 1, 7, 9)), (3, (1, 1, 5, 6, 7))):
        a = (torch.eye(nd, nd + 1) + 0.2 * torch.randn(nd, nd + 1, generator=g)).double()[None]
        b = (torch.eye(nd, nd + 1) + 0.2 * torch.randn(nd, nd + 1, generator=g)).double()[None]
        c = tr.compose_theta(a, b)
        assert c.shape == (1, nd, nd + 1) and c.dtype == torch.float64
        grid_b = F.affine_grid(b, shape, align_corners=False)                 # where the second warp reads the intermediate image
        ones = torch.ones(*grid_b.shape[:-1], 1, dtype=torch.float64)
        chained = torch.cat([grid_b, ones], dim=-1) @ a[0].T                 # ... mapped through the first warp's matrix
        assert torch.allclose(F.affine_grid(c, shape, align_corners=False), chained, atol=1e-12)
    a2 = torch.eye(3, 4)                                                      # [nd, nd+1] input, batch broadcast, dtype kept
    b2 = torch.eye(3, 4)[None].repeat(4, 1, 1)
    b2[:, 0, 3] = torch.arange(4.0)
    out = tr.compose_theta(a2, b2)
    assert out.shape == (4, 3, 4) and out.dtype == torch.float32 and torch.equal(out, b2)
    with pytest.raises(ValueError):
        tr.compose_theta(torch.eye(2, 3), torch.eye(3, 4))


def test_exact_footprint_plan_model_never_misses_a_cell():
    """The position-independent plan of the exact-footprint kernel (csrc/affine_eft.h: ef_row_window / ef_dims), in its numpy restatement
    (tools/eft_plan_check.py): for rotated / zoomed / sheared maps the per-row x-windows cover every cell the 2 x 2 x 2 neighbourhoods of a
    16^3 tile touch, whatever the fractional origin, and the granule count of the two bench poses fits one LDS buffer (2304 granules).  The
    kernel itself is checked against the oracle on the GPU (tests/test_gpu_eft.py); this keeps the geometry argument honest on the CPU."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("eft_plan_check", os.path.join(ROOT, "tools", "eft_plan_check.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    rng = np.random.default_rng(3)
    poses = [m.rot(.5, .4, .3) @ np.diag([1.05, .95, 1.02]), m.rot(0.4963, 0.7682, 0.0885), np.eye(3)]
    for _ in range(3):
        a = rng.random(3) * 1.2 * rng.choice([-1, 1], 3)
        poses.append(m.rot(*a) @ np.diag(0.9 + 0.2 * rng.random(3)) + 0.05 * (rng.random((3, 3)) - 0.5))
    for i, A in enumerate(poses):
        rows, G, dims = m.plan(A)
        missed, exact = m.check(A, rows, n=4, seed=i)
        assert missed == 0
        assert G >= exact
        if i < 2:
            assert G <= 2304 and dims[0] <= 32 and dims[1] <= 32


def test_affine_workspace_holds_the_notes_between_kernels():
    """The affine workspace ends in int rows_used[B + 11]: per pair the rows / body of the last step, then what the kernels of ONE step leave each
    other - pairs left by the z-streaming kernel, its pair mask (2 ints), the exact-footprint kernel's work tickets (8 ints, zeroed by the
    kernel in front on every launch: callers never initialise the workspace).  The size the library asks for must hold them for every batch."""
    from torchregister_amd import _lib
    lib = _lib.load()
    for B in (1, 8, 53, 54, 61, 62, 64):
        v = _lib.Volumes()
        v.moving, v.target, v.ndim, v.B, v.D, v.H, v.W = 16, 16, 3, B, 64, 64, 64
        need, off = lib.trx_affine_workspace_bytes(ctypes.byref(v)), lib.trx_affine_workspace_rows_offset(ctypes.byref(v))
        assert off > 0 and need >= off + (B + 11) * 4, (B, need, off)


def test_near_identity_helper_without_gpu():
    """trx_affine_near_identity is host arithmetic (the z-streaming kernel's window test on host thetas): the identity and a small affine
    perturbation of a chip-filling 3-D batch are inside, a rotation is not; 2-D, small launches and bad arguments answer 0."""
    import numpy as np
    from torchregister_amd import _lib
    lib = _lib.load()
    v = _lib.Volumes()
    v.moving, v.target, v.ndim, v.B, v.D, v.H, v.W = 16, 16, 3, 8, 256, 256, 256
    def ask(th):
        a = np.ascontiguousarray(np.stack([np.asarray(th, dtype=np.float32).reshape(12)] * v.B))
        return lib.trx_affine_near_identity(ctypes.byref(v), a.ctypes.data_as(ctypes.c_void_p))
    eye = np.eye(3, 4)
    assert ask(eye) == 1
    assert ask(eye + 0.01 * np.sin(np.arange(12.0)).reshape(3, 4)) == 1
    c, s_ = np.cos(0.5), np.sin(0.5)
    assert ask([[c, -s_, 0, 0], [s_, c, 0, 0], [0, 0, 1, 0]]) == 0
    assert ask(np.full((3, 4), np.nan)) == 0
    assert lib.trx_affine_near_identity(ctypes.byref(v), None) == 0
    v.B = 1; v.D = v.H = v.W = 64
    assert ask(eye) == 0          # a launch the z-streaming kernel is not offered
    v.ndim, v.D = 2, 1
    assert ask(eye) == 0
