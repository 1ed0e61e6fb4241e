"""Pin the oracle (C restatement + torch composition) to the reference's golden outputs.

CPU-only.  Golden vectors come from tests/golden/make_golden.py (reference imported in the
build container).  Tolerances: fp64 instantiation vs the reference's fp64 run — 1e-9 relative;
fp32 — max(stated floor, 2x the reference's own fp32-vs-fp64 gap).
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import compose
import phantoms as ph
from conftest import bar

AFFINE_CASES = ["A3", "B2", "OOB3", "ROT3", "OOB2", "ID3", "ID2"]
FLOW_CASES = ["D3", "D2", "F3", "F2"]
LOSSES = {"ncc": dict(w_ncc=1.0), "mse": dict(w_mse=1.0), "ssd": dict(w_ssd=1.0)}


def _inputs(shape, dtype):
    mov = ph.vol(shape, 0.37, "sin")[0, 0].numpy().astype(dtype)
    tgt = ph.vol(shape, 0.23, "cos")[0, 0].numpy().astype(dtype)
    return mov, tgt


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


@pytest.mark.parametrize("case", AFFINE_CASES)
def test_c_oracle_affine_fp64(single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, tgt = _inputs(shape, np.float64)
    th = g[f"{case}/theta"]
    # base-coordinate tables built with ATen's own expression: at exactly-integer sample
    # coordinates (identity theta) the one-sided derivative taken depends on their last bit
    tabs = oracle.base_tables(shape, np.float64)
    w = oracle.c_affine_warp(mov, th, tabs)
    assert np.max(np.abs(w - g[f"{case}/warped64"][0, 0])) < 1e-12
    for lname, kw in LOSSES.items():
        total, _, dth, _ = oracle.c_affine_loss_grad(mov, tgt, th, oracle.wts(**kw), tabs)
        assert abs(total - float(g[f"{case}/{lname}64"])) <= 1e-9 * max(1.0, abs(total)), lname
        assert relerr(dth, g[f"{case}/d{lname}64"]) < 1e-9, lname


@pytest.mark.parametrize("case", AFFINE_CASES)
def test_c_oracle_affine_fp32(single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    mov, tgt = _inputs(shape, np.float32)
    th = g[f"{case}/theta"].astype(np.float32)
    tabs = oracle.base_tables(shape, np.float32)
    w = oracle.c_affine_warp(mov, th, tabs)
    g32, g64 = g[f"{case}/warped32"][0, 0], g[f"{case}/warped64"][0, 0]
    assert np.max(np.abs(w - g32)) <= bar(g32, g64, 2e-6)
    for lname, kw in LOSSES.items():
        total, _, dth, _ = oracle.c_affine_loss_grad(mov, tgt, th, oracle.wts(**kw), tabs)
        l32, l64 = float(g[f"{case}/{lname}32"]), float(g[f"{case}/{lname}64"])
        assert abs(total - l32) <= bar(l32, l64, 2e-5 * max(1.0, abs(l64))), lname
        d32, d64 = g[f"{case}/d{lname}32"], g[f"{case}/d{lname}64"]
        assert np.max(np.abs(dth - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64))), lname


@pytest.mark.parametrize("case", FLOW_CASES)
def test_c_oracle_flow(single_step, case):
    g = single_step
    shape = tuple(g[f"{case}/shape"])
    amp = float(g[f"{case}/amp"])
    for dt, tag in ((np.float64, "64"), (np.float32, "32")):
        mov, tgt = _inputs(shape, dt)
        fl = ph.flow_field(shape, amp)[0].numpy().astype(dt)
        w = oracle.c_flow_warp(mov, fl)
        g32, g64 = g[f"{case}/warped32"][0, 0], g[f"{case}/warped64"][0, 0]
        ref = g[f"{case}/warped{tag}"][0, 0]
        # the fp64 golden ran the reference's float32 identity grid promoted to double: exact
        tol = 1e-12 if tag == "64" else bar(g32, g64, 2e-6)
        assert np.max(np.abs(w - ref)) <= tol
        for lname in ("ncc", "mse"):
            total, _, dfl, _ = oracle.c_flow_loss_grad(mov, tgt, fl, oracle.wts(**LOSSES[lname]))
            l32, l64 = float(g[f"{case}/{lname}32"]), float(g[f"{case}/{lname}64"])
            d32, d64 = g[f"{case}/d{lname}32"][0], g[f"{case}/d{lname}64"][0]
            if tag == "64":
                assert abs(total - l64) <= 1e-9 * max(1.0, abs(l64))
                assert relerr(dfl, d64) < 1e-9
            else:
                assert abs(total - l32) <= bar(l32, l64, 2e-5 * max(1.0, abs(l64)))
                assert np.max(np.abs(dfl - d32)) <= bar(d32, d64, 1e-4 * np.max(np.abs(d64)))


def test_c_oracle_theta(single_step):
    g = single_step
    for name in ("theta3", "theta2"):
        x = g[f"{name}/x"]
        for dt, tag, tol in ((np.float64, "64", 1e-14), (np.float32, "32", 3e-7)):
            th = oracle.c_theta_fwd(x.astype(dt))
            assert np.max(np.abs(th.reshape(-1) - g[f"{name}/out{tag}"])) <= tol
            gv = np.arange(1, th.size + 1, dtype=dt)
            dx = oracle.c_theta_vjp(x.astype(dt), gv)
            assert np.max(np.abs(dx - g[f"{name}/jtv{tag}"])) <= tol * 50


def test_compose_matches_golden_single_step(single_step):
    """torch-op composition == reference bitwise in fp32 (same ATen kernels, same op order)."""
    g = single_step
    for case in AFFINE_CASES:
        shape = tuple(g[f"{case}/shape"])
        mov, tgt = ph.vol(shape, 0.37, "sin"), ph.vol(shape, 0.23, "cos")
        th = torch.tensor(g[f"{case}/theta"], dtype=torch.float32)[None].requires_grad_()
        w = compose.affine_warp(th, mov)
        assert np.array_equal(w.detach().numpy(), g[f"{case}/warped32"])
        e = compose.ncc_loss(tgt, w)
        e.backward()
        assert np.allclose(e.item(), g[f"{case}/ncc32"], rtol=1e-6)
        assert np.allclose(th.grad[0].numpy(), g[f"{case}/dncc32"], rtol=1e-4, atol=1e-5 * np.abs(g[f"{case}/dncc32"]).max())
    for case in FLOW_CASES:
        shape = tuple(g[f"{case}/shape"])
        mov = ph.vol(shape, 0.37, "sin")
        fl = ph.flow_field(shape, float(g[f"{case}/amp"]))
        w = compose.flow_warp(mov, fl)
        assert np.allclose(w.numpy(), g[f"{case}/warped32"], atol=1e-6)


def test_ncc_self_and_norm(single_step):
    x = ph.vol((5, 6, 7), 0.37)[0, 0].numpy()
    total, _ = oracle.c_loss_terms(x, x, oracle.wts(w_ncc=1.0))
    assert abs(total - float(single_step["ncc_self"])) < 1e-4
    assert abs(total) < 1e-4


# ----------------------------------------------------------------------------- trajectories
def _mov_tgt(g, name, dtype):
    shape = tuple(g[f"{name}/shape"])
    seed = int(g[f"{name}/meta"][2])
    tgt = ph.blobs(shape, 1000 + seed)[0, 0].numpy().astype(dtype)
    mov = g[f"{name}/moving"][0, 0].astype(dtype)
    return mov, tgt


@pytest.mark.parametrize("name,loss", [("rigid2d_mse", "mse"), ("rigid3d_mse", "mse"), ("affine3d_mse", "mse"),
                                       ("affine2d_mse", "mse")])
def test_c_oracle_driver_trajectories(trajectories, name, loss):
    """ref affine_register / rigid_register (through Register) vs the C-oracle loop."""
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    pose0 = g[f"{name}/init"] if f"{name}/init" in g else None
    # fp64 loop vs the fp64 arbiter trajectory
    mov, tgt = _mov_tgt(g, name, np.float64)
    r = oracle.c_affine_loop(mov, tgt, oracle.wts(**LOSSES[loss]), lr, iters, pose0=None if pose0 is None else pose0.astype(np.float64),
                             tables=oracle.base_tables(mov.shape, np.float64))
    assert relerr(r["losses"], g[f"{name}/losses64"]) < 1e-8
    assert np.max(np.abs(r["thetas"] - g[f"{name}/thetas64"])) < 1e-8
    # fp32 loop vs the reference's own fp32 run, bar from the fp32-vs-fp64 gap
    mov, tgt = _mov_tgt(g, name, np.float32)
    tabs = oracle.base_tables(mov.shape)
    r = oracle.c_affine_loop(mov, tgt, oracle.wts(**LOSSES[loss]), lr, iters, pose0=pose0, tables=tabs)
    l32, l64 = g[f"{name}/losses"], g[f"{name}/losses64"]
    assert np.max(np.abs(r["losses"] - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(r["final_theta"] - g[f"{name}/final_theta"][0])) <= bar(g[f"{name}/final_theta"][0], g[f"{name}/thetas64"][-1], 1e-4)
    # best-theta semantics (Q8): first strict minimum, theta before that step
    best_idx = int(np.argmin(l32))
    assert np.max(np.abs(r["thetas"][best_idx] - g[f"{name}/best_theta"][0])) <= bar(g[f"{name}/final_theta"][0], g[f"{name}/thetas64"][-1], 1e-4)


@pytest.mark.parametrize("name,rigid", [("c_affine3d_ncc", False), ("c_affine2d_ncc", False), ("c_rigid3d_ncc", True)])
def test_c_oracle_ncc_trajectories(trajectories, name, rigid):
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    pose0 = g[f"{name}/init"] if rigid else None
    mov, tgt = _mov_tgt(g, name, np.float64)
    r = oracle.c_affine_loop(mov, tgt, oracle.wts(w_ncc=1.0), lr, iters, pose0=None if pose0 is None else pose0.astype(np.float64),
                             tables=oracle.base_tables(mov.shape, np.float64))
    assert relerr(r["losses"], g[f"{name}/losses64"]) < 1e-7
    assert np.max(np.abs(r["thetas"] - g[f"{name}/thetas64"])) < 1e-7
    mov, tgt = _mov_tgt(g, name, np.float32)
    r = oracle.c_affine_loop(mov, tgt, oracle.wts(w_ncc=1.0), lr, iters, pose0=pose0, tables=oracle.base_tables(mov.shape))
    l32, l64 = g[f"{name}/losses32"], g[f"{name}/losses64"]
    assert np.max(np.abs(r["losses"] - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(r["thetas"] - g[f"{name}/thetas32"])) <= bar(g[f"{name}/thetas32"], g[f"{name}/thetas64"], 1e-4)


@pytest.mark.parametrize("name,loss", [("c_flow3d_ncc", "ncc"), ("c_flow3d_mse", "mse"), ("c_flow2d_ncc", "ncc")])
def test_c_oracle_flow_trajectories(trajectories, name, loss):
    g = trajectories
    lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
    mov, tgt = _mov_tgt(g, name, np.float64)
    r = oracle.c_flow_loop(mov, tgt, oracle.wts(**LOSSES[loss]), lr, iters)
    # the fp64 golden ran with the reference's float32 identity-grid buffer promoted: exact match expected
    assert relerr(r["losses"], g[f"{name}/losses64"]) < 1e-6
    assert np.max(np.abs(r["flow"] - g[f"{name}/flow64"][0])) < 1e-6 * max(1.0, np.abs(g[f"{name}/flow64"]).max())
    mov, tgt = _mov_tgt(g, name, np.float32)
    r = oracle.c_flow_loop(mov, tgt, oracle.wts(**LOSSES[loss]), lr, iters)
    l32, l64 = g[f"{name}/losses32"], g[f"{name}/losses64"]
    assert np.max(np.abs(r["losses"] - l32)) <= bar(l32, l64, 1e-4 * np.max(np.abs(l64)))
    assert np.max(np.abs(r["flow"] - g[f"{name}/flow32"][0])) <= bar(g[f"{name}/flow32"], g[f"{name}/flow64"], 1e-4)


def test_compose_trajectory_bitwise(trajectories):
    """The torch composition reproduces the reference's fp32 runs exactly (same ATen ops)."""
    g = trajectories
    for name, kw in (("affine3d_mse", dict(w_mse=1.0)), ("rigid3d_mse", dict(w_mse=1.0))):
        lr, iters = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1])
        shape = tuple(g[f"{name}/shape"])
        seed = int(g[f"{name}/meta"][2])
        tgt = ph.blobs(shape, 1000 + seed)
        mov = torch.from_numpy(g[f"{name}/moving"])
        pose0 = torch.from_numpy(g[f"{name}/init"]) if f"{name}/init" in g else None
        r = compose.affine_loop(mov, tgt, lr, iters, pose0=pose0, **kw)
        assert np.allclose(r["losses"].numpy(), g[f"{name}/losses"], rtol=1e-6, atol=0)
        assert np.allclose(r["thetas"][-1].numpy(), g[f"{name}/final_theta"][0], atol=1e-6)
        assert np.allclose(r["thetas"][r["best_idx"]].numpy(), g[f"{name}/best_theta"][0], atol=1e-6)
    name = "c_affine3d_ncc"
    lr, iters, seed = float(g[f"{name}/meta"][0]), int(g[f"{name}/meta"][1]), int(g[f"{name}/meta"][2])
    tgt = ph.blobs(tuple(g[f"{name}/shape"]), 1000 + seed)
    r = compose.affine_loop(torch.from_numpy(g[f"{name}/moving"]), tgt, lr, iters, w_ncc=1.0)
    assert np.allclose(r["losses"].numpy(), g[f"{name}/losses32"], rtol=1e-5)


def test_local_ncc_definition_properties():
    """The local-NCC specification (oracle/compose.py::local_ncc_loss, an extension with no reference counterpart):
    cc = 1 for proportional intensities, invariance to the gain of the warped image (an offset is NOT invariant at the
    border: the zero padding enters the window sums with n fixed), and agreement with an explicit loop on a tiny image."""
    from oracle import compose
    torch.manual_seed(3)
    y = torch.rand(1, 1, 7, 8, 9, dtype=torch.float64)
    assert compose.local_ncc_loss(y, 2.5 * y, eps=0.0).abs().item() < 1e-9
    yp = torch.rand(1, 1, 7, 8, 9, dtype=torch.float64)
    a, b = compose.local_ncc_loss(y, yp, 5, eps=0.0), compose.local_ncc_loss(y, 3.0 * yp, 5, eps=0.0)
    assert abs(a.item() - b.item()) < 1e-9
    # explicit loop, 2-D, window 3
    y2, p2 = torch.rand(1, 1, 5, 6, dtype=torch.float64), torch.rand(1, 1, 5, 6, dtype=torch.float64)
    cc = []
    for i in range(5):
        for j in range(6):
            s = [0.0] * 5
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    ii, jj = i + di, j + dj
                    if 0 <= ii < 5 and 0 <= jj < 6:
                        u, v = y2[0, 0, ii, jj].item(), p2[0, 0, ii, jj].item()
                        s = [s[0] + u, s[1] + v, s[2] + u * u, s[3] + v * v, s[4] + u * v]
            c = s[4] - s[0] * s[1] / 9
            cc.append(c * c / ((s[2] - s[0] ** 2 / 9) * (s[3] - s[1] ** 2 / 9) + 1e-5))
    assert abs(compose.local_ncc_loss(y2, p2, 3).item() - (1 - sum(cc) / 30)) < 1e-12


# ----------------------------------------------------------------------------- round 3: nearest-mode SpatialTransformer
def _r3():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixtures_r3.npz"))


def _smooth_flow(shape, seed, amp):
    """tests/golden/make_golden_r3.py::smooth_flow (closed form)"""
    nd = len(shape)
    axes = torch.meshgrid(*[torch.arange(s, dtype=torch.float64) for s in shape], indexing="ij")
    chans = []
    for c in range(nd):
        f = torch.zeros(shape, dtype=torch.float64)
        for k in range(3):
            arg = sum((0.041 + 0.013 * ((c + k + d) % 3)) * axes[d] for d in range(nd)) + 0.37 * seed + 1.1 * c + 0.7 * k
            f = f + torch.sin(arg) / (k + 1.0)
        chans.append(amp * f)
    return torch.stack(chans)[None].float()


@pytest.mark.parametrize("name,shape", [("st_nearest_2d", (48, 64)), ("st_nearest_3d", (20, 36, 28))])
def test_compose_nearest_warp_matches_reference(name, shape):
    """oracle/compose.flow_warp(mode='nearest') is the reference's SpatialTransformer(mode='nearest') bit for bit; its voxel-space
    restatement (the HIP kernel's definition) agrees everywhere except at the listed ties."""
    from oracle import compose
    g = _r3()
    amp, seed = g[f"{name}/meta"]
    src = ph.blobs(shape, 901) + 0.2 * ph.vol(shape, 0.021, "sin")
    flow = _smooth_flow(shape, int(seed), float(amp))
    ref = torch.from_numpy(g[f"{name}/warped"])
    assert torch.equal(compose.flow_warp(src, flow, mode="nearest"), ref)
    vox = compose.flow_warp_nearest_voxel_space(src, flow)
    tie = torch.from_numpy(g[f"{name}/tie"])[:, None]
    assert torch.equal(vox[~tie], ref[~tie])
    assert (ref == 0).float().mean() > 0.01   # part of the flow leaves the image: the zero padding is exercised
