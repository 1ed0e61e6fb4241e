"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the TorchRegister hot path used as the parity checker.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
package (torchregister_amd) never does and fails loudly when its HIP library is missing.

Two independent checkers live here:
  * oracle.c_*      — ctypes bindings of oracle/liboracle.so (trx_oracle.c: plain C, fp32 and
                      fp64 instantiations; see the header of trx_oracle_body.h for citations)
  * oracle.compose  — the same loops composed from torch's own CPU ops at exactly the
                      reference's call sites (F.affine_grid/F.grid_sample/torch.optim.SGD);
                      this is also what bench.py times as the CPU baseline ("port").
Both are pinned against tests/golden/ (outputs of the imported reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WTS_FIELDS = ("w_mse", "w_ncc", "ncc_alpha", "w_ssd", "ssd_alpha")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("trx_oracle.c", "trx_oracle_body.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _suf(dtype):
    return "_f32" if np.dtype(dtype) == np.float32 else "_f64"


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dims(shape):
    if len(shape) == 2:
        return 2, 1, int(shape[0]), int(shape[1])
    return 3, int(shape[0]), int(shape[1]), int(shape[2])


def wts(w_mse=0.0, w_ncc=0.0, ncc_alpha=100.0, w_ssd=0.0, ssd_alpha=3.0):
    return np.asarray([w_mse, w_ncc, ncc_alpha, w_ssd, ssd_alpha], dtype=np.float64)


def base_tables(shape, dtype=np.float32):
    """Per-axis base coordinates exactly as ATen builds them: linspace(-1,1,S)*(S-1)/S in `dtype`."""
    import torch
    td = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
    return [(torch.linspace(-1, 1, int(s), dtype=td) * (int(s) - 1) / int(s)).numpy().copy() for s in shape]


def c_affine_warp(mov, theta, tables=None):
    """mov: ndarray [*spatial]; theta [nd, nd+1]; returns warped ndarray (same dtype)."""
    mov = np.ascontiguousarray(mov)
    dt = mov.dtype
    nd, D, H, W = _dims(mov.shape)
    th = np.ascontiguousarray(theta, dtype=dt).reshape(-1)
    out = np.empty_like(mov)
    tz, ty, tx = (None, None, None) if tables is None else ((None,) + tuple(tables) if nd == 2 else tuple(tables))
    getattr(lib(), "orc_affine_warp" + _suf(dt))(_p(mov), _p(th), _p(out), nd, D, H, W, _p(tz), _p(ty), _p(tx))
    return out


def c_affine_loss_grad(mov, tgt, theta, w, tables=None):
    """Returns (total_loss, terms[3], dtheta [nd, nd+1], warped)."""
    mov, tgt = np.ascontiguousarray(mov), np.ascontiguousarray(tgt)
    dt = mov.dtype
    nd, D, H, W = _dims(mov.shape)
    th = np.ascontiguousarray(theta, dtype=dt).reshape(-1)
    warped = np.empty_like(mov)
    terms = np.zeros(3, dtype=np.float64)
    dth = np.zeros(nd * (nd + 1), dtype=dt)
    w = np.ascontiguousarray(w, dtype=np.float64)
    tz, ty, tx = (None, None, None) if tables is None else ((None,) + tuple(tables) if nd == 2 else tuple(tables))
    getattr(lib(), "orc_affine_loss_grad" + _suf(dt))(_p(mov), _p(tgt), _p(th), nd, D, H, W, _p(tz), _p(ty), _p(tx),
                                                     _p(w), _p(warped), _p(terms), _p(dth))
    total = w[0] * terms[0] + w[1] * terms[1] + w[3] * terms[2]
    return total, terms, dth.reshape(nd, nd + 1), warped


def c_loss_terms(tgt, warped, w):
    tgt, warped = np.ascontiguousarray(tgt), np.ascontiguousarray(warped, dtype=tgt.dtype)
    terms = np.zeros(3, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    fn = getattr(lib(), "orc_loss_terms" + _suf(tgt.dtype))
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    fn(_p(tgt), _p(warped), tgt.size, _p(w), _p(terms))
    return w[0] * terms[0] + w[1] * terms[1] + w[3] * terms[2], terms


def c_theta_fwd(x):
    x = np.ascontiguousarray(x)
    nd = 3 if x.size == 6 else 2
    th = np.zeros(nd * (nd + 1), dtype=x.dtype)
    getattr(lib(), "orc_theta_fwd" + _suf(x.dtype))(_p(x), nd, _p(th))
    return th.reshape(nd, nd + 1)


def c_theta_vjp(x, g):
    x = np.ascontiguousarray(x)
    nd = 3 if x.size == 6 else 2
    g = np.ascontiguousarray(g, dtype=x.dtype).reshape(-1)
    dx = np.zeros_like(x)
    getattr(lib(), "orc_theta_vjp" + _suf(x.dtype))(_p(x), nd, _p(g), _p(dx))
    return dx


def c_flow_warp(mov, flow):
    mov, flow = np.ascontiguousarray(mov), np.ascontiguousarray(flow)
    nd, D, H, W = _dims(mov.shape)
    out = np.empty_like(mov)
    getattr(lib(), "orc_flow_warp" + _suf(mov.dtype))(_p(mov), _p(flow), _p(out), nd, D, H, W)
    return out


def c_flow_loss_grad(mov, tgt, flow, w):
    """Returns (total_loss, terms, dflow [nd,*spatial], warped)."""
    mov, tgt, flow = np.ascontiguousarray(mov), np.ascontiguousarray(tgt), np.ascontiguousarray(flow)
    nd, D, H, W = _dims(mov.shape)
    warped = np.empty_like(mov)
    dfl = np.empty_like(flow)
    terms = np.zeros(3, dtype=np.float64)
    w = np.ascontiguousarray(w, dtype=np.float64)
    getattr(lib(), "orc_flow_loss_grad" + _suf(mov.dtype))(_p(mov), _p(tgt), _p(flow), nd, D, H, W, _p(w), _p(warped),
                                                          _p(terms), _p(dfl))
    return w[0] * terms[0] + w[1] * terms[1] + w[3] * terms[2], terms, dfl, warped


# ------------------------------------------------------------------------------------- loops
def c_affine_loop(mov, tgt, w, lr, iters, pose0=None, theta0=None, tables=None):
    """The reference's driver loop (ref:warpings.py:67-93 affine, :138-159 rigid) on the C oracle.

    SGD on theta (affine; Q3: the MLP is dead, theta starts at identity) or on the pose vector
    (rigid, Theta chain).  Returns dict(losses[iters], thetas[iters+1], best_idx, best_theta,
    final_theta).  Loss values are rounded to the data dtype before the strict-< best test,
    like error.item() of an fp32 tensor.
    """
    dt = mov.dtype
    nd = mov.ndim
    rigid = pose0 is not None
    if rigid:
        p = np.array(pose0, dtype=dt)
    else:
        p = np.eye(nd, nd + 1, dtype=dt) if theta0 is None else np.array(theta0, dtype=dt).reshape(nd, nd + 1)
    lr = dt.type(lr)
    losses, thetas, best, best_idx = [], [], None, -1
    for t in range(iters):
        th = c_theta_fwd(p) if rigid else p
        thetas.append(np.array(th, copy=True))
        total, _, dth, _ = c_affine_loss_grad(mov, tgt, th, w, tables)
        total = float(dt.type(total))
        losses.append(total)
        if best is None or total < best:
            best, best_idx = total, t
        g = c_theta_vjp(p, dth) if rigid else dth
        p = (p - lr * g).astype(dt)
    thetas.append(np.array(c_theta_fwd(p) if rigid else p, copy=True))
    return dict(losses=np.asarray(losses), thetas=np.asarray(thetas), best_idx=best_idx,
                best_theta=thetas[best_idx], final_theta=thetas[-1], pose=p if rigid else None)


def c_flow_loop(mov, tgt, w, lr, iters, flow0=None):
    dt = mov.dtype
    nd = mov.ndim
    fl = np.zeros((nd,) + mov.shape, dtype=dt) if flow0 is None else np.array(flow0, dtype=dt)
    lr = dt.type(lr)
    losses = []
    for _ in range(iters):
        total, _, dfl, _ = c_flow_loss_grad(mov, tgt, fl, w)
        losses.append(float(dt.type(total)))
        fl = (fl - lr * dfl).astype(dt)
    return dict(losses=np.asarray(losses), flow=fl, final_warped=c_flow_warp(mov, fl))
