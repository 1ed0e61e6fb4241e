"""ctypes binding of libtrx.so (the C ABI declared in include/trx.h).

There is NO fallback: if the HIP library is missing or a call fails, an exception is raised.
`import torch` must happen before the library is loaded so that libtrx.so binds to the same
libamdhip64 (SONAME libamdhip64.so.7) that PyTorch-ROCm already mapped — streams and device
pointers are only meaningful inside one HIP runtime.
"""
import ctypes
import os

import torch  # noqa: F401  (loads PyTorch's HIP runtime first; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtrx.so")

PSTRIDE = 12
OPT_SGD, OPT_ADAM = 0, 1
PARAM_AFFINE, PARAM_RIGID = 0, 1
# trx_volumes.flags (include/trx.h)
FLAG_GATHER_PATH, FLAG_SINGLE_GEOM, FLAG_TWO_PASS_FLOW, FLAG_DEEP_TILE, FLAG_NO_ROT_DEEP_TILE, FLAG_NO_ZSTREAM, FLAG_ZSTREAM, FLAG_NEAREST, FLAG_SAVE_LAST = 1, 2, 4, 8, 16, 32, 64, 128, 256
FLAG_NO_EFT, FLAG_EFT = 512, 1024
FLAG_WALK_DOWN, FLAG_NO_PINGPONG = 8192, 16384   # z-streaming columns walked downward / trx_affine_run does not alternate the direction
FLAG_NO_ZS_FLAT = 4096   # the z-streaming kernel without its flat tile (measured alternative)
FLAG_ONE_KERNEL = 32768   # chip-filling launches next to the identity: the z-streaming kernel alone (it runs stray pairs on GeomR's body)
FLAG_NO_CARRY = 65536    # trx_affine_run keeps a step kernel + a finalise kernel per iteration (measured alternative of the carry form)
FLAG_ZS_FUSED = 2048   # keep the z-streaming body inside the tile kernel (measured alternative of round 5)

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class TrxError(RuntimeError):
    pass


class Volumes(ctypes.Structure):
    _fields_ = [("moving", ctypes.c_void_p), ("target", ctypes.c_void_p),
                ("moving_stride", ctypes.c_size_t), ("target_stride", ctypes.c_size_t),
                ("ndim", ctypes.c_int), ("B", ctypes.c_int), ("D", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("xn", ctypes.c_void_p), ("yn", ctypes.c_void_p), ("zn", ctypes.c_void_p), ("flags", ctypes.c_uint)]


class LossCfg(ctypes.Structure):
    _fields_ = [("w_mse", ctypes.c_float), ("w_ncc", ctypes.c_float), ("ncc_alpha", ctypes.c_float),
                ("w_ssd", ctypes.c_float), ("ssd_alpha", ctypes.c_float)]


class OptCfg(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("eps", ctypes.c_float)]


class AffineState(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("param", ctypes.c_void_p), ("theta", ctypes.c_void_p),
                ("adam_m", ctypes.c_void_p), ("adam_v", ctypes.c_void_p), ("best_theta", ctypes.c_void_p),
                ("best_loss", ctypes.c_void_p), ("best_idx", ctypes.c_void_p), ("losses", ctypes.c_void_p),
                ("losses_capacity", ctypes.c_int), ("step", ctypes.c_void_p), ("grad", ctypes.c_void_p)]


class FlowState(ctypes.Structure):
    _fields_ = [("flow", ctypes.c_void_p), ("flow_tmp", ctypes.c_void_p), ("adam_m", ctypes.c_void_p),
                ("adam_v", ctypes.c_void_p), ("losses", ctypes.c_void_p), ("losses_capacity", ctypes.c_int),
                ("step", ctypes.c_void_p), ("smooth_weight", ctypes.c_float), ("stop_crit", ctypes.c_float),
                ("stopped", ctypes.c_void_p), ("flow_last", ctypes.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/trx.h declares (tests check this)
_P = ctypes.c_void_p
SIGNATURES = {
    "trx_version": (ctypes.c_int, []),
    "trx_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "trx_affine_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Volumes)]),
    "trx_affine_workspace_rows_offset": (ctypes.c_size_t, [ctypes.POINTER(Volumes)]),
    "trx_affine_near_identity": (ctypes.c_int, [ctypes.POINTER(Volumes), _P]),
    "trx_affine_step": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), ctypes.POINTER(OptCfg),
                                       ctypes.POINTER(AffineState), _P, ctypes.c_size_t, _P]),
    "trx_affine_accumulate": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, _P, ctypes.c_size_t, _P]),
    "trx_affine_run": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), ctypes.POINTER(OptCfg),
                                      ctypes.POINTER(AffineState), ctypes.c_int, _P, ctypes.c_size_t, _P]),
    "trx_affine_loss": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_affine_warp": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, ctypes.c_int, _P, _P]),
    "trx_affine_warp_backward": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, ctypes.c_int, _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_theta_chain": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P]),
    "trx_affine_warp_lattice": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, _P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P]),
    "trx_affine_warp_lattice_backward": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, _P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P,
                                                        ctypes.c_size_t, _P]),
    "trx_flow_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Volumes)]),
    "trx_flow_warp": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, ctypes.c_int, _P, _P]),
    "trx_flow_step": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), ctypes.POINTER(OptCfg),
                                     ctypes.POINTER(FlowState), _P, ctypes.c_size_t, _P]),
    "trx_flow_run": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), ctypes.POINTER(OptCfg),
                                    ctypes.POINTER(FlowState), ctypes.c_int, _P, ctypes.c_size_t, _P]),
    "trx_flow_slab_moments": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_flow_slab_boundary_smooth": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, _P, _P, _P]),
    "trx_peer_signal": (ctypes.c_int, [_P, ctypes.c_uint, _P]),
    "trx_peer_wait": (ctypes.c_int, [_P, ctypes.c_uint, ctypes.c_uint, _P, _P]),
    "trx_peer_publish": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_uint, _P]),
    "trx_peer_gather": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, _P, _P, _P]),
    "trx_peer_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "trx_peer_free": (ctypes.c_int, [_P]),
    "trx_peer_export": (ctypes.c_int, [_P, _P]),
    "trx_peer_import": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_void_p)]),
    "trx_peer_close": (ctypes.c_int, [_P]),
    "trx_flow_slab_update": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.c_int, ctypes.c_int, ctypes.POINTER(LossCfg),
                                            ctypes.POINTER(OptCfg), ctypes.POINTER(FlowState), _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_flow_slab_update_fused": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.c_int, ctypes.c_int, ctypes.POINTER(LossCfg),
                                                  ctypes.POINTER(OptCfg), ctypes.POINTER(FlowState), _P, _P, ctypes.c_size_t, _P]),
    "trx_flow_slab_moments_ready": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_size_t, _P]),
    "trx_flow_loss_grad": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.POINTER(LossCfg), _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_flow_warp_backward": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, ctypes.c_int, _P, _P, _P]),
    "trx_kde_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_long, ctypes.c_int]),
    "trx_kde_pdf": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, _P, _P, ctypes.c_size_t, _P]),
    "trx_kde_pdf_backward": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, _P, _P]),
    "trx_kde_series_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_long, ctypes.c_int]),
    "trx_kde_pdf_series": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_double, _P, _P, ctypes.c_size_t, _P]),
    "trx_kde_pdf_series_backward": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_double, _P, _P,
                                                   ctypes.c_size_t, _P]),
    "trx_kde_pdf_series_cached": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_double, _P, _P]),
    "trx_nmi_from_pdfs_pooled": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, _P, _P, _P, _P, _P]),
    "trx_nmi_lattice_lines": (ctypes.c_int, [ctypes.POINTER(Volumes), _P, _P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P, ctypes.c_int,
                                             ctypes.c_int, _P, _P, _P, ctypes.c_size_t, _P]),
    "trx_nmi_loop_update": (ctypes.c_int, [ctypes.c_int, _P, _P, _P, _P, ctypes.c_float, _P, ctypes.c_int, _P, _P, _P, _P, _P]),
    "trx_nmi_from_pdfs": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "trx_flow_lncc_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Volumes)]),
    "trx_flow_lncc_run": (ctypes.c_int, [ctypes.POINTER(Volumes), ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.POINTER(OptCfg),
                                         ctypes.POINTER(FlowState), ctypes.c_int, _P, ctypes.c_size_t, _P]),
    "trx_lncc_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "trx_lncc_loss_grad": (ctypes.c_int, [_P, _P] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_float, _P, _P, _P, ctypes.c_size_t, _P]),
}

_lib = None


def load():
    """Load libtrx.so (once). Raises TrxError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TrxError(f"HIP library not built: {LIB_PATH} is missing. Run `make` (or "
                       f"`python -c 'import __graft_entry__ as g; g.build()'`) at the repo root. "
                       f"There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().trx_status_string(rc).decode()
        raise TrxError(f"{what} failed: {msg} (status {rc})")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
