"""Cases that randomised sweeps found ABOVE the bars of their time, pinned (a third from round 5 at the end of the file): each bar was widened after the case, so
each case is now a fixed regression test that holds the widened bar - and records, in its assertions, by how much the case needs it.

1. tests/fuzz_flow_lncc.py, seed 51, case 231: 2-D 61 x 59, window 3, two pairs.  Local-window NCC gradient 3.08e-4 of its maximum
   against 2 x 1.46e-4 = 2.92e-4, the fp32-vs-fp64 gap of the torch specification itself (compose.local_ncc_loss evaluated in fp32 and
   fp64).  Analysis in fp64: with a window of 3 x 3 = 9 voxels the local variance is a difference of two nearly equal sums (sum w^2 and
   (sum w)^2 / 9 agree to 3-4 digits where the blob phantoms are flat); torch's conv sums the nine products in one order, the kernels
   slide sums along x (add the entering column, subtract the leaving one), and the two orders lose different last bits of that
   difference, which the 1 / (var_w var_y) factor of the gradient then amplifies.  The kernel's error is 2.1 x torch's own fp32 error at
   the worst voxel, not a different formula: the fp64 specification is met to 3.1e-4 of the gradient's maximum.  Bar since: 2.5 x.
   (Round 5: with the x window summed in registers - TRX_LNCC_DIRECT - the same case is at 0.99 x torch's fp32 error.)
2. tests/fuzz_zstream.py, seed 13, case 111: next to the identity whole bands of voxels sample within fp32 rounding of a lattice plane,
   where trilinear interpolation has a kink (the derivative jumps by up to a voxel value).  The oracle measures that sensitivity on
   itself: its fp64 gradient re-evaluated with the translations nudged by +-1 fp32 ulp of a coordinate (kink_variants).  The kernels'
   coordinate arithmetic (base + row term, fused multiply-adds) rounds differently from the oracle's expression, so the band they put
   on the other side of a kink is not the band the nudge moves: BOTH kernel families (z-streaming body and tile kernels) are 1.6 x the
   nudge's movement away from the fp64 gradient and agree with each other to 1e-5 - the disagreement is between fp32 coordinate
   roundings, not between kernels.  Bar since: 2 x the sensitivity (1.5 x in fuzz_affine, whose poses are not lattice-aligned)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_lncc_window3_seed51_case231():
    import fuzz_flow_lncc
    det = []
    fails, _ = fuzz_flow_lncc.run(232, 51, verbose=True, only=231, details=det)
    assert len(det) == 1 and det[0]["case"] == 231 and det[0]["shape"] == (61, 59) and det[0]["win"] == 3, det
    d = det[0]
    assert fails == 0, d
    ratio = d["lncc_grad_err"] / d["lncc_grad_fp32_gap"]
    # round 4's kernels (x window by sliding sums over the LDS tile) sat at 2.11 x torch's own fp32 error here - over the factor 2 of that time -;
    # round 5's direct form (x window summed in registers, windows 3 and 5) sits at 0.99 x.  The case holds the CURRENT bar (2.5 x) and, absolutely,
    # the fp64 specification to 4e-4 of the gradient's maximum
    assert ratio <= 2.5, (ratio, d)
    assert d["lncc_grad_err"] <= 4e-4, d
    assert d["lncc_loss_err"] <= d["lncc_loss_bar"], d


def test_zstream_kink_seed13_case111():
    import fuzz_zstream
    det = []
    fails, _ = fuzz_zstream.run(112, 13, verbose=True, only=111, details=det)
    assert fails == 0 and det and all(d["case"] == 111 for d in det), det
    worst = max(det, key=lambda d: max(d["err_zs"], d["err_tile"]) / max(d["ksens"], 1e-30))
    over = max(worst["err_zs"], worst["err_tile"]) / worst["ksens"]
    # the pair that needed the wider bar: both kernel families beyond 1.5 x the oracle's own kink sensitivity, inside 2 x, and equal to each other
    assert worst["ksens"] > 2e-4, worst                      # (the bar of this pair IS the kink bar, not the 2e-4 floor)
    assert over <= 2.0, (over, worst)
    assert worst["zs_vs_tile"] <= 0.1 * worst["ksens"], worst


def test_flow_single_sample_seed65_case32():
    """Round 5, `python tests/fuzz_flow_lncc.py 100 65`, case 32: a 3 x 7 image under flows of +-6 pixels, MSE + NCC.  ONE sample of the warped
    image lies inside the moving image; dL/dflow is non-zero at that voxel only (-9.5e-4, -6.3e-4) and the kernels differ from the fp64
    oracle there by 2.5e-6 = 2.6e-3 of the maximum - thirteen times the 2e-4 bar, while the oracle's own fp32 build differs by 3e-10.
    Analysis (fp64 oracle, this file's docstring of record): NCC does not change when w is scaled, so with a single non-zero sample its
    derivative at that voxel is zero by symmetry (NCC alone: -5e-8, MSE alone: -1.7e-3) - as the sum cy y + cw w + c0 of three terms of
    magnitude ~40 (ncc_alpha = 100).  The kernels evaluate that sum in fp32 (as torch's fp32 autograd of the reference does); the oracle's fp32
    build keeps the coefficients and the sum in double.  2^-24 x 40 x |grad w| = 2.4e-6 is what is measured: rounding of a cancellation, not
    an error of the path.  The sweep's bar now contains that floor (fuzz_flow_lncc.cancellation_floor, from the fp64 oracle alone); this test
    holds the case against it and checks that the floor explains the error to within a factor of 4."""
    from fuzz_flow_lncc import run
    det = []
    fails, _ = run(33, 65, verbose=True, only=32, details=det)
    assert fails == 0, det
    d = det[0]
    assert d["shape"] == (3, 7)
    assert 2e-4 < d["flow_grad_err"] <= d["flow_grad_bar"] < 4.0 * max(d["flow_grad_err"], 1e-3), d
