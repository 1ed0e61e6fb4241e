"""Randomised parity sweeps.  tests/fuzz_affine.py (fused affine and rigid steps, forward warp, warp backward): random ragged shapes
(3..99 per axis, W % 4 != 0 included), batches of 1-3 pairs, theta from near-identity to large rotations / zoom / flips,
random MSE + NCC weights; checker = the C oracle in fp64 (gradient bar 3e-4 of max or twice the fp32 oracle's own gap:
random large rotations sit a little above the 2e-4 floor of the fixed cases; 1 of 400 cases reached 2.7e-4).
The case counts here are sized for the suite's time limit (round 6: 36 / 40 / 42 / 18 / 4 cases, ~2.5 minutes in all); the full sweeps
(hundreds of cases per file, `python tests/fuzz_*.py N seed`) run per round on the final library and are recorded in profiles/r0N*_fuzz.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_random_sweep():
    import fuzz_affine
    fails, worst = fuzz_affine.run(36, 2024, grad_bar=3e-4, verbose=True)
    assert fails == 0, worst


def test_random_sweep_flow_and_local_ncc():
    """Dense-flow loss + dL/dflow vs the C oracle, local-window NCC loss + gradient vs its torch-conv specification:
    random 2-D / 3-D shapes, flow amplitudes 0.3 .. 6 voxels, windows 3..9, batches (tests/fuzz_flow_lncc.py)."""
    import fuzz_flow_lncc
    fails, worst = fuzz_flow_lncc.run(40, 11, verbose=True)
    assert fails == 0, worst


def test_degenerate_shapes():
    """Axes of 1..5 voxels in 2-D and 3-D through the affine step, the forward warp, the flow loss/gradient and the local NCC
    (tests/fuzz_degenerate.py).  Two corners are outside the reference's domain and excluded there: a flow along an axis of
    one voxel (the reference divides by S - 1) and the NCC of fewer than 8 voxels."""
    import fuzz_degenerate
    assert fuzz_degenerate.run(verbose=True) == 0


def test_random_sweep_2d_channels_trajectories():
    """2-D affine / rigid steps, warp and warp backward; multi-channel warps; loss-only evaluation; short SGD trajectories
    (loss curve, best index, final theta) in 2-D and 3-D (tests/fuzz_misc.py)."""
    import fuzz_misc
    fails, worst = fuzz_misc.run(42, 7, verbose=True)
    assert fails == 0, worst


def test_random_sweep_zstream_body():
    """The z-streaming F1 body at random distances from the identity - inside its window, at the edge (re-anchoring) and beyond (fallback
    inside the same launch) - mixed batches, NCC and MSE-only steps, against the C oracle and against the tile kernels
    (tests/fuzz_zstream.py)."""
    import fuzz_zstream
    fails, worst = fuzz_zstream.run(18, 5, grad_bar=3e-4, verbose=True)
    assert fails == 0, worst


def test_random_sweep_chip_filling_launches():
    """Round 5: launches that fill the chip - the z-streaming kernel in front (both tiles, walking up or down), the exact-footprint kernel and the
    tile kernel behind it in ONE launch, poses per pair from the identity to general rotations - against the same launch on the tile kernels
    alone (tests/fuzz_zs_flat.py)."""
    import fuzz_zs_flat
    fails, worst = fuzz_zs_flat.run(4, 3, verbose=True)
    assert fails == 0, worst
