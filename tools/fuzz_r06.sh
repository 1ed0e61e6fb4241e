#!/bin/bash
# Round 6: the randomised sweeps on the final library (one GPU call) -> gpurun_out/<tag>/fuzz.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
{ sha256sum torchregister_amd/lib/libtrx.so
  echo "== tests/fuzz_affine.py 500 61";      python3 tests/fuzz_affine.py 500 61 2>&1 | grep -v amdgpu | tail -4
  echo "== FUZZ_BIG tests/fuzz_affine.py 6 61"; FUZZ_BIG=1 python3 tests/fuzz_affine.py 6 61 2>&1 | grep -v amdgpu | tail -3
  echo "== tests/fuzz_zstream.py 300 62";     python3 tests/fuzz_zstream.py 300 62 2>&1 | grep -v amdgpu | tail -4
  echo "== tests/fuzz_zs_flat.py 40 65";      python3 tests/fuzz_zs_flat.py 40 65 2>&1 | grep -v amdgpu | tail -4
  echo "== tests/fuzz_misc.py 360 64";        python3 tests/fuzz_misc.py 360 64 2>&1 | grep -v amdgpu | tail -4
  echo "== tests/fuzz_flow_lncc.py 300 63";   python3 tests/fuzz_flow_lncc.py 300 63 2>&1 | grep -v amdgpu | tail -4
  echo "== tests/fuzz_degenerate.py";         python3 tests/fuzz_degenerate.py 2>&1 | grep -v amdgpu | tail -3
} > $O/fuzz.txt 2>&1
tail -40 $O/fuzz.txt
