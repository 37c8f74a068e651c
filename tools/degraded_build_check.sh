#!/bin/bash
# Does the aggregate parity guard (tests/helpers.py: AGG_MEDIAN_BOUND) catch a UNIFORM loss of precision that the per-step bounds
# max(5 x TOL, 3 x floor) let through?  Builds the library with the triplet kernel's activations cut by PG_DEGRADE_BITS mantissa bits
# (default 9: ~14 bits left, a ~10x precision regression of one kernel) into phoregen_amd/_lib_degraded and runs the sampler parity
# tests against it; then the same tests on the product build.  Run on the GPU box from the repository root:
#   bash tools/degraded_build_check.sh [bits] > gpurun_out/r04_degraded_build_check.txt
bits=${1:-9}
make -C phoregen_amd/csrc -j8 OUT=../_lib_degraded EXTRA=-DPG_DEGRADE_BITS=$bits > /dev/null || exit 1
echo "== degraded build (triplet activations lose $bits mantissa bits): sampler parity tests =="
PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=$PWD/phoregen_amd/_lib_degraded/libphoregen_hip.so python3 -m pytest tests/test_gpu_parity.py -q -m gpu \
  -k "teacher_forced or closed_loop" 2>&1 | grep -E "passed|failed|median\(err|^FAILED|AssertionError" | cut -c1-260
echo "== product build: the same tests =="
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "teacher_forced or closed_loop" 2>&1 | tail -2
