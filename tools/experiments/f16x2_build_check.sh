#!/bin/bash
# EXPERIMENT (round 5, not the product): the triplet kernel with the two K = 128 products of a row tile as error-compensated 2-term f16
# splits on the 16-bit matrix pipe (-DPG_T2_F16X2, csrc/triplet2.hip), built into phoregen_amd/_lib_f16x2.  Measures the kernel alone, the
# sampler step, and runs the sampler parity tests against that build; then the same on the product build.  GPU box, repository root:
#   bash tools/experiments/f16x2_build_check.sh > gpurun_out/r05_f16x2_build_check.txt 2>&1
git apply tools/experiments/triplet2_f16x2_split.patch && make -C phoregen_amd/csrc -j8 OUT=../_lib_f16x2 EXTRA=-DPG_T2_F16X2 > /dev/null; rc=$?; git apply -R tools/experiments/triplet2_f16x2_split.patch; [ $rc = 0 ] || exit 1
X="PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=$PWD/phoregen_amd/_lib_f16x2/libphoregen_hip.so"
for rep in 1 2; do
  echo "== triplet kernel alone, headline batch (tools/bench_triplet.py 30): product build, then the f16x2 build =="
  python3 tools/bench_triplet.py 30 2>/dev/null | tail -1
  env $X python3 tools/bench_triplet.py 30 2>/dev/null | tail -1
done
for g in 128 16; do
  echo "== sampler step, $g graphs (bench.py --no-cpu-baseline): product, f16x2 =="
  python3 bench.py --graphs $g --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo
  env $X python3 bench.py --graphs $g --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo
done
rm -f gpurun_out/parity_ratios.jsonl
echo "== f16x2 build: forward / sampler parity tests against the reference goldens and the oracle =="
env $X python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "forward_against or teacher_forced or closed_loop or free_running or row_tile or max_size or e3_equivariance or full_size_batch" 2>&1 \
  | grep -E "passed|failed|^FAILED|AssertionError|median\(err" | cut -c1-300
echo "== f16x2 build: parity ratio table (err / floor per fixture step) =="
python3 tools/parity_ratio_table.py gpurun_out/parity_ratios.jsonl f16x2 2>/dev/null | tail -3
