#!/bin/bash
# Timing-only ablations of the triplet kernel (results are WRONG when a bit is set): builds the -DPG_ABLATE variant of the
# library into phoregen_amd/_lib_ablate and times tools/bench_triplet.py under each mask.  Run on the GPU box (gpurun).
#   1 no P-row loads   2 no query fold   4 no value unfold   8 no angular features   16 no Q   32 no pass B   64 no pass A
cd ${GRAFT_REPO_ROOT:-.}
make -C phoregen_amd/csrc -j8 EXTRA=-DPG_ABLATE OUT=../_lib_ablate > /dev/null || exit 1
export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=$PWD/phoregen_amd/_lib_ablate/libphoregen_hip.so
for m in ${@:-0 1 2 4 8 16 6 22 23 31 32 64 96}; do
  echo -n "ablate=$m: "; PG_TRI_ABLATE=$m python3 tools/bench_triplet.py 10 | tail -1
done
