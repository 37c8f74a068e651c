#!/bin/bash
# section cycle counters of csrc/triplet2.hip: builds the -DPG_T2_PROF variant into phoregen_amd/_lib_prof (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
make -C phoregen_amd/csrc -j8 EXTRA=-DPG_T2_PROF OUT=../_lib_prof > /dev/null || exit 1
export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=$PWD/phoregen_amd/_lib_prof/libphoregen_hip.so
for d in ${@:-0 4}; do echo "debug mask $d"; PG_SEG_DEBUG=$d python3 tools/prof_triplet2.py; done
