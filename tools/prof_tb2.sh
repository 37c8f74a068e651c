#!/bin/bash
# builds a profiling copy of the library (section clock counters in the channel-split triplet adjoint, -DPG_TB2_PROF) into
# phoregen_amd/_lib_prof and prints the split over one training step's six launches (GPU box)
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
set -e
cd $GRAFT_REPO_ROOT/phoregen_amd/csrc && make -j8 EXTRA=-DPG_TB2_PROF OUT=../_lib_prof > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=phoregen_amd/_lib_prof/libphoregen_hip.so python3 tools/prof_tb2.py "$@"
