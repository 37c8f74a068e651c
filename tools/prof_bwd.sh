#!/bin/bash
# builds a profiling copy of the library (section cycle counters in pg_seg_attn_bwd) on the GPU box and prints the split
set -e
cd $GRAFT_REPO_ROOT/phoregen_amd/csrc
cp ../_lib/libphoregen_hip.so /tmp/lib_keep.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=on -Wno-unused-variable -Wno-unused-but-set-variable -DPG_BWD_PROF -c seg_attn_bwd.hip -o /tmp/bwd_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../_lib/gemm.o ../_lib/graph_ops.o ../_lib/seg_attn.o ../_lib/triplet.o ../_lib/node_attn.o ../_lib/posterior.o /tmp/bwd_prof.o ../_lib/train_ops.o -o ../_lib/libphoregen_hip.so
cd $GRAFT_REPO_ROOT && python3 tools/prof_bwd.py; cp /tmp/lib_keep.so phoregen_amd/_lib/libphoregen_hip.so
