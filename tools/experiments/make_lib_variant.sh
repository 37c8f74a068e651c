#!/bin/bash
# Link a library variant: ONE kernel file compiled with extra flags, every other object from the product build (phoregen_amd/_lib, `make` first).
#   tools/experiments/make_lib_variant.sh <name> <file without .hip> <extra hipcc flags ...>
#   e.g.  make_lib_variant.sh t2_maxilp triplet2 -mllvm -amdgpu-sched-strategy=max-ilp
# -> phoregen_amd/_lib_var/<name>/libphoregen_hip.so, loaded with PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=<that path> (ab_lib_variants.sh, ab_triplet_sched_flags.sh).
# (The product flags of the Makefile for that file are NOT applied: pass them explicitly to vary on top of them.)
set -e
cd "$(dirname "$0")/../../phoregen_amd/csrc"
name=$1; file=$2; shift 2
mkdir -p ../_lib_var/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=on -Wno-unused-variable -Wno-unused-but-set-variable "$@" -c $file.hip -o ../_lib_var/$name/$file.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../_lib/*.o | grep -v "/$file.o") ../_lib_var/$name/$file.o -o ../_lib_var/$name/libphoregen_hip.so
rm ../_lib_var/$name/$file.o
echo "built phoregen_amd/_lib_var/$name/libphoregen_hip.so"
