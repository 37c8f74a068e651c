#!/bin/bash
# Matrix-pipe work per sampler step by kernel, measured: one rocprofv3 PMC pass (SQ_INSTS_MFMA, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES,
# GRBM_GUI_ACTIVE) over a short one-stream bench + its kernel trace -> gpurun_out/<tag>_step_mfma_by_kernel.md.  Cross-checks the executed-FLOP
# counts of bench.executed_flops (step_roofline.exec_frac) against the hardware's own instruction counters.
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
set -e
tag=${1:-r04}
steps=8
warmup=2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mfma_${tag}
export PHOREGEN_DEBUG=1 PG_STREAMS=0
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mfma_${tag}/p -- python3 bench.py --no-cpu-baseline --steps $steps --warmup $warmup --repeats 1 > gpurun_out/mfma_${tag}/bench.json 2> gpurun_out/mfma_${tag}/log.txt \
  || { echo "rocprofv3 failed:"; tail -20 gpurun_out/mfma_${tag}/log.txt; exit 1; }
python3 tools/step_mfma.py gpurun_out/mfma_${tag} $((steps + warmup)) > gpurun_out/${tag}_step_mfma_by_kernel.md
rm -rf gpurun_out/mfma_${tag}/p
