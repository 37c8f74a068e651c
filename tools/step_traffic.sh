#!/bin/bash
# HBM-side traffic per sampler step by kernel: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes as the guide
# prescribes) over a short serial bench, aggregated by tools/step_traffic.py into gpurun_out/<tag>_step_traffic_by_kernel.md
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
set -e
tag=${1:-r03}
steps=8
warmup=2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic_${tag}
export PHOREGEN_DEBUG=1 PG_STREAMS=0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic_${tag}/$c -- python3 bench.py --no-cpu-baseline --steps $steps --warmup $warmup --repeats 1 > gpurun_out/traffic_${tag}/$c.log 2>&1 \
    || { echo "rocprofv3 --pmc $c failed:"; tail -20 gpurun_out/traffic_${tag}/$c.log; exit 1; }
done
# every profiled step counts: warm-up + timed (bench.py --repeats 1 runs exactly steps + warmup sampler steps)
python3 tools/step_traffic.py gpurun_out/traffic_${tag} $((steps + warmup)) > gpurun_out/${tag}_step_traffic_by_kernel.md
rm -rf gpurun_out/traffic_${tag}/FETCH_SIZE gpurun_out/traffic_${tag}/WRITE_SIZE
