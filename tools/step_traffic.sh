#!/bin/bash
# HBM-side traffic per sampler step by kernel: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes as the guide
# prescribes) over a short serial bench, aggregated by tools/step_traffic.py into gpurun_out/<tag>_step_traffic_by_kernel.md
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PG_STREAMS=0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic_${tag}/$c -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --repeats 1 > gpurun_out/traffic_${tag}/$c.log 2>&1
done
python3 tools/step_traffic.py gpurun_out/traffic_${tag} 10 > gpurun_out/${tag}_step_traffic_by_kernel.md
rm -rf gpurun_out/traffic_${tag}/FETCH_SIZE gpurun_out/traffic_${tag}/WRITE_SIZE
