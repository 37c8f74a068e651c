#!/bin/bash
# Kernel timelines of one sampler step at 16 / 32 / 128 graphs (rocprofv3 --kernel-trace of a short bench, tools/timeline.py) and the
# strong-scaling estimate from the actual rank shares (tools/predict_scaling.py).   usage: tools/timelines_round.sh <tag>   (GPU box)
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
for g in 16 32 128; do
  d=gpurun_out/${tag}_tl_$g; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --graphs $g --no-cpu-baseline --steps 10 --warmup 3 --repeats 1 > /dev/null 2> $d.log < /dev/null
  python3 tools/timeline.py $d 3 > gpurun_out/${tag}_timeline_${g}graphs.txt 2>> $d.log || tail -3 $d.log
  rm -rf $d
done
timeout 600 python3 tools/predict_scaling.py > gpurun_out/${tag}_predicted_scaling.txt 2>&1 < /dev/null
tail -12 gpurun_out/${tag}_predicted_scaling.txt | cut -c1-220
