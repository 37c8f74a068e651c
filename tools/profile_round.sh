#!/bin/bash
# Round profile set (run on the GPU box through gpurun): kernel-trace of the default bench (streams + serial), PMC passes of
# the dominant kernel with FETCH/WRITE, the summaries land in gpurun_out/<tag>_* and are then copied into profiles/ by
# tools/save_profile.py.   usage: tools/profile_round.sh <tag> <kernel-pattern>
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
tag=${1:-r03}; pat=${2:-triplet2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in streams serial; do
  if [ $mode = serial ]; then export PHOREGEN_DEBUG=1 PG_STREAMS=0; else unset PG_STREAMS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$mode -- python3 bench.py --no-cpu-baseline --steps 20 --repeats 1 > gpurun_out/${tag}_bench_$mode.json 2> gpurun_out/${tag}_prof_$mode.log
  # (the summary before the trace is deleted: it takes the span of the overlapping triplet launches from the timestamps)
  python3 tools/save_profile.py stats gpurun_out/${tag}_prof_$mode gpurun_out/${tag}_bench_${mode}_kernel_stats.md "$tag: kernel trace of the default bench ($mode)" "$( [ $mode = serial ] && echo 'PHOREGEN_DEBUG=1 PG_STREAMS=0 ' )rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --repeats 1" 25 $pat
  find gpurun_out/${tag}_prof_$mode -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
unset PG_STREAMS
bash tools/pmc_triplet.sh ${tag}_tri $pat > gpurun_out/${tag}_tri_pmc.txt 2>&1
python3 tools/save_profile.py pmc gpurun_out/pmc_${tag}_tri $pat gpurun_out/${tag}_triplet_pmc.md gpurun_out/${tag}_triplet_traffic.json "$tag: PMC counters of the triplet kernel (tools/bench_triplet.py, headline workload)"
rm -rf gpurun_out/pmc_${tag}_tri/p*
