#!/bin/bash
# timing-only ablations of pg_seg_attn_bwd (PG_BWD_ABLATE bit mask; results are wrong when non-zero); prints the average
# launch time of the training kernels matching KERNELS (default: the triplet and knn-node adjoints)
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in ${@:-0 1 2 4 8 16 32 64 96}; do
  export PG_BWD_ABLATE=$a
  rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/abl$a -o t -- python3 tools/bench_train.py --steps 1 --warmup 0 > gpurun_out/abl$a.log 2>&1
  python3 - <<PY
import csv,glob,os
pats=os.environ.get('KERNELS','seg_attn_bwd_kernel<4;seg_attn_bwd_kernel<0').split(';')
f=glob.glob('gpurun_out/abl$a/**/t_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(p in r['Name'] for p in pats):
        print('ablate=$a', r['Name'][:44], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'total ms', round(int(r['TotalDurationNs'])/1e6,1))
PY
  rm -rf gpurun_out/abl$a gpurun_out/abl$a.log
done
