#!/bin/bash
# timing-only ablations (PG_BWD_ABLATE bit mask, -DPG_ABLATE build; results are wrong when non-zero) of the two-pass adjoints:
# average launch time of the value pass (<mode, 8, true, 1>) and the key pass (<mode, 8, true, 2>).  On the GPU box.
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd $GRAFT_REPO_ROOT/phoregen_amd/csrc && make -j8 EXTRA=-DPG_ABLATE OUT=../_lib_ablate > /dev/null 2>&1 < /dev/null || { echo build failed; exit 1; }
cd $GRAFT_REPO_ROOT
export PHOREGEN_DEBUG=1 PG_BWD_SPLIT=2 PHOREGEN_HIP_LIB=phoregen_amd/_lib_ablate/libphoregen_hip.so
for a in ${@:-0 1 2 4 256 512}; do
  export PG_BWD_ABLATE=$a
  rm -rf gpurun_out/abl$a
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$a -- python3 tools/bench_train.py --steps 1 --warmup 0 > gpurun_out/abl$a.log 2>&1 < /dev/null
  python3 - <<PY
import csv,glob
fs=glob.glob('gpurun_out/abl$a/**/*kernel_stats.csv',recursive=True)
if not fs: print('ablate=$a: no stats')
else:
    for r in csv.DictReader(open(fs[0])):
        if 'seg_attn_bwd_kernel<4, 8' in r['Name'] or 'seg_attn_bwd_kernel<0, 8' in r['Name']:
            print('ablate=$a', r['Name'][9:44], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf gpurun_out/abl$a gpurun_out/abl$a.log
done
