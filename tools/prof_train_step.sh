#!/bin/bash
# kernel table of a training step (config 5) + PMC record of the triplet adjoint -> gpurun_out/<tag>_train_kernel_stats.md,
# gpurun_out/<tag>_train_adjoint_mfma.json (copied to profiles/).   usage: tools/prof_train_step.sh <tag>
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_train_prof
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_train.py --steps 3 --warmup 1 > gpurun_out/${tag}_train_prof.log 2>&1
python3 - $out gpurun_out/${tag}_train_kernel_stats.md "$tag" <<'PY'
import csv, glob, os, sys
src, dst, tag = sys.argv[1:4]
f = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 6          # bench_train: 1 warm-up + 3 timed + 2 more for the launch timers
tot = sum(int(r['TotalDurationNs']) for r in rows)
with open(dst, 'w') as out:
    out.write(f'# {tag}: kernel trace of a training step (config 5)\n\n`rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py --steps 3 --warmup 1`\n\n'
              f'{steps} training steps in the trace (warm-up + timed steps + 2 more for the launch timers), config-5 batch (256 pairs, 163 346 bond edges), '
              f'1x MI355X; total kernel time {tot/1e6:.1f} ms = {tot/1e6/steps:.1f} ms per step\n\n| kernel | calls / step | ms / step | avg us | % |\n|---|---|---|---|---|\n')
    for r in rows[:40]:
        out.write(f"| `{r['Name'][:110]}` | {int(r['Calls'])/steps:.1f} | {int(r['TotalDurationNs'])/1e6/steps:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
PY
head -24 gpurun_out/${tag}_train_kernel_stats.md | cut -c1-200
bash tools/pmc_train_adjoint.sh $tag | tail -c 1500
