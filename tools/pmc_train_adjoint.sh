#!/bin/bash
# PMC passes over a short training benchmark for the dominant kernel of config 5, the triplet adjoint (seg_attn_bwd_kernel<4, ...>):
# MFMA / VALU instruction counts and HBM-side bytes per launch -> gpurun_out/<tag>_train_adjoint_mfma.json (copied to
# profiles/train_adjoint_mfma.json, which tools/bench_train.py reads for its `roofline` object).   usage: tools/pmc_train_adjoint.sh <tag>
: ${GRAFT_REPO_ROOT:?run through gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_${tag}_trainadj
mkdir -p $out
i=0
for set in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 tools/bench_train.py --steps 1 --warmup 1 > $out/log$i.txt 2>&1
done
python3 - $out $tag <<'PY'
import csv, glob, json, subprocess, sys, collections
root, tag = sys.argv[1], sys.argv[2]
pats = ('seg_attn_bwd_kernel<4', 'triplet_bwd2_kernel')
# the adjoint is one launch of the C ABI and one (both MLP paths in a wave) or two (value pass + key pass) kernels: per kernel name the
# mean over its dispatches, then the sum over the names = per adjoint launch
acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
short = lambda n: n.split('(')[0].replace('void ', '')
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(q in r['Kernel_Name'] for q in pats):
            acc[r['Counter_Name']][short(r['Kernel_Name'])].append(float(r['Counter_Value']))
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(q in r['Kernel_Name'] for q in pats):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
avg = {k: sum(sum(v) / len(v) for v in by.values()) for k, by in acc.items()}
name = ' + '.join(sorted(dur))
per_kernel_us = {k: sum(v) / len(v) for k, v in dur.items()}
dur = [sum(per_kernel_us.values())] * max(len(v) for v in dur.values())
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from bench_train import train_workload
_, na = train_workload(256, seed=4321)
rec = {'kernel': name, 'mfma_per_launch': avg.get('SQ_INSTS_MFMA'), 'valu_per_launch': avg.get('SQ_INSTS_VALU'),
       'padded_rows': int((na * (na - 1) * ((na + 15) // 16 * 16)).sum()),
       'hbm_bytes_per_launch': avg.get('FETCH_SIZE', 0) * 1024 * 2 + avg.get('WRITE_SIZE', 0) * 1024,
       'fetch_bytes_x2': avg.get('FETCH_SIZE', 0) * 1024 * 2, 'write_bytes': avg.get('WRITE_SIZE', 0) * 1024,
       'avg_launch_us_under_pmc': sum(dur) / max(len(dur), 1), 'dispatches': len(dur), 'avg_us_by_kernel_under_pmc': per_kernel_us,
       'commit': subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None,
       'workload': 'BASELINE configs[4] shape: 256 synthetic ligand-phore pairs (tools/bench_train.py, seed 4321)',
       'method': 'rocprofv3 --kernel-trace --pmc in separate passes (SQ_INSTS_MFMA ... / FETCH_SIZE / WRITE_SIZE); FETCH_SIZE KiB x 2 '
                 '(gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE KiB x 1; every MFMA of the kernel is a v_mfma_f32_16x16x4_f32 (2 048 FLOP)'}
json.dump(rec, open(f'gpurun_out/{tag}_train_adjoint_mfma.json', 'w'), indent=1)
print(json.dumps(rec))
PY
rm -rf $out/p*
