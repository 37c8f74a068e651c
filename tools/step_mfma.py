"""Per-kernel matrix-pipe work per sampler step from the PMC pass of tools/step_mfma.sh (argv: dir, steps).
FLOPs per MFMA instruction: the GEMM kernels issue v_mfma_f32_32x32x2_f32 (4 096), everything else v_mfma_f32_16x16x4_f32 (2 048)."""
import csv, glob, json, re, sys, collections
root, steps = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for f in glob.glob(f'{root}/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:64]
        acc[name][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_INSTS_MFMA':
            calls[name] += 1
dur = collections.defaultdict(float)
for f in glob.glob(f'{root}/p/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:64]
        dur[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
if not acc:
    sys.exit(f'step_mfma: no counter rows under {root}')
rows = []
for name, c in acc.items():
    per = 4096 if 'gemm' in name or 'rows_linear' in name else 2048
    flop = c.get('SQ_INSTS_MFMA', 0) * per / steps
    if flop <= 0:
        continue
    us = dur[name] / steps
    rows.append((flop, name, calls[name] / steps, us, c.get('SQ_INSTS_MFMA', 0) / steps, c.get('SQ_INSTS_VALU', 0) / steps,
                 c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), c.get('GRBM_GUI_ACTIVE', 0)))
other = [r for r in rows if not r[1].startswith('pg::')]       # library GEMMs of the one-time weight packing, not part of a step
rows = sorted((r for r in rows if r[1].startswith('pg::')), reverse=True)
tot_flop = sum(r[0] for r in rows)
tot_us = sum(dur.values()) / steps
print(f'# Matrix-pipe work per sampler step by kernel, from the hardware counters (one stream, rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU '
      f'SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, {steps} steps; kernel times are under the counters and a little longer than in an unprofiled run)\n')
print('| kernel | launches/step | us/step | MFMA instr/step (M) | other VALU instr/step (M) | GFLOP/step | TF/s | of fp32 peak | MFMA pipe busy |\n|---|---|---|---|---|---|---|---|---|')
for flop, name, c, us, nm, nv, busy, gui in rows:
    busy_pct = busy / 1024 / (gui / 8) * 100 if gui else float('nan')
    print(f'| `{name}` | {c:.0f} | {us:.0f} | {nm / 1e6:.2f} | {(nv - nm) / 1e6:.1f} | {flop / 1e9:.1f} | {flop / us / 1e6:.1f} | {flop / us / 1e6 / 157.3:.2f} | {busy_pct:.0f} % |')
print(f'\nall kernels with matrix instructions: {tot_flop / 1e12:.3f} TFLOP per step; all kernels {tot_us / 1e3:.2f} ms per step (one stream, under the counters) '
      f'= {tot_flop / tot_us / 1e6 / 157.3:.2f} of the fp32 MFMA peak')
print('\n(The count from the launch list also contains the query fold / value unfold of the attention kernels, which run on the vector ALU: '
      '2 x 2 x 128 x 128 FLOP per segment / node -- 80 GFLOP per step for the triplet kernel alone; the counters above are matrix instructions only.)')
if other:
    print('\nNot per step: ' + ', '.join(f'`{r[1][:40]}...` x {r[2] * steps:.0f}' for r in other) + ' (rocBLAS inside the one-time weight packing).')
try:
    line = json.loads([l for l in open(f'{root}/bench.json').read().splitlines() if l.startswith('{')][-1])
    ex = line['step_roofline']['flops_executed']
    print(f'\n`bench.executed_flops` (the numerator of `step_roofline.exec_frac`, counted from the launch list): {ex / 1e12:.3f} TFLOP per step '
          f'-> counters / count = {tot_flop / ex:.3f}')
except Exception as e:
    print(f'\n(no bench line: {e})')
