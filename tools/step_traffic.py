"""Per-kernel HBM-side traffic per sampler step from the two PMC passes of tools/step_traffic.sh (argv: dir, steps)."""
import csv, glob, sys, collections, re
root, steps = sys.argv[1], int(sys.argv[2])
tot = {c: collections.defaultdict(float) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
calls = collections.defaultdict(int)
for c in tot:
    for f in glob.glob(f'{root}/{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            name = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
            tot[c][name] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                calls[name] += 1
rows = []
for name in set(tot['FETCH_SIZE']) | set(tot['WRITE_SIZE']):
    rd = tot['FETCH_SIZE'][name] * 1024 * 2 / steps / 1e9          # KiB, x2 on gfx950 (MI355X_MICROARCH.md)
    wr = tot['WRITE_SIZE'][name] * 1024 / steps / 1e9
    rows.append((rd + wr, name, calls[name] / steps, rd, wr))
rows.sort(reverse=True)
if not rows:
    sys.exit(f'step_traffic: no counter_collection.csv rows under {root} (did rocprofv3 run?)')
print(f'# HBM-side traffic per sampler step by kernel (PG_STREAMS=0, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, '
      f'{steps} steps; FETCH_SIZE KiB x2, WRITE_SIZE KiB x1)\n')
print('| kernel | launches/step | read GB/step | write GB/step | total GB/step |\n|---|---|---|---|---|')
for t, name, c, rd, wr in rows[:16]:
    print(f'| {name} | {c:.0f} | {rd:.2f} | {wr:.2f} | {t:.2f} |')
print(f'\nall kernels: {sum(r[0] for r in rows):.1f} GB per step')
