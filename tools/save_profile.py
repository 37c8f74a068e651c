"""Turn rocprofv3 csv output (gpurun_out/...) into the small tracked summaries under profiles/."""
import csv, glob, json, os, sys, collections

def kernel_stats(src_dir, dst_md, title, cmd, steps, pat=None):
    f = glob.glob(os.path.join(src_dir, '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r['TotalDurationNs']) for r in rows)
    with open(dst_md, 'w') as out:
        out.write(f'# {title}\n\n`{cmd}`\n\n{steps} sampler steps, 128 graphs, 1x MI355X; total kernel time {tot/1e6:.1f} ms '
                  f'= {tot/1e6/steps:.2f} ms/step\n\n| kernel | calls | total ms | avg us | % | ms/step |\n|---|---|---|---|---|---|\n')
        for r in rows[:18]:
            out.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                      f"{float(r['Percentage']):.2f} | {int(r['TotalDurationNs'])/1e6/steps:.2f} |\n")
        # kernels that run BESIDE each other (the triplet sub-layer's two launches on two lanes) overlap in time: their durations do not add.
        # From the trace's timestamps: the union of the overlapping `pat` dispatches = one sub-layer
        tr = glob.glob(os.path.join(src_dir, '**', '*kernel_trace.csv'), recursive=True)
        if tr and pat:
            iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(tr[0])) if pat in r['Kernel_Name'])
            spans = []
            for a, b in iv:
                if spans and a <= spans[-1][1]:
                    spans[-1][1] = max(spans[-1][1], b)
                else:
                    spans.append([a, b])
            wall = [(b - a) / 1e3 for a, b in spans]
            out.write(f'\n`{pat}` dispatches overlap in time (two launches per sub-layer on two lanes since round 6): {len(iv)} dispatches form {len(spans)} '
                      f'spans (first start to last end), mean {sum(wall)/len(wall):.1f} us, median {sorted(wall)[len(wall)//2]:.1f} us per sub-layer -- the figure '
                      f"bench.py's `roofline.avg_launch_ms` measures with HIP events; the per-kernel rows above count the time a kernel's workgroups wait for "
                      f'a free CU as its duration\n')

def pmc(src_dir, pat, dst_md, dst_json, title):
    # the sub-layer may be one kernel or two (the triplet kernel by row tiles of the ligands): per kernel name the mean over its dispatches,
    # then the sum over the names = per sub-layer launch
    short = lambda n: n.split('(')[0].replace('void ', '')
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(src_dir + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']][short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    durs = collections.defaultdict(list)
    for f in glob.glob(src_dir + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                durs[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    avg = {k: sum(sum(v) / len(v) for v in by.values()) for k, by in acc.items()}
    names = sorted(durs)
    us = sum(sum(v) / len(v) for v in durs.values())
    dur = [us] * max(len(v) for v in durs.values())
    joined = ' + '.join(names)
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
    # wide coalesced reads -> doubled; WRITE_SIZE is exact.
    fetch_b = avg.get('FETCH_SIZE', 0) * 1024 * 2
    write_b = avg.get('WRITE_SIZE', 0) * 1024
    with open(dst_md, 'w') as out:
        out.write(f'# {title}\n\nrocprofv3 --kernel-trace --pmc <set> (separate passes), kernel(s) `{joined}`, {len(dur)} dispatches each, avg {us:.1f} us per sub-layer (the sum of the kernels\' durations: counter passes run dispatches one at a time)\n\n'
                  '| counter | avg per dispatch |\n|---|---|\n')
        for k in sorted(avg):
            out.write(f'| {k} | {avg[k]:.0f} |\n')
        out.write(f'\nHBM-side traffic per launch (guide corrections: FETCH_SIZE KiB x2, WRITE_SIZE KiB x1): '
                  f'read {fetch_b/1e6:.0f} MB + write {write_b/1e6:.0f} MB = {(fetch_b+write_b)/1e6:.0f} MB\n')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'GRBM_GUI_ACTIVE' in avg:
            cyc = avg['GRBM_GUI_ACTIVE'] / 8
            out.write(f'\nMFMA pipe busy: {avg["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc*100:.1f} % of SIMD cycles '
                      f'(clock {cyc/us/1e3:.2f} GHz); VALU+MFMA instructions {avg.get("SQ_INSTS_VALU",0)/1e6:.0f} M '
                      f'(MFMA {avg.get("SQ_INSTS_MFMA",0)/1e6:.1f} M)\n')
    if dst_json:
        json.dump({'hbm_bytes_per_launch': fetch_b + write_b, 'read_bytes': fetch_b, 'write_bytes': write_b,
                   'avg_launch_us': us, 'source': os.path.basename(dst_md), 'kernel': joined,
                   'avg_us_by_kernel': {k: sum(v) / len(v) for k, v in durs.items()}}, open(dst_json, 'w'))

if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        kernel_stats(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7] if len(sys.argv) > 7 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 6 else None, sys.argv[-1])
