"""Kernel timeline of ONE sampler step from a rocprofv3 --kernel-trace csv (argv: dir [step index from the end, default 3]).
Prints start (us, relative to the step's first kernel), duration, queue and kernel name; the step = from one pg::embed_ctx_kernel
to the next."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'embed_ctx_kernel' in r['Kernel_Name']]
a, b = marks[-back - 1], marks[-back]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
queues = {}
print(f'# one sampler step: {len(step)} kernels, span {(max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3:.0f} us')
print('# start_us  dur_us  queue  kernel')
for r in step:
    q = queues.setdefault(r['Queue_Id'], 'q%d' % (len(queues) + 1))
    name = re.sub(r'\(.*', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    print('%9.1f %8.1f %s   %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, q, name[:90]))
