"""Average PMC counters per dispatch of the kernels whose name contains argv[2], from rocprofv3 csv output dirs."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f'kernel ~{pat}: {len(dur)} dispatches, avg {sum(dur)/max(len(dur),1):.1f} us')
for k in sorted(acc):
    v = acc[k]
    print(f'{k:28s} avg {sum(v)/len(v):16.0f}   (n={len(v)})')
