"""GPU busy fraction from a rocprofv3 kernel trace: union of kernel intervals / span (after skipping the first `skip` fraction)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)))
t0, t1 = iv[0][0], iv[-1][1]
cut = t0 + (t1 - t0) * skip
iv = [x for x in iv if x[0] >= cut]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print('kernels %d  span %.1f ms  busy %.1f ms  = %.1f %%' % (len(iv), span / 1e6, busy / 1e6, 100.0 * busy / span))
