cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PHOREGEN_DEBUG=1
for v in 0 2; do
  PG_POS_TILED=$v PG_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_pt_$v -- python3 bench.py --graphs 16 --no-cpu-baseline --steps 12 --repeats 1 > /dev/null 2> gpurun_out/r04_pt_$v.log
  echo "== PG_POS_TILED=$v (one stream, 16 graphs)"
  python3 - gpurun_out/r04_pt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'node_attn' in r['Name']:
        print('%-70s calls %4s avg %7.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf gpurun_out/r04_pt_$v
done
