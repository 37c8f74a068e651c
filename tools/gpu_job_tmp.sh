cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "variants or golden or mid_size or larger or max_size" > gpurun_out/r04_t.txt 2>&1; grep -E "passed|failed|Fatal|assert" gpurun_out/r04_t.txt | head -5
export PHOREGEN_DEBUG=1
for v in 0 1000000; do
  PG_NODE_TILED_BELOW=$v PG_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_pt_$v -- python3 bench.py --graphs 16 --no-cpu-baseline --steps 12 --repeats 1 > /dev/null 2> gpurun_out/r04_pt_$v.log
  echo "== PG_NODE_TILED_BELOW=$v (one stream, 16 graphs)"
  python3 - gpurun_out/r04_pt_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'node' in r['Name'] and 'attn' in r['Name'] or 'knn_node' in r['Name']:
        print('%-70s calls %4s avg %7.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf gpurun_out/r04_pt_$v
done
unset PHOREGEN_DEBUG
for v in base "node_tiled_below=700" "node_tiled_below=10**9" "node_tiled_below=10**9,tri_grid=208"; do python tools/bench_variants.py 16,32 "$v" 2>&1 | grep -v amdgpu.ids; done
