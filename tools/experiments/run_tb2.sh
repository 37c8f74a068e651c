cd $GRAFT_REPO_ROOT
export PHOREGEN_DEBUG=1
for args in "1,2 0 4" "1,2 1 16" "1,2 2 8"; do echo "== $args"; timeout 200 python3 tools/ab_tri_bwd.py $args 2>&1 | grep -v Warning | grep "ligands\|fault\|Error" | cut -c1-200; done
run() { timeout 300 python3 tools/bench_train.py --steps 6 --warmup 2 $2 > gpurun_out/tb2_x.json 2> gpurun_out/tb2_x.err; python3 -c "
import json; d=json.loads(open('gpurun_out/tb2_x.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for f in 0 2 0 2; do PG_TRI_BWD_FORM=$f run "form $f config5" ""; done
