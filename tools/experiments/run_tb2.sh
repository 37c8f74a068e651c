cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python3 bench.py --train > gpurun_out/tb2_train_line.json 2> gpurun_out/tb2_x.err; python3 -c "
import json; d=json.loads(open('gpurun_out/tb2_train_line.json').read().strip().splitlines()[-1]); print(d['value'], d.get('peak_mem_gb'), d.get('roofline'))"
