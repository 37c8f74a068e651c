#!/bin/bash
# Round 6, GPU job 9: the staged triplet kernel visits n - 2 rows per segment (the target's own row skipped instead of masked): parity + training
# tests, the isolated kernel, the bench line.
tag=${1:-r06i}
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | tail -4
python3 tools/bench_triplet.py 2>&1 | tail -3
python3 bench.py --no-secondary > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_line.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['useful_frac'], d['step_roofline']['exec_frac'])"
