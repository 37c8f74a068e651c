#!/bin/bash
# Round 6, GPU job 2: suite after the advisor fixes, the bench line with the secondary configurations, the pipelined-loop race hunt.
tag=${1:-r06b}
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
( time python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err ) 2> gpurun_out/${tag}_bench_time.txt
python3 tools/stress_pipeline.py > gpurun_out/${tag}_stress_pipeline.txt 2>&1
tail -3 gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_bench_time.txt; tail -2 gpurun_out/${tag}_stress_pipeline.txt
python3 - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench_line.json'))
print(d['ms_per_step'], d['roofline']['frac'])
for k in ('config2_s','train_ms_per_step','config4_graphs_per_hour','secondary_wall_s'):
    print(k, json.dumps(d.get(k))[:400])
PY
