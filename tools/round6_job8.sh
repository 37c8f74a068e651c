#!/bin/bash
# Round 6, GPU job 8 (final HEAD): race hunts (four lanes vs one stream; pipelined vs plain loop), the bench line, small-batch lines.
tag=${1:-r06h}
python3 tools/stress_bits.py > gpurun_out/${tag}_stress_bits.txt 2>&1; tail -2 gpurun_out/${tag}_stress_bits.txt
python3 tools/stress_pipeline.py 200 4 > gpurun_out/${tag}_stress_pipeline.txt 2>&1; tail -2 gpurun_out/${tag}_stress_pipeline.txt
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
for g in 64 32 16 8; do python3 bench.py --graphs $g --no-cpu-baseline > gpurun_out/${tag}_graphs$g.json 2>> gpurun_out/${tag}_bench_line.err; done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/${tag}_bench_line.json gpurun_out/${tag}_graphs*.json
python3 tools/check_schedule.py > gpurun_out/${tag}_check_schedule.txt 2>&1; tail -1 gpurun_out/${tag}_check_schedule.txt
