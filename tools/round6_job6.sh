#!/bin/bash
# Round 6, GPU job 6: calibration of the triplet grid moved into begin_sampling + switches removed: whole suite, bench line, schedule fit table.
tag=${1:-r06f}
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
python3 bench.py --no-secondary > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_line.json')); print(d['ms_per_step'], d['roofline']['frac'], d['warmup'])"
python3 tools/fit_schedule.py > gpurun_out/${tag}_schedule_fit.txt 2>&1
tail -14 gpurun_out/${tag}_schedule_fit.txt
python3 tools/predict_scaling.py > gpurun_out/${tag}_predicted_scaling.txt 2>&1
grep predicted gpurun_out/${tag}_predicted_scaling.txt
