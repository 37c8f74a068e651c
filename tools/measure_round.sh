#!/bin/bash
# The round's measurement set (run on the GPU box through gpurun): driver-style bench line, the sustained 1000-step line, the
# small-batch lines (one GPU's share of the headline batch on 2 / 4 / 8 GPUs), config 2, the training step; everything lands in
# gpurun_out/<tag>_*.   usage: tools/measure_round.sh <tag>
tag=${1:-r04}
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python3 bench.py --steps 1000 --warmup 0 --repeats 1 --no-cpu-baseline > gpurun_out/${tag}_bench_1000steps.json 2>> gpurun_out/${tag}_bench_line.err
for g in 64 32 16 8; do
  python3 bench.py --graphs $g --no-cpu-baseline > gpurun_out/${tag}_graphs$g.json 2>> gpurun_out/${tag}_bench_line.err
done
python3 tools/bench_config2.py > gpurun_out/${tag}_config2_sample100.json 2>> gpurun_out/${tag}_bench_line.err
python3 tools/bench_train.py --steps 10 --warmup 2 > gpurun_out/${tag}_train_bench_line.json 2>> gpurun_out/${tag}_bench_line.err
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/${tag}_bench_line.json gpurun_out/${tag}_bench_1000steps.json gpurun_out/${tag}_graphs*.json
tail -c 400 gpurun_out/${tag}_config2_sample100.json; tail -c 300 gpurun_out/${tag}_train_bench_line.json
