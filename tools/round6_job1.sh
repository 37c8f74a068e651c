#!/bin/bash
# Round 6, first GPU job: the suite at HEAD (order-point fix of the pipelined v2 schedule), the bench line, the pipelined-loop race hunt,
# the scaling prediction from real shares and the HIP half of the 32-graph x 1000-step match-rate record.
tag=${1:-r06a}
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python3 tools/stress_pipeline.py > gpurun_out/${tag}_stress_pipeline.txt 2>&1
python3 tools/predict_scaling.py > gpurun_out/${tag}_predicted_scaling.txt 2>&1
python3 tools/match_rate.py hip 1000 32 > gpurun_out/${tag}_match_rate_hip.json 2> gpurun_out/${tag}_match_rate_hip.err
tail -3 gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_bench_line.json | cut -c1-400; tail -2 gpurun_out/${tag}_stress_pipeline.txt
grep predicted gpurun_out/${tag}_predicted_scaling.txt; cut -c1-600 gpurun_out/${tag}_match_rate_hip.json; tail -3 gpurun_out/${tag}_match_rate_hip.err
