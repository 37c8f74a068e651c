#!/bin/bash
# Round 6, GPU job 7: embedded features in a buffer of their own (options.embed_buffer): schedule check + tests, parity file, A/B, scaling prediction.
tag=${1:-r06g}
python3 tools/check_schedule.py 2 16 72 128 2>&1 | grep -v amdgpu.ids | tail -10
python3 -m pytest tests/test_gpu_schedule.py tests/test_gpu_parity.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_sampler_free_running_1000_steps_matches_reference 2>&1 | tail -4
python3 tools/bench_variants.py 8,16,32,128 "embed_buffer=False" "embed_buffer=True" > gpurun_out/${tag}_ab_embed_buffer.txt 2>&1
tail -16 gpurun_out/${tag}_ab_embed_buffer.txt
python3 tools/stress_pipeline.py > gpurun_out/${tag}_stress_pipeline.txt 2>&1; tail -2 gpurun_out/${tag}_stress_pipeline.txt
python3 tools/predict_scaling.py > gpurun_out/${tag}_predicted_scaling.txt 2>&1
grep predicted gpurun_out/${tag}_predicted_scaling.txt
