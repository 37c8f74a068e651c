python tools/bench_variants.py 4,12,20 base "knn_merge=\"never\"" 2>&1 | grep G=
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or tuning or mid_size" 2>&1 | grep -E "passed|failed"
timeout 900 python3 tools/predict_scaling.py by_size > gpurun_out/r04f_predicted_scaling.txt 2>&1 < /dev/null; tail -5 gpurun_out/r04f_predicted_scaling.txt | cut -c1-260 | head -4
