#!/bin/bash
# Round 6, GPU job 5: P's h_bond part ahead of the coordinates + ONE [P | Q] launch behind them (options.pq_panels): the GEMM form's test, the
# sampler parity file, A/B per batch size in one process, stress.
tag=${1:-r06e}
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
python3 tools/bench_variants.py 8,16,32,64,128 "pq_panels=False" "pq_panels=True" > gpurun_out/${tag}_ab_pq_panels.txt 2>&1
tail -20 gpurun_out/${tag}_ab_pq_panels.txt
