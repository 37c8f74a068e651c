#!/bin/bash
# Round 6, GPU job 3: Q rows inside the staged triplet kernel (options.tri_q_inkernel): parity suite, then A/B per batch size in one process.
tag=${1:-r06c}
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
python3 tools/bench_variants.py 8,16,32,64,128 "tri_q_inkernel=False" "tri_q_inkernel=True" > gpurun_out/${tag}_ab_q_inkernel.txt 2>&1
tail -20 gpurun_out/${tag}_ab_q_inkernel.txt
