#!/bin/bash
# Round 6, GPU job 4: Q rows inside the staged triplet kernel, corrected form (the P product stays on the streaming kernel; the source atom's
# half is read per atom in-kernel): sampler parity tests, then the A/B per batch size.
tag=${1:-r06d}
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_pytest.txt
tail -3 gpurun_out/${tag}_pytest.txt
python3 tools/bench_variants.py 8,16,32,64,128 "tri_q_inkernel=False" "tri_q_inkernel=True" > gpurun_out/${tag}_ab_q_inkernel.txt 2>&1
tail -20 gpurun_out/${tag}_ab_q_inkernel.txt
