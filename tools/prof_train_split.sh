#!/bin/bash
# Kernel table of a training step with BOTH adjoints in the two-pass form (PG_BWD_SPLIT=2): which pass costs what.  On the GPU box.
out=gpurun_out/prof_split; rm -rf $out; mkdir -p $out
export PHOREGEN_DEBUG=1 PG_BWD_SPLIT=2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_train.py --steps 3 --warmup 1 > $out/log.txt 2>&1 < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -40 "$f" | cut -c1-230 > gpurun_out/train_split_kernel_stats.txt; else echo "no stats file"; tail -5 $out/log.txt; fi
find $out -name "*_trace.csv" -delete
find $out -name "*.db" -delete
du -sh $out
