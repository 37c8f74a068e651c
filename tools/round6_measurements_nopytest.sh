# Round 6: the whole measurement job at HEAD (suite, bench lines, kernel traces, PMC of the dominant kernel, step traffic / MFMA, timelines, scaling prediction, parity ratios).
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/free_running_1000.jsonl
bash tools/measure_round.sh r06 > gpurun_out/r06_measure.log 2>&1; tail -6 gpurun_out/r06_measure.log | cut -c1-300
bash tools/profile_round.sh r06 triplet2 > gpurun_out/r06_profile.log 2>&1
python3 tools/save_profile.py stats gpurun_out/r06_prof_streams gpurun_out/r06_bench_kernel_stats.md "r06: kernel trace of the default bench (four lanes)" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --repeats 1" 25
python3 tools/save_profile.py stats gpurun_out/r06_prof_serial gpurun_out/r06_bench_serial_kernel_stats.md "r06: kernel trace of the bench on ONE stream (PG_STREAMS=0)" "PHOREGEN_DEBUG=1 PG_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 20 --repeats 1" 25
bash tools/step_traffic.sh r06 > gpurun_out/r06_traffic.log 2>&1; tail -3 gpurun_out/r06_traffic.log
bash tools/step_mfma.sh r06 > gpurun_out/r06_mfma.log 2>&1; tail -3 gpurun_out/r06_mfma.log
bash tools/timelines_round.sh r06 > gpurun_out/r06_timelines.log 2>&1; tail -8 gpurun_out/r06_timelines.log | cut -c1-250
python3 tools/parity_ratio_table.py gpurun_out/parity_ratios.jsonl HEAD > gpurun_out/r06_parity_ratio_table.md 2>/dev/null; tail -2 gpurun_out/r06_parity_ratio_table.md
ls gpurun_out | grep r06_ | head -60
