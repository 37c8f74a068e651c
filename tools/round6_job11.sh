# Round 6, after the triplet launches went side by side: the whole measurement set again at HEAD.
tag=r06
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/free_running_1000.jsonl
python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_full.log 2>&1; grep -E "passed|failed" gpurun_out/${tag}_pytest_full.log | tail -2
python3 tools/check_schedule.py > gpurun_out/${tag}_check_schedule.txt 2>&1; tail -1 gpurun_out/${tag}_check_schedule.txt
bash tools/measure_round.sh $tag > gpurun_out/${tag}_measure.log 2>&1; tail -6 gpurun_out/${tag}_measure.log | cut -c1-300
bash tools/profile_round.sh $tag triplet2 > gpurun_out/${tag}_profile.log 2>&1
python3 tools/experiments/ab_tri_overlap.py > gpurun_out/${tag}_triplet_side_by_side.txt 2>&1
python3 tools/fit_schedule.py > gpurun_out/${tag}_schedule_fit.txt 2>&1; tail -12 gpurun_out/${tag}_schedule_fit.txt | cut -c1-250
bash tools/timelines_round.sh $tag > gpurun_out/${tag}_timelines.log 2>&1; tail -8 gpurun_out/${tag}_timelines.log | cut -c1-250
python3 tools/parity_ratio_table.py gpurun_out/parity_ratios.jsonl HEAD > gpurun_out/${tag}_parity_ratio_table.md 2>/dev/null; tail -2 gpurun_out/${tag}_parity_ratio_table.md
