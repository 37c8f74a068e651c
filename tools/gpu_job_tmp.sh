cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04_tl_16 -- python3 bench.py --graphs 16 --no-cpu-baseline --steps 12 --repeats 1 > gpurun_out/r04_tl_16.json 2> gpurun_out/r04_tl_16.log
python3 tools/timeline.py gpurun_out/r04_tl_16 3 > gpurun_out/r04_timeline_16graphs_b.txt 2>&1
rm -rf gpurun_out/r04_tl_16
sed -n 1,80p gpurun_out/r04_timeline_16graphs_b.txt
