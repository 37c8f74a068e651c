# Library variants under phoregen_amd/_lib_var/<name>/ against the product build: headline step and (with `train`) the config-5 step, alternating.  GPU box.
for rep in 1 2 3; do
for v in default $(ls phoregen_amd/_lib_var); do
  if [ $v = default ]; then unset PHOREGEN_HIP_LIB; else export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=phoregen_amd/_lib_var/$v/libphoregen_hip.so; fi
  echo "== step, $v: $(python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_knn_node"]["avg_sublayer_ms"])')"
  if [ "$1" = train ] && [ $rep -lt 3 ]; then echo "== train, $v: $(python tools/bench_train.py --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')"; fi
done; done
