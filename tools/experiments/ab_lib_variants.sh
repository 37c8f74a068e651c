# Library variants under phoregen_amd/_lib_var/<name>/ against the product build, alternating: the headline step for the variants whose name does not
# start with tb_ / sb_ (adjoint files), the config-5 training step for those.  GPU box.   (see tools/experiments/README.md for how the variants are linked)
for rep in 1 2; do
for v in default $(ls phoregen_amd/_lib_var); do
  if [ $v = default ]; then unset PHOREGEN_HIP_LIB; else export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=phoregen_amd/_lib_var/$v/libphoregen_hip.so; fi
  case $v in tb_*|sb_*) ;; *) echo "== step, $v: $(python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_knn_node"]["avg_sublayer_ms"])')";; esac
  case $v in tb_*|sb_*|default) echo "== train, $v: $(python tools/bench_train.py --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')";; esac
done; done
