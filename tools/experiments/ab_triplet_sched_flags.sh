# The triplet kernel built with alternative LLVM scheduler settings (objects linked into phoregen_amd/_lib_var/<name>/, see tools/experiments/README.md):
# the sub-layer alone (tools/bench_triplet.py), then the headline step for the candidates given as arguments.   GPU box.
for rep in 1 2; do
for v in default $(ls phoregen_amd/_lib_var); do
  if [ $v = default ]; then unset PHOREGEN_HIP_LIB; else export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=phoregen_amd/_lib_var/$v/libphoregen_hip.so; fi
  echo "== $v: $(python tools/bench_triplet.py 30 2>/dev/null | tail -1 | cut -c1-90)"
done; done
for rep in 1 2 3; do
for v in default "$@"; do
  if [ $v = default ]; then unset PHOREGEN_HIP_LIB; else export PHOREGEN_DEBUG=1 PHOREGEN_HIP_LIB=phoregen_amd/_lib_var/$v/libphoregen_hip.so; fi
  echo "== step, $v: $(python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])')"
done; done
