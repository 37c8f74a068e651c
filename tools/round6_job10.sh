#!/bin/bash
# Round 6, GPU job 10 (experiment): the tail of the triplet queue in half / quarter groups.  (A record: the knob PG_TRI_TAIL_PARTS existed in plan.py
# only for this experiment; the product rule -- the last 256 entries halved -- came out of it: profiles/r06_triplet_queue_tail.txt.)
export PHOREGEN_DEBUG=1
for k in 0 256 384 256,256 512,256 512,512 1024,512; do
  echo "PG_TRI_TAIL_PARTS=$k: $(PG_TRI_TAIL_PARTS=$k python3 tools/bench_triplet.py 40 2>&1 | tail -1)"
done
for k in 0 256 512,256 1024,512 0 256 512,256 1024,512; do
  echo "PG_TRI_TAIL_PARTS=$k: $(PG_TRI_TAIL_PARTS=$k python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
done
for g in 16 64; do for k in 0 256 512,256 0 256 512,256; do
  echo "G=$g PG_TRI_TAIL_PARTS=$k: $(PG_TRI_TAIL_PARTS=$k python3 bench.py --graphs $g --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done; done
