#!/bin/bash
# Kernel time of the fused develop per edit set and data distribution on ONE box (profiles/r03_sets.txt): the cost of each
# stage on top of the pristine chain (D), on uniform-random (U, worst case for gathers) and natural-like (N) data.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for d in U N; do
  for p in D V T B C S G; do
    timeout -k 10 300 python bench.py --steps 200 --warmup 20 --params $p --data $d --no-cpu-baseline --no-secondary ${SETS_ARGS:-} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'set $p', 'data $d', 'MP/s', r['value'], 'kernel_ms', r['roofline']['kernel_ms_avg'], 'min', r['config']['kernel_ms_min'], 'frac', r['roofline']['frac'])"
  done
done
done
