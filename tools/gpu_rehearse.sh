#!/bin/bash
# One-GPU rehearsal of bench.py's N > 1 code path (self-neighbour RCCL exchange): bash tools/gpu_rehearse.sh <tag>
set -u
TAG=${1:-rehearse}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rm -f $OUT/rehearse.txt
for M in 2 4 8; do
  for S in serial pingpong once; do
    if [ $S = once ]; then SCHED="--halo once"; else SCHED="--schedule $S"; fi
    timeout -k 10 200 python bench.py --steps 400 --warmup 50 --rehearse-ranks $M $SCHED "${@:2}" 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
json.dump(r, open('$OUT/bench_rehearse_${M}_${S}.json','w'))
print('rank %d/%d rows %d schedule %-16s band MP/s %9.1f  step %.4f ms  (x%d = %.0f MP/s if every rank ran like this one)' % ($M//2, $M, r['config']['ranks'][0]['band'][1], r['config']['schedule'], r['value'], r['ms_per_step'], $M, r['value']*$M))" | tee -a $OUT/rehearse.txt
  done
done
