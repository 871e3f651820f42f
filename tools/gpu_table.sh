#!/bin/bash
# Every line of DESIGN.md's "Measured" table on ONE box: bash tools/gpu_table.sh <tag>
set -u
TAG=${1:-table}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s MP/s %9.1f  step %.4f ms  kernel avg %.4f min %.4f  frac %.4f' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min'], r['roofline']['frac']))" "$@" | tee -a $OUT/table.txt; }
rm -f $OUT/table.txt
run --workload demosaic
for p in D S G V T; do run --workload develop --params $p; done
for p in D S G V T; do run --workload develop --params $p --precision fast; done   # RPF_PRECISION_FAST specialisation (round 4)
run --workload develop --data N
run --workload develop --data N --precision fast
run --workload develop_packed
run --workload develop_packed --pack mipi
run --workload unpack
run --workload apply --params D
run --workload apply --params S
run --workload apply --params G
run --workload develop --out rgb
run --workload develop --out u8
for m in 2 4 8 16; do run --workload develop --masks $m; done
run --workload batch --frames 16
run --workload batch --frames 16 --no-prefetch
