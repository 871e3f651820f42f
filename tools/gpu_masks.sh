#!/bin/bash
# Launches with user masks: parity subset, then kernel times by number of masks, shipped vs variant 23 (only "main"'s curve in LDS).  $1 = tag
set -u
TAG=${1:-masks}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_shard.py tests/test_unpack.py -m gpu -q -x -k "mask or Mask or sixteen or shard" > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL - stopping"; exit 1; fi
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/tests.log | head -20; fi
run() { timeout -k 10 300 python bench.py --steps 100 --warmup 10 --preheat 200 --no-cpu-baseline --no-secondary "$@" 2>>$OUT/err.log | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s MP/s %9.1f  step %.4f ms  kernel avg %.4f min %.4f' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min']))" "$@" | tee -a $OUT/table.txt; }
rm -f $OUT/table.txt
for m in 2 3 4 8; do run --masks $m; run --masks $m --variant 23; done
run --masks 2 --data N; run --masks 2 --data N --variant 23
run --masks 4 --data N; run --masks 4 --data N --variant 23
