#!/bin/bash
set -u
TAG=${1:-applyfast}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fast.py tests/test_legacy_flavour.py tests/test_preview.py -m gpu -q -x > $OUT/tests.log 2>&1; rc=$?
tail -4 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL - stopping"; exit 1; fi
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/tests.log | head -20; fi
run() { timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary "$@" 2>>$OUT/err.log | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s MP/s %9.1f  step %.4f ms  kernel avg %.4f min %.4f  frac %.4f' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min'], r['roofline']['frac']))" "$@" | tee -a $OUT/table.txt; }
rm -f $OUT/table.txt
for p in D S G T; do run --workload apply --params $p; run --workload apply --params $p --precision fast; done
run --masks 16
