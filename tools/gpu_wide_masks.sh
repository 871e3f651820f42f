#!/bin/bash
# Editors with more masks than one coverage word holds (rpf_chain.h WideMasks, k_apply<SINGLE = 7>): parity subset, then the
# RGBA pass's kernel time by number of masks on both sides of the 16-mask boundary.  $1 = tag
set -u
TAG=${1:-wide}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mask" > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL - stopping"; exit 1; fi
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/tests.log | head -20; fi
run() { timeout -k 10 300 python bench.py --steps 50 --warmup 5 --preheat 50 --no-cpu-baseline --no-secondary "$@" 2>>$OUT/err.log | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s MP/s %9.1f  step %.4f ms  kernel avg %.4f min %.4f' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min']))" "$@" | tee -a $OUT/table.txt; }
rm -f $OUT/table.txt
for m in 8 15 16 17 24 32 48; do run --workload apply --masks $m; done
