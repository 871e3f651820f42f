#!/bin/bash
# multi-process shard tests + timing-only A/B of the fast kernel.  $1 = tag
set -u
TAG=${1:-mp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_shard_multiproc.py tests/test_shard.py -m gpu -q -x > $OUT/mp_tests.log 2>&1; rc=$?
tail -6 $OUT/mp_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL - stopping"; exit 1; fi
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/mp_tests.log | head -30; fi
AB_ARGS="--precision fast --no-secondary" bash tools/ab_bench.sh "D" tmp_libs/librpf_base.so tmp_libs/librpf_f_nostore.so tmp_libs/librpf_f_noconf.so tmp_libs/librpf_f_both.so 2>&1 | tee $OUT/ab_fast_timing_only.txt
