#!/bin/bash
set -u
TAG=${1:-fast2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fast.py -m gpu -q -x > $OUT/gpu_fast_tests.log 2>&1; rc=$?
tail -5 $OUT/gpu_fast_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL - stopping"; exit 1; fi
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/gpu_fast_tests.log | head -20; fi
run() { timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary "$@" 2>>$OUT/err.log | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s MP/s %9.1f  step %.4f ms  kernel avg %.4f min %.4f  frac %.4f' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min'], r['roofline']['frac']))" "$@" | tee -a $OUT/table.txt; }
rm -f $OUT/table.txt
run --params D --precision fast
run --params D --precision fast --variant 30
run --params D --precision fast --variant 31
run --params D
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"
KERN=k_hp bash $GRAFT_REPO_ROOT/tools/gpu_pmc_groups.sh ${TAG}_pmc "--params D --precision fast" "$G1" "$G2" "$G4"
