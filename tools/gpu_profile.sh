#!/bin/bash
# The round's profile pass on ONE box: the default bench command, then the same command (without its CPU leg and `secondary`
# block) under rocprofv3 --kernel-trace --stats for the exact and the fast kernel and edit set S, then counter passes.
#   bash tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>/ ; python tools/summarize_profiles.py <tag> <label> condenses it
set -u
TAG=${1:-r04}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "TIMEOUT/KILL in $2 - stopping"; exit 1; fi; }
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?; stop_if_killed $rc bench_default
python -c "
import json; r=json.load(open('$OUT/bench_default.json')); print('bench default: value', r['value'], 'ms/step', r['ms_per_step'], 'kernel avg', r['roofline']['kernel_ms_avg'], 'frac', r['roofline']['frac']); s=r.get('secondary',{}); print({k:(v.get('kernel_ms_avg'), v.get('frac')) for k,v in s.items() if isinstance(v,dict) and 'frac' in v})"
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary"   # the default command: preheat 1000, warmup 50, steps 400
for cfg in "develop_D:" "develop_D_fast:--precision fast" "develop_S:--params S" "develop_S_fast:--params S --precision fast"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- $B $args > $OUT/rocprof_$name.log 2>&1; rc=$?; stop_if_killed $rc "rocprof $name"
  grep "^{" $OUT/rocprof_$name.log | tail -1 > $OUT/bench_under_rocprof_$name.json
done
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --preheat 60 --no-cpu-baseline --no-secondary"   # counter passes: fewer launches
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for w in "develop:" "developfast:--precision fast"; do
    name=${w%%:*}; args=${w#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_${name}_$i -o pmc -- $P $args > $OUT/rocprof_pmc_${name}_$i.log 2>&1; rc=$?; stop_if_killed $rc "pmc $name $i"
  done
done
# input that really comes from HBM: the pool of 8 frames (configs[4]'s access pattern), FETCH_SIZE / WRITE_SIZE
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_batch_$pmc -o pmc -- $P --workload batch --frames 8 --no-prefetch > $OUT/rocprof_pmc_batch_$pmc.log 2>&1; rc=$?; stop_if_killed $rc "pmc batch"
done
ls $OUT | head -60
