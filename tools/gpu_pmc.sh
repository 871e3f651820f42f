#!/bin/bash
# GPU pass: all parity tests, bench lines, rocprofv3 kernel stats and PMC passes.  $1 = tag for output names.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "TIMEOUT/KILL in $2 - stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; rc=$?
echo "pytest rc=$rc" >> $OUT/gpu_tests.log; tail -4 $OUT/gpu_tests.log; stop_if_killed $rc pytest
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -1 $OUT/smoke.log; stop_if_killed $rc smoke
for cfg in "develop D 0" "develop S 0" "develop G 0" "develop D 20" "develop S 20" "develop G 20" "demosaic D 0" "apply D 0" "apply S 0"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --workload $1 --params $2 --variant $3 --no-cpu-baseline --no-secondary >> $OUT/bench_variants.log 2>&1; rc=$?
  stop_if_killed $rc "bench $cfg"
done
python - <<PY
import json
for l in open("$OUT/bench_variants.log"):
    if l.startswith("{"):
        r=json.loads(l); print(r["config"]["workload"].split("fused ")[1][:8], r["config"]["workload"].split("edit set ")[1][:1], "v%d"%r["config"]["variant"], "MP/s", r["value"], "kern_ms", r["roofline"]["kernel_ms_avg"], "frac", r["roofline"]["frac"])
PY
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?; cat $OUT/bench_default.json; stop_if_killed $rc bench_default
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --preheat 200 --no-cpu-baseline --no-secondary"   # the default command (steps 400, warmup 50) with 200 instead of 1000 preheat steps, without its CPU leg and without the `secondary` block, whose 340 launches of the SAME kernel on frames from HBM would be averaged into the headline kernel's statistics
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --preheat 60 --no-cpu-baseline --no-secondary"   # counter passes: fewer launches
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_develop_D -o develop_D -- $B > $OUT/rocprof_stats.log 2>&1; rc=$?; stop_if_killed $rc rocprof_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_develop_S -o develop_S -- $B --params S >> $OUT/rocprof_stats.log 2>&1; rc=$?; stop_if_killed $rc rocprof_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_demosaic -o demosaic -- $B --workload demosaic >> $OUT/rocprof_stats.log 2>&1; rc=$?; stop_if_killed $rc rocprof_stats
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for w in "develop D" "demosaic D"; do
    set -- $w
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc_$1_$i -o pmc -- $P --workload $1 --params $2 > $OUT/rocprof_pmc_$1_$i.log 2>&1; rc=$?; stop_if_killed $rc "pmc $i"
  done
done
ls $OUT | head -40
