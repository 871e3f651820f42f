#!/bin/bash
# Quick GPU pass: parity tests (optionally a subset), then bench lines.  $1 = tag, $2 = pytest -k expr (optional)
set -u
TAG=${1:-q}; KEXPR=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
stop_if_killed() { if [ $1 -eq 124 ] || [ $1 -eq 137 ]; then echo "TIMEOUT/KILL in $2 - stopping"; exit 1; fi; }
if [ -n "$KEXPR" ]; then timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "$KEXPR" > $OUT/gpu_tests.log 2>&1; else timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/gpu_tests.log 2>&1; fi
rc=$?; tail -4 $OUT/gpu_tests.log; stop_if_killed $rc pytest
if [ $rc -ne 0 ]; then grep -E "^E " $OUT/gpu_tests.log | head -20; fi
CFGS=${BENCH_CFGS:-develop:D:0 develop:S:0 develop:G:0 develop:D:1 develop:D:2 develop:S:2 demosaic:D:0 demosaic:D:1 apply:D:0 apply:S:0}
for cfg in $CFGS; do
  IFS=: read w p v <<< "$cfg"
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --workload $w --params $p --variant $v --no-cpu-baseline --no-secondary >> $OUT/bench_variants.log 2>&1; rc=$?
  stop_if_killed $rc "bench $cfg"
done
python - <<PY
import json
for l in open("$OUT/bench_variants.log"):
    if l.startswith("{"):
        r=json.loads(l); print(r["config"]["workload"].split("fused ")[1][:8], r["config"]["workload"].split("edit set ")[1][:1], "v%d"%r["config"]["variant"], "MP/s", r["value"], "kern_ms", r["roofline"]["kernel_ms_avg"], "min", r["config"]["kernel_ms_min"], "frac", r["roofline"]["frac"])
    elif "Error" in l or "error" in l: print(l.strip()[:200])
PY
