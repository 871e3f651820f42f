#!/bin/bash
# Same-box A/B of two (or more) builds of librpf_hip.so: interleaved bench runs, kernel ms (HIP events) per build.
#   bash tools/ab_bench.sh "D S G" tmp_libs/librpf_a.so tmp_libs/librpf_b.so      (paths relative to the repo root)
# Optional: AB_ARGS="--workload apply" adds bench arguments; AB_TEST="-k expr" runs that subset of the GPU tests against the LAST library first.
cd $GRAFT_REPO_ROOT
SETS=$1; shift
LAST=${@: -1}
if [ -n "${AB_TEST:-}" ]; then RPF_HIP_LIB=$GRAFT_REPO_ROOT/$LAST timeout -k 10 600 python -m pytest tests -m gpu -q -x $AB_TEST 2>&1 | tail -2; fi
for rep in 1 2; do
  for lib in "$@"; do
    for p in $SETS; do
      RPF_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --steps 200 --warmup 20 --params $p --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$p', r['value'], r['roofline']['kernel_ms_avg'], r['config']['kernel_ms_min'])"
    done
  done
done
