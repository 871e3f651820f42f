#!/bin/bash
# First GPU pass: parity tests, smoke, bench lines, rocprof kernel stats.  Stops after a timeout/kill.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/device.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_full_size_6000x4000 > gpurun_out/gpu_tests.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?; tail -2 gpurun_out/smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
for cfg in "develop D 0" "develop S 0" "develop D 1" "develop S 1" "develop D 2" "develop S 2" "develop G 0" "demosaic D 0" "demosaic D 1" "apply D 0" "apply S 0"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload $1 --params $2 --variant $3 --no-cpu-baseline >> gpurun_out/bench_variants.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in bench $cfg - stopping"; exit 1; fi
done
grep -o '"value": [0-9.]*\|"workload": "[^"]*\|"variant": [0-9]\|"frac": [0-9.]*\|"kernel_ms_avg": [0-9.]*' gpurun_out/bench_variants.log | paste - - - - - | sed 's/6000x4000 RGGB u16 -> RGBA32F per GPU, //' | cut -c1-200
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
rc=$?; cat gpurun_out/bench_default.json
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py::test_full_size_6000x4000 -q > gpurun_out/gpu_fullsize.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_fullsize.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_develop -o develop -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_develop.log 2>&1
echo "rocprof rc=$?"
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_develop | head -20
