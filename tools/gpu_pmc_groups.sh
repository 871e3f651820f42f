#!/bin/bash
# Arbitrary PMC counter groups on the bench command, one rocprofv3 pass per group (counters only, no other trace domain).
#   bash tools/gpu_pmc_groups.sh <tag> "<bench args>" "<group 1 counters>" ["<group 2 counters>" ...]
# Prints, per counter, the average per launch of the fused / demosaic kernel and per pixel-pair wave-iteration (128 px).
set -u
TAG=$1; ARGS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --preheat 60 --no-cpu-baseline --no-secondary $ARGS"
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/g_$i -o pmc -- $P > $OUT/g_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/g_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/g_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "rpf" in r["Kernel_Name"] and "${KERN:-k_stream}" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
it = 24e6 / 128
with open("$OUT/summary.txt", "w") as o:
    for k, v in agg.items():
        line = f"{k:28s} {sum(v)/len(v):16.1f} per launch {sum(v)/len(v)/it:10.2f} per pixel-pair wave-iteration"
        print(line); o.write(line + "\n")
PY
