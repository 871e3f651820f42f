#!/bin/bash
# Dynamic VALU instruction mix of the fused kernel (rocprofv3 --pmc, two passes).  $1 = tag, $2 = params set (D)
set -u
TAG=${1:-mix}; PSET=${2:-D}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P="python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --preheat 60 --no-cpu-baseline --no-secondary --params $PSET"
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/mix_$i -o pmc -- $P > $OUT/mix_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/mix_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/mix_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "rpf" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
it = 24e6 / 128
for k, v in agg.items():
    print(f"{k:28s} {sum(v)/len(v)/it:8.1f} per pixel-pair wave-iteration")
PY
