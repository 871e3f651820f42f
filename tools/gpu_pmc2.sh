#!/bin/bash
# Instruction-class breakdown of a bench configuration: $1 tag, rest = bench args
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU2 SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/pmc$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'rpf' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
it=24e6/128
for k,v in agg.items(): print(f"{k:28s} {sum(v)/len(v):14.0f}  per pair-iteration {sum(v)/len(v)/it:8.1f}")
PY
