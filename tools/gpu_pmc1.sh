#!/bin/bash
# One PMC pass (instruction counts) for a bench configuration: $1 tag, rest = bench args
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.OrderedDict()
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rpf' in r['Kernel_Name']: agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
print({k:round(sum(v)/len(v)) for k,v in agg.items()})
PY
