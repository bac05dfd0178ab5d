#!/bin/bash
# usage: tools/pmc.sh <outname> <kernel-substring> -- <opbench args...>   (two PMC passes, averaged per launch)
name=$1; kern=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $root/gpurun_out/$name/p1 -- python3 $root/tools/opbench.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/$name/p2 -- python3 $root/tools/opbench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/$name/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-28s %14.0f  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
