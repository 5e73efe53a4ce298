#!/bin/bash
# SQ counter passes for the decode kernel (GPU box, through gpurun).  One rocprofv3 run per group,
# no trace domains.  Output: gpurun_out/$1/sq_<n>_counter_collection.csv
tag=${1:-sq}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out -o sq_$i -- $B > $out/sq_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$out/sq_*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
