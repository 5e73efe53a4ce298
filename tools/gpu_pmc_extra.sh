#!/bin/bash
# more SQ counters for the decode kernel (instruction fetch, memory-instruction levels, FIFO stalls); one rocprofv3 run per group
out=gpurun_out/${1:-sqx}
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --batch 16 --no-parity --no-cpu-baseline"
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VSKIPPED" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out -o x_$i -- $B > $out/x_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$out/x_*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "jda_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-32s per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
