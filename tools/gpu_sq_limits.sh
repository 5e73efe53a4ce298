#!/bin/bash
# What stands behind the VALU port in the metric kernel: LDS-array cycles and conflicts, FIFO stalls, store path, instruction fetch
# (rocprofv3 --pmc in passes of four counters; bench.py --steps 2 --warmup 1 --batch 16) -> gpurun_out/sq_limits/summary.txt
#   gpurun -- 'bash tools/gpu_sq_limits.sh [extra bench args]'
out=gpurun_out/sq_limits; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1)); (cd /tmp && timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o g$i -- python $R/bench.py --steps 2 --warmup 1 --batch 16 --ramp-ms 0 --no-parity --no-cpu-baseline --e2e-batches 0 --no-configs "$@" > /dev/null 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$out/g*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "jda_decode_tiles_persistent" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as o:
    for k, v in sorted(acc.items()):
        o.write("%-30s %.6g\n" % (k, sum(v) / len(v)))
print(open("$out/summary.txt").read())
PY
