#!/bin/bash
# SQ counters of the pre-scan's kernels (one rocprofv3 --pmc pass per group, no trace domains): pipeline at depth 1, 2 batches of 64
out=gpurun_out/pmc_walk; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_THREAD_CYCLES_VALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1)); (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out -o g$i -- python $R/tools/pipeline_bench.py --depth 1 --batches 2 --distinct 16 > $R/$out/g$i.txt 2>&1)
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/walk_sq_counters.txt", "w") as o:
    for k in sorted(acc):
        if not any(x in k for x in ("segscan", "filter", "decode")): continue
        o.write(k + "\n")
        for c, v in sorted(acc[k].items()):
            o.write("    %-30s %.4g (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open("$out/walk_sq_counters.txt").read())
PY
