#!/bin/bash
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "0 0 0" "16384 4096 4096" "16384 8192 8192" "16384 16384 16384" "12288 24576 24576" "20480 2048 2048"; do
  set -- $cfg
  (cd /tmp && JDA_EXP_WALK_LDS=$1 JDA_EXP_FUSED_LDS=$2 JDA_EXP_WRITE_LDS=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o occ -- python $R/tools/pipeline_bench.py --depth 1 --threads 8 --batches 4 > /dev/null 2>&1)
  echo "== extra LDS spec $1 fused $2 write $3"
  python - <<PY
import csv
for r in csv.DictReader(open("$R/$out/occ_kernel_stats.csv")):
    if "segscan" in r["Name"] and "sums" not in r["Name"]: print("   %-62s calls %4s avg %9.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
rows = [r for r in csv.DictReader(open("$R/$out/occ_kernel_trace.csv")) if "fused<3>" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
print("   fused<3> longest launches (us):", [round(x) for x in d[-9:]])
PY
done
