#!/bin/bash
# how do the speculative rounds react to fewer workgroups per CU?  tools/gpu_occ.sh tag
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for extra in 0 4096 10240 16384 24576 65536; do
  (cd /tmp && JDA_EXP_WALK_LDS=$extra timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o occ -- python $R/tools/pipeline_bench.py --depth 1 --threads 8 --batches 4 > /dev/null 2>&1)
  echo "== extra LDS $extra"
  python - <<PY
import csv
for r in csv.DictReader(open("$R/$out/occ_kernel_stats.csv")):
    if "fused" in r["Name"]: print("   %-62s calls %4s avg %9.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
rows = [r for r in csv.DictReader(open("$R/$out/occ_kernel_trace.csv")) if "fused<3>" in r["Kernel_Name"] or "fusedILi3" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
print("   fused<3> longest launches (us):", [round(x) for x in d[-9:]])
PY
done
