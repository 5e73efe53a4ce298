#!/bin/bash
# how the walks' time depends on the workgroups per CU (LDS padding): round 0 (JDA_WALK_LDS_R0) and the counting rounds (JDA_WALK_LDS_R1)
out=gpurun_out/r3_u; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 0" "8192 0" "21504 0" "8192 8192" "8192 21504" "8192 49152"; do
  set -- $cfg
  (cd /tmp && JDA_WALK_LDS_R0=$1 JDA_WALK_LDS_R1=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o r0_$1_r1_$2 -- python $R/tools/pipeline_bench.py --depth 1 --batches 6 --distinct 16 > /dev/null 2>&1)
done
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/*kernel_stats.csv")):
    d = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
    print("%-40s r0 %6.1f  counting rounds (avg of 3) %6.1f  tail %6.1f" % (f.split("/")[-1].replace("_kernel_stats.csv", ""), d.get("jda_segscan_fused<0>", 0), d.get("jda_segscan_fused<3>", 0), d.get("jda_segscan_tail", 0)))
PY
