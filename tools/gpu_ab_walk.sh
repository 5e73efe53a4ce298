#!/bin/bash
# A/B of two library builds on the pipeline's kernels, one batch in flight (no overlap between kernels): tools/gpu_ab_walk.sh tag libA libB ..
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  for extra in "" "--restart-rows 1" "--quality 98"; do
    (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o ${name} -- python $R/tools/pipeline_bench.py --depth 1 --threads 8 --batches 6 $extra 2>&1 | tail -1 > $R/$out/${name}.json)
    echo "== $name $extra"
    python - <<PY
import csv, json
try: d = json.loads(open("$R/$out/${name}.json").read()); print("   %.0f Mpix/s" % d["mpix_s"])
except Exception as e: print("   no json", e)
for r in csv.DictReader(open("$R/$out/${name}_kernel_stats.csv")):
    if "rocclr" in r["Name"]: continue
    print("   %-62s calls %4s avg %9.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
