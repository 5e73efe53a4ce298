#!/bin/bash
# kernel times of the pipeline (one batch in flight) under several library builds: tools/gpu_ab_quick.sh tag lib ..
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$R/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o ${name} -- python $R/tools/pipeline_bench.py --depth 1 --threads 8 --batches 4 $AB_ARGS > /dev/null 2>&1)
  echo "== $name"
  python - <<PY
import csv
for r in csv.DictReader(open("$R/$out/${name}_kernel_stats.csv")):
    if "rocclr" in r["Name"]: continue
    print("   %-62s calls %4s avg %9.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
