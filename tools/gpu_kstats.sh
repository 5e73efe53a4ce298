#!/bin/bash
# kernel-time table of the pipeline bench for each library given: tools/gpu_kstats.sh tag lib...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for lib in "$@"; do
  b=$(basename $lib .so)
  (cd /tmp && JDA_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o $b -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 1 --threads 8 --batches 4 ${PIPE_ARGS} > $GRAFT_REPO_ROOT/$out/$b.json 2>/dev/null)
  echo "== $b $(python -c "import json;d=json.loads(open('$out/$b.json').read().strip().splitlines()[-1]);print(round(d['mpix_s']),'Mpix/s')")"
  python - <<PY
import csv
for r in csv.DictReader(open("$out/${b}_kernel_stats.csv")):
    if float(r["TotalDurationNs"]) > 2e5: print("   %-64s calls %4s  avg %9.1f us  total %8.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
