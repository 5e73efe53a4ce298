#!/bin/bash
# kernel time per batch of the streamed pipeline, nothing overlapped (depth 1): tools/gpu_pipeline_stats.sh [prefix]
#   rocprofv3 --kernel-trace --stats of tools/pipeline_bench.py --depth 1 for the metric batch (64 x 4096x4096) and for batches of small
#   images (1024 x 1280x720, 1024 x 1920x1080, 2048 x 640x480) -> gpurun_out/pipe_stats/<prefix>_pipeline_<shape>_kernel_stats_depth1.csv
prefix=${1:-r06}
out=gpurun_out/pipe_stats; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4096 4096 64 16" "1280 720 1024 8" "1920 1080 1024 8" "640 480 2048 8"; do
  set -- $cfg
  tag=${1}x${2}
  (cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o ${tag} -- python $R/tools/pipeline_bench.py --depth 1 --batches 10 --width $1 --height $2 --batch $3 --distinct $4 > $R/$out/${tag}_line.json 2> /dev/null)
  cp $out/${tag}_kernel_stats.csv $out/${prefix}_pipeline_${tag}_kernel_stats_depth1.csv 2>/dev/null
  echo "== $tag (batch $3)"; cat $out/${tag}_line.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth 1:', round(d['mpix_s']), 'Mpix/s', d['stats'])"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$out/${tag}_kernel_stats.csv")))
tot = 0
for r in rows:
    per_batch = float(r["TotalDurationNs"]) / 12.0 / 1e3      # 10 timed + 2 warm-up batches
    tot += per_batch
    print("  %-60s calls %5s  %8.1f us per batch" % (r["Name"][:60], r["Calls"], per_batch))
print("  total %.1f us per batch" % tot)
PY
done
