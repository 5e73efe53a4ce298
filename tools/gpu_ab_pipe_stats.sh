#!/bin/bash
# same-box A/B of library builds on the pipeline's kernel times at depth 1 (metric batch and 1,024 x 1280x720): tools/gpu_ab_pipe_stats.sh ab/lib_x.so ..
out=gpurun_out/ab_pipe; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@" jpegdec_amd/libjpegdec_amd.so; do
  for cfg in "4096 4096 64 16" "1280 720 1024 8"; do
    set -- $cfg
    (cd /tmp && JDA_LIBRARY=$R/$lib timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o s -- python $R/tools/pipeline_bench.py --depth 1 --batches 10 --width $1 --height $2 --batch $3 --distinct $4 > /dev/null 2>&1)
    python - <<PY
import csv
rows = list(csv.DictReader(open("$out/s_kernel_stats.csv")))
sel = {}
for r in rows:
    n = r["Name"]
    for k in ("fused<0", "fused<4, true", "fused<4, false", "tail", "decode_tiles", "finalize", "filter_count", "filter_write"):
        if k in n: sel[k] = float(r["TotalDurationNs"]) / 12e3
print("$lib ${1}x${2}", " ".join("%s %.0f" % (k, v) for k, v in sel.items()), "| total %.0f us" % (sum(float(r["TotalDurationNs"]) for r in rows) / 12e3))
PY
  done
done
