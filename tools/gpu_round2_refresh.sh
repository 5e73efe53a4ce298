#!/bin/bash
# Refresh of the round-2 evidence that depends on the pre-scan kernels (the decode kernels' own tables are gpu_round2_profiles.sh's):
#   tools/gpu_round2_refresh.sh [tag] -> gpurun_out/<tag>/ (copied to profiles/r02_* by hand)
tag=${1:-r02_refresh}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $out/pytest_gpu.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o kt -- python $R/bench.py --no-parity --no-cpu-baseline --e2e-batches 0 > /dev/null 2>&1)
: > $out/pipe.jsonl
P="python tools/pipeline_bench.py --threads 8"
for c in "--depth 1" "--depth 2" "--depth 3" "--depth 4" "--depth 3 --distinct 16" "--depth 4 --distinct 16" "--depth 3 --batch 128 --batches 8" "--depth 3 --batch 256 --batches 6" "--depth 4 --batch 256 --batches 6 --distinct 16" "--depth 3 --restart-rows 1" "--depth 3 --restart-rows 4" \
         "--depth 3 --subsampling 4:4:4" "--depth 3 --quality 98 --batches 6" \
         "--depth 3 --width 1920 --height 1080 --batch 256 --batches 8" "--depth 3 --width 1920 --height 1080 --batch 1024 --batches 6" \
         "--depth 3 --width 1280 --height 720 --batch 512 --batches 8" "--depth 3 --width 1280 --height 720 --batch 1024 --batches 8" "--depth 3 --width 1280 --height 720 --batch 2048 --batches 6" \
         "--depth 4 --width 1280 --height 720 --batch 1024 --batches 8 --distinct 64" "--depth 4 --width 1920 --height 1080 --batch 256 --batches 8 --distinct 16"; do
  timeout 300 $P $c 2>/dev/null | tail -1 >> $out/pipe.jsonl
done
kt() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o $name -- python $R/tools/pipeline_bench.py --threads 8 "$@" > /dev/null 2>&1)
}
kt pipe_d1 --depth 1 --batches 6
kt pipe_d3 --depth 3 --batches 8
kt pipe_1080p_d1 --depth 1 --batches 6 --width 1920 --height 1080 --batch 256
kt pipe_720p_d1 --depth 1 --batches 6 --width 1280 --height 720 --batch 512
timeout 300 python tools/single_image_latency.py > $out/single_image.txt 2>&1
bash tools/gpu_soak.sh $tag/soak > /dev/null 2>&1; cp gpurun_out/$tag/soak/soak.txt $out/soak.txt
rm -f $out/*_agent_info.csv $out/*_domain_stats.csv $out/*_kernel_trace.csv
cat $out/pytest_gpu.txt; cat $out/bench_default.json | cut -c1-400
python - <<PY
import json
for l in open("$out/pipe.jsonl"):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print("%8.0f Mpix/s  %.4f ms/img  host %.4f ms/img  batch %d depth %d distinct %d rounds %d devimgs %d/%d" % (d["mpix_s"], d["ms_per_image"], d["host_submit_ms_per_image"], d["batch"], d["depth"], d.get("distinct", 2), d["stats"]["spec_rounds_max"], d["stats"]["device_images"], d["stats"]["images"]))
PY
tail -3 $out/soak.txt; tail -8 $out/single_image.txt
