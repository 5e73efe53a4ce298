#!/bin/bash
# other inputs through the pipeline after "two symbols a step": 1080p, 720p (batches of 64 and 256), 4:4:4, gray, q98, restart rows
out=gpurun_out/r3_x; mkdir -p $out
run() { echo -n "$* : " >> $out/other.txt; timeout 300 python tools/pipeline_bench.py --depth 4 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), 'Mpix/s, host submit', round(d['host_submit_ms_per_image'],4), 'ms/img, rounds', d['stats']['spec_rounds_max'], 'host path', d['stats']['host_path_images'])" >> $out/other.txt; }
run --batches 24 --distinct 16
run --batches 24 --distinct 2
run --batches 24 --distinct 16 --restart-rows 1
run --width 1920 --height 1080 --batches 40 --distinct 16
run --width 1920 --height 1080 --batch 256 --batches 12 --distinct 16
run --width 1280 --height 720 --batches 60 --distinct 16
run --width 1280 --height 720 --batch 256 --batches 16 --distinct 16
run --subsampling 4:4:4 --batches 16 --distinct 8
run --subsampling gray --width 8192 --height 8192 --batch 16 --batches 16 --distinct 4
run --quality 98 --batches 12 --distinct 8
cat $out/other.txt
