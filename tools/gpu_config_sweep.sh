#!/bin/bash
# BASELINE.json configs on one MI355X (GPU box, through gpurun); every run re-checks its first image bit-exact
# against oracle/_ref.  Output: gpurun_out/sweep.txt
out=gpurun_out/sweep.txt
: > $out
run() { label="$1"; shift; python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-62s %9.0f Mpix/s  %7.3f ms/step  frac %.3f  bit_exact %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['bit_exact'] if d.get('parity') else None))" "$label" >> $out; }
run "C2  batch 1024 x 1280x720  4:2:0 -> RGB8888" --batch 1024 --width 1280 --height 720 --distinct 4
run "C3  batch   64 x 4096x4096 4:4:4 -> RGB8888" --batch 64 --subsampling 4:4:4
run "C4  batch 1024 x 1920x1080 4:2:0 -> RGB8888 (one GPU's share)" --batch 1024 --width 1920 --height 1080 --distinct 4
run "C5  batch   16 x 8192x8192 gray  -> GRAY8 scale 1" --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8
run "C5  batch   16 x 8192x8192 gray  -> GRAY8 scale 1/2" --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8 --options 2
run "C5  batch   16 x 8192x8192 gray  -> GRAY8 scale 1/4" --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8 --options 4
run "C5  batch   16 x 8192x8192 gray  -> GRAY8 scale 1/8" --batch 16 --width 8192 --height 8192 --subsampling gray --pixel-type gray8 --options 8
run "metric batch 64 x 4096x4096 4:2:0 -> RGB8888" 
run "metric, RGB565 output" --pixel-type rgb565
run "metric, restart marker per MCU row + device pre-scan" --restart-rows 1 --device-prescan
run "4096x4096 4:2:2 -> RGB8888" --subsampling 4:2:2
run "metric image -> GRAY8" --pixel-type gray8
run "metric image, JPEG_SCALE_HALF -> RGB8888" --options 2
run "metric image, JPEG_SCALE_HALF -> RGB565" --options 2 --pixel-type rgb565
run "metric image, 1/4 -> RGB8888" --options 4
run "metric image, 1/8 -> RGB8888 (thumbnail)" --options 8
run "4096x4096 4:4:4, JPEG_SCALE_HALF -> RGB8888" --subsampling 4:4:4 --options 2
run "batch 1024 x 640x480 4:2:0 -> RGB565 (the reference's own test image size)" --batch 1024 --width 640 --height 480 --distinct 4 --pixel-type rgb565
cat $out
