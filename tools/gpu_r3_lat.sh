#!/bin/bash
# one image at a time after "two symbols a step": the one-call C entry and the drop-in class
out=gpurun_out/r3_lat; mkdir -p $out
timeout 600 python tools/single_image_latency.py > $out/single.txt 2>&1; tail -12 $out/single.txt
timeout 600 python tools/gpu_class_latency.py > $out/class.txt 2>&1; tail -12 $out/class.txt
