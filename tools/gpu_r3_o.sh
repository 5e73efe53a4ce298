#!/bin/bash
out=gpurun_out/r03_o
mkdir -p $out
make nodeuser semuser >/dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
(echo "python tools/single_image_latency.py (GPU box, one MI355X, round 3)"; timeout 600 python tools/single_image_latency.py 2>&1 | tail -8; echo; echo "python tools/gpu_class_latency.py"; timeout 600 python tools/gpu_class_latency.py 2>&1 | tail -6) > $out/latency.txt
cat $out/latency.txt
