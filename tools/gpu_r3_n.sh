#!/bin/bash
out=gpurun_out/r03_n
mkdir -p $out
(echo "python tools/gpu_fuzz_pipeline.py 100 30303 (round 3, RECORD-mode pre-scan; twelve base files incl. the reference's photographs, 60 per batch, three batches in flight):"; timeout 1200 python tools/gpu_fuzz_pipeline.py 100 30303 2>&1 | tail -3) > $out/fuzz.txt
cat $out/fuzz.txt
bash tools/gpu_soak.sh r03_n_soak 2>&1 | tail -6
