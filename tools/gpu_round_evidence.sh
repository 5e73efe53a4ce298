#!/bin/bash
# the round's pipeline evidence in one call: per-kernel times at depth 1 for four batch shapes, one steady-state period (timeline) of the
# metric batch and of 1,024 x 1280x720 at depth 4, corrupted streams through the pipeline against the oracle -> gpurun_out/evidence/
prefix=${1:-r06}
out=gpurun_out/evidence; mkdir -p $out
bash tools/gpu_pipeline_stats.sh $prefix > $out/${prefix}_pipeline_kernel_stats_depth1.txt 2>&1
cp gpurun_out/pipe_stats/${prefix}_pipeline_*_kernel_stats_depth1.csv $out/ 2>/dev/null
bash tools/gpu_pipeline_timeline.sh 4096 4096 64 16 4 > $out/${prefix}_pipeline_timeline_4096x4096_depth4.txt 2>&1
bash tools/gpu_pipeline_timeline.sh 1280 720 1024 8 4 > $out/${prefix}_pipeline_timeline_1280x720_depth4.txt 2>&1
(python tools/gpu_fuzz_pipeline.py 200 2501 mixed; python tools/gpu_fuzz_pipeline.py 200 2502) 2>&1 | grep -v amdgpu.ids | tail -4 > $out/${prefix}_gpu_fuzz.txt
tail -3 $out/*.txt | cut -c1-220
