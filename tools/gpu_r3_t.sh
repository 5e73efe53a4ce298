#!/bin/bash
# the pipeline's timeline at depth 4 (kernels + copies), 16 distinct files per batch
out=gpurun_out/r3_t; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$out -o d4 -- python $R/tools/pipeline_bench.py --depth 4 --batches 16 --distinct 16 > $R/$out/run.txt 2>&1)
tail -1 $out/run.txt
python tools/pipeline_timeline.py $out/d4
