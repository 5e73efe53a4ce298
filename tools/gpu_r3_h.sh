#!/bin/bash
out=gpurun_out/r03_h
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o rec_d2 -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 4 --threads 8 --batches 12 --distinct 2 > /dev/null 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out -o rec_d16 -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 4 --threads 8 --batches 12 --distinct 16 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/$out
