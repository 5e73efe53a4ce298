#!/bin/bash
out=gpurun_out/r02_pipe1
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt | tail -12
for cfg in "--depth 1 --threads 8" "--depth 2 --threads 8" "--depth 3 --threads 8" "--depth 3 --threads 16" "--depth 3 --threads 4 --batch 32 --batches 24"; do
  timeout 300 python tools/pipeline_bench.py $cfg 2>&1 | tail -1 >> $out/pipe.txt
done
timeout 300 python tools/pipeline_bench.py --restart-rows 1 2>&1 | tail -1 >> $out/pipe.txt
timeout 300 python tools/pipeline_bench.py --width 1920 --height 1080 --batch 256 --batches 8 2>&1 | tail -1 >> $out/pipe.txt
cat $out/pipe.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o pipe -- python $GRAFT_REPO_ROOT/tools/pipeline_bench.py --depth 3 --threads 8 --batches 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; ls $out; head -20 $out/pipe_kernel_stats.csv
