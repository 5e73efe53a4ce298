#!/bin/bash
# one steady-state period of the streamed pipeline at depth 3: tools/gpu_pipeline_timeline.sh W H BATCH [DISTINCT]  (kernel + copy trace -> tools/pipeline_timeline.py)
out=gpurun_out/timeline; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$out -o tl_$1 -- python $R/tools/pipeline_bench.py --depth ${5:-3} --batches 16 --width $1 --height $2 --batch $3 --distinct ${4:-8} > $R/$out/tl_$1_line.json 2>/dev/null)
cat $out/tl_$1_line.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['mpix_s']), 'Mpix/s (traced)')"
python tools/pipeline_timeline.py $out/tl_$1
