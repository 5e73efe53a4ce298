#!/bin/bash
# how many pre-scan rounds get a launch of their own before jda_segscan_tail takes the rest (JDA_PIPE_SPEC_ROUNDS: ab/lib_spec<N>.so; in-tree: 4)
out=gpurun_out/r3_spec; rm -rf $out; mkdir -p $out
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in ab/lib_spec2.so ab/lib_spec3.so jpegdec_amd/libjpegdec_amd.so; do
    echo -n "$lib " >> $out/e2e.txt
    JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 --batches 24 --distinct 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']), d['stats']['spec_rounds_max'], d['stats']['host_path_images'])" >> $out/e2e.txt
  done
done
for lib in ab/lib_spec2.so jpegdec_amd/libjpegdec_amd.so; do
  echo -n "1080p x256 $lib " >> $out/e2e.txt
  JDA_LIBRARY=$R/$lib timeout 300 python tools/pipeline_bench.py --depth 4 --batch 256 --batches 12 --distinct 16 --width 1920 --height 1080 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['mpix_s']))" >> $out/e2e.txt
done
cat $out/e2e.txt
