#!/bin/bash
# whole GPU suite, smoke, the default bench line (with configs), fuzz of the pipeline
out=gpurun_out/r3_w; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
timeout 600 python tools/gpu_fuzz_pipeline.py > $out/fuzz.txt 2>&1; tail -2 $out/fuzz.txt
