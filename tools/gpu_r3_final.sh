#!/bin/bash
# the round's closing run: whole GPU suite, smoke, the default bench line twice (second copy with configs kept for profiles/), the evidence passes
out=gpurun_out/r3_final; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
timeout 900 python bench.py > $out/bench_default_configs.json 2> $out/bench.err; tail -c 600 $out/bench_default_configs.json
bash tools/gpu_r3_profiles.sh r03_prof > $out/profiles.txt 2>&1; tail -3 $out/profiles.txt
