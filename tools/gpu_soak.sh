#!/bin/bash
# a longer run for leaks and drift: tools/gpu_soak.sh tag
tag=${1:-soak}; out=gpurun_out/$tag; mkdir -p $out
python - <<'PY' > $out/soak.txt 2>&1
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import jpegdec_amd as J
from tests.cases import SYNTH_CASES, PROGRESSIVE_CASES, jpeg_for
from tests.ref_fixtures import GOOD, FAIL_IN_DECODE, REJECTED_AT_OPEN, ref_jpeg
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
ctx = J.Context(0)
names = sorted(SYNTH_CASES) + sorted(PROGRESSIVE_CASES)
jp = [jpeg_for(n) for n in names] + [ref_jpeg(n) for n in GOOD + FAIL_IN_DECODE + REJECTED_AT_OPEN]
pts = [(i % 4) if not (names[i].startswith(("gray", "pgray")) and i % 4 == 2) else 0 for i in range(len(names))] + [2] * (len(jp) - len(names))
opts = [0] * len(jp)
pipe = J.Pipeline(ctx, max_images=64, depth=2, host_threads=4)
outs = []
for j, pt in zip(jp, pts):
    info = J.parse(j)
    if info["status"] != 0 or info["mcu_w"] == 0:
        outs.append((ctx.malloc(4096), 64, 16, 16)); continue
    ii = J.binding.ImageInfo(**{k: v for k, v in info.items() if k != "status"})
    try:
        g = J.output_geometry(ii, pt, 0)
    except J.JdaError:
        outs.append((ctx.malloc(4096), 64, 16, 16)); continue
    pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
    outs.append((ctx.malloc(pitch * g["canvas_h"]), pitch, g["canvas_w"], g["canvas_h"]))
first = None
t0 = time.time(); f0 = free_mb()
N = 300
prev = None
for it in range(N):
    t = pipe.submit(jp, outs, pts, opts)
    if prev is not None:
        st = pipe.wait(prev)
        if first is None: first = list(st)
        assert list(st) == first, (it, st, first)
    prev = t
    if it % 60 == 0: print("iteration", it, "free MB", round(free_mb()), flush=True)
st = pipe.wait(prev)
assert list(st) == first
print("batches", N, "images", N * len(jp), "seconds %.1f" % (time.time() - t0), "free MB before/after", round(f0), round(free_mb()), "stats", pipe.stats)
# the class, many objects in a row
from oracle.loader import RefDecoder
cls = RefDecoder(False, path=os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests", "libjpegdec_class_shim.so"))
f1 = free_mb(); t0 = time.time()
for it in range(400):
    r = cls.decode_cb(jp[it % len(names)], pts[it % len(names)], 0)
print("class decodes 400 in %.1f s, free MB before/after" % (time.time() - t0), round(f1), round(free_mb()))
PY
cat $out/soak.txt | tail -12
