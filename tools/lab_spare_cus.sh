for sp in 0 8 16 32; do echo "spare $sp"; JDA_LIBRARY=$GRAFT_REPO_ROOT/ab/lib_lab.so JDA_DECODE_SPARE_CUS=$sp python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import jpegdec_amd as J
from bench import cached_jpeg
ctx = J.Context(0)
pool = [cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(16)]
eb, depth, pt = 64, 4, J.RGB8888
p0 = J.PreparedImage(pool[0]); geo = p0.geometry(pt, 0); p0.close()
pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15; img_bytes = pitch * geo["canvas_h"]
surf = [ctx.malloc(img_bytes * eb) for _ in range(depth)]
hot = J.PinnedFiles(pool)
pipe = J.Pipeline(ctx, max_images=eb, depth=depth, host_threads=4)
packed = [pipe.pack_pinned(hot, [i % 16 for i in range(eb)], [(b + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(eb)], [pt] * eb, [0] * eb) for b in surf]
for rep in range(2):
    inflight, t0 = [], 0.0
    for k in range(4 + 32):
        if k == 4:
            while inflight: pipe.wait(inflight.pop(0))
            ctx.sync(); t0 = time.perf_counter()
        if len(inflight) == depth: pipe.wait(inflight.pop(0))
        inflight.append(pipe.submit_packed(packed[k % depth], 1))
    while inflight: pipe.wait(inflight.pop(0))
    ctx.sync(); dt = time.perf_counter() - t0
    print("  %.1f Gpix/s" % (geo["out_w"] * geo["out_h"] * eb * 32 / dt / 1e9), flush=True)
PY
done
