#!/usr/bin/env python3
"""End-to-end pipeline, input modes side by side on one box: pageable (mirror), page-locked (direct DMA, adjacent files coalesced), at
1 / 2 / 8 host threads.  python tools/e2e_modes.py [batches]   (JDA_LIBRARY=ab/lib_lab.so JDA_PIPE_TIME=1: submit phases on stderr)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jpegdec_amd as J  # noqa: E402
from bench import cached_jpeg  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = J.Context(0)
pool = [cached_jpeg(4096, 4096, "4:2:0", 1234 + i) for i in range(16)]
eb, depth, pt = 64, 4, J.RGB8888
p0 = J.PreparedImage(pool[0]); geo = p0.geometry(pt, 0); p0.close()
pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
img_bytes = pitch * geo["canvas_h"]
surf = [ctx.malloc(img_bytes * eb) for _ in range(depth)]
outs_of = [[(b + i * img_bytes, pitch, geo["canvas_w"], geo["canvas_h"]) for i in range(eb)] for b in surf]
picks = [i % 16 for i in range(eb)]
hot = J.PinnedFiles(pool)


def run(threads, mode):
    pipe = J.Pipeline(ctx, max_images=eb, depth=depth, host_threads=threads)
    if mode == "pageable":
        packed = [pipe.pack([pool[k] for k in picks], o, [pt] * eb, [0] * eb) for o in outs_of]
        flags = 0
    else:
        packed = [pipe.pack_pinned(hot, picks, o, [pt] * eb, [0] * eb) for o in outs_of]
        flags = J.SUBMIT_PINNED_INPUT
    inflight, warm, t0, ts = [], 4, 0.0, 0.0
    for k in range(warm + batches):
        if k == warm:
            while inflight:
                pipe.wait(inflight.pop(0))
            ctx.sync(); t0 = time.perf_counter()
        if len(inflight) == depth:
            pipe.wait(inflight.pop(0))
        a = time.perf_counter()
        inflight.append(pipe.submit_packed(packed[k % depth], flags))
        if k >= warm:
            ts += time.perf_counter() - a
    while inflight:
        pipe.wait(inflight.pop(0))
    ctx.sync()
    dt = time.perf_counter() - t0
    st = pipe.stats
    pipe.close()
    return geo["out_w"] * geo["out_h"] * eb * batches / dt / 1e9, ts / batches * 1e3, st["h2d_bytes"] / st["images"] * eb / 1e6


for rep in range(2):
    for mode in ("pageable", "pinned"):
        for th in (1, 2, 8):
            g, sub, mb = run(th, mode)
            print("%-9s threads %d: %6.1f Gpix/s, submit %.3f ms/batch, H2D %.1f MB/batch" % (mode, th, g, sub, mb), flush=True)
