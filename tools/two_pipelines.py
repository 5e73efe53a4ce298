#!/usr/bin/env python3
"""One pipeline at depth 4 against two pipelines (a context each) at depth 2 + 2 / 4 + 4 on one GPU, fed in turn by one thread: files in
page-locked host memory -> pixels in HBM (the metric batch).  Usage (GPU box): python tools/two_pipelines.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegdec_amd as J
from bench import cached_jpeg

W = H = 4096; BATCH = 64; NB = 24
jpegs = [cached_jpeg(W, H, "4:2:0", 1234 + i) for i in range(16)]
files = [jpegs[i % 16] for i in range(BATCH)]
pin = J.PinnedFiles(files)

def make(depth):
    ctx = J.Context(0)
    p = J.PreparedImage(jpegs[0]); g = p.geometry(J.RGB8888, 0); p.close()
    pitch = (g["canvas_w"] * 4 + 15) & ~15
    img = pitch * g["canvas_h"]
    surf = [ctx.malloc(img * BATCH) for _ in range(depth)]
    pipe = J.Pipeline(ctx, max_images=BATCH, depth=depth, host_threads=4)
    packed = [pipe.pack_pinned(pin, list(range(BATCH)), [(b + i * img, pitch, g["canvas_w"], g["canvas_h"]) for i in range(BATCH)], [J.RGB8888] * BATCH, [0] * BATCH) for b in surf]
    return ctx, pipe, packed, depth

def run(pipes, nb):
    infl = [[] for _ in pipes]
    t0 = None
    warm = 8
    for k in range(warm + nb):
        if k == warm:
            for q, (ctx, pipe, packed, depth) in zip(infl, pipes):
                while q: pipe.wait(q.pop(0))
            t0 = time.perf_counter()
        i = k % len(pipes)
        ctx, pipe, packed, depth = pipes[i]
        if len(infl[i]) == depth:
            pipe.wait(infl[i].pop(0))
        infl[i].append(pipe.submit_packed(packed[(k // len(pipes)) % depth], J.SUBMIT_PINNED_INPUT))
    for q, (ctx, pipe, packed, depth) in zip(infl, pipes):
        while q: pipe.wait(q.pop(0))
    dt = time.perf_counter() - t0
    return W * H * BATCH * nb / dt / 1e9

import threading


def run_threads(pipes, nb):
    """every pipeline fed by a thread of its own (ctypes calls release the interpreter's lock)"""
    res = [0.0] * len(pipes)
    bar = threading.Barrier(len(pipes))

    def feed(i):
        bar.wait()
        t0 = time.perf_counter()
        run([pipes[i]], nb)
        res[i] = time.perf_counter() - t0
    th = [threading.Thread(target=feed, args=(i,)) for i in range(len(pipes))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return None


one = [make(4)]
two = [make(2), make(2)]
two4 = [make(4), make(4)]
for rep in range(3):
    print("one pipeline depth 4: %.1f Gpix/s   two pipelines depth 2+2: %.1f   two pipelines depth 4+4: %.1f" % (run(one, NB), run(two, NB), run(two4, NB)), flush=True)
# two feeder threads, a pipeline each: the whole job's rate (both run NB timed batches behind 8 warm-up ones; the clock runs from the barrier to the last join, warm-up included)
for rep in range(3):
    t0 = time.perf_counter()
    run_threads(two4, NB)
    dt = time.perf_counter() - t0
    print("two pipelines depth 4 + 4, a feeder thread each: %.1f Gpix/s (warm-up batches counted)" % (W * H * BATCH * (NB + 8) * 2 / dt / 1e9), flush=True)
