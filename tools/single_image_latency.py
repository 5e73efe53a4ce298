#!/usr/bin/env python3
"""Latency of the one-call path the JPEGDEC class uses (jda_decode_to_host: prepare + upload + plan + decode + copy back)
for single images of a few sizes.  Usage (GPU box): python tools/single_image_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegdec_amd as J  # noqa: E402
from jpegdec_amd.synth import synth_jpeg  # noqa: E402

ctx = J.Context(0)
for w, h in ((640, 480), (1280, 720), (1920, 1080), (4096, 4096)):
    jpeg = synth_jpeg(w, h, "4:2:0", seed=5)
    for _ in range(3):
        J.decode_to_host(ctx, jpeg, J.RGB565_LE, 0)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        rc, px, g = J.decode_to_host(ctx, jpeg, J.RGB565_LE, 0)
    dt_fresh = (time.perf_counter() - t0) / n          # a fresh canvas per call: its page faults are in the copy back
    t0 = time.perf_counter()
    for _ in range(n):
        rc, px, g = J.decode_to_host(ctx, jpeg, J.RGB565_LE, 0, out=px)
    dt = (time.perf_counter() - t0) / n
    p = J.PreparedImage(jpeg)
    t1 = time.perf_counter()
    for _ in range(5):
        q = J.PreparedImage(jpeg); q.close()
    tp = (time.perf_counter() - t1) / 5
    p.close()
    print("%dx%d: decode_to_host %.2f ms per image into the same canvas (%.2f into a fresh one; the host prepare alone, default flags: %.2f ms) = %.0f Mpix/s"
          % (w, h, dt * 1e3, dt_fresh * 1e3, tp * 1e3, w * h / dt / 1e6))
ctx.close()
