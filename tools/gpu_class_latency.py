#!/usr/bin/env python3
"""Latency of the drop-in class (include/JPEGDEC.h through tests/libjpegdec_class_shim.so): openRAM + decode with a draw callback that
counts pixels + close, one object reused, per image size.  Usage (GPU box): python tools/gpu_class_latency.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.loader import RefDecoder  # noqa: E402  (the shim's loader: the same C entry points as the reference's shim, our class behind them)
from jpegdec_amd.synth import synth_jpeg  # noqa: E402

cls = RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))
for w, h in ((640, 480), (1280, 720), (1920, 1080), (4096, 4096)):
    jpeg = synth_jpeg(w, h, "4:2:0", seed=5)
    # a C loop on a thread of its own: openRAM + decode (no-op draw callback) + close on one object.  The thread's device context is
    # created on its first decode (milliseconds, once per thread): runs long enough to make that a per cent
    n = 2000 if w * h < 1000000 else (1000 if w * h < 4000000 else 500)
    r = cls.bench([jpeg], 0, 0, reps=n, threads=1)
    dt = r["seconds"] / n
    assert r["failures"] == 0
    print("%dx%d: class openRAM + decode (RGB565, draw callbacks) + close %.2f ms = %.0f Mpix/s" % (w, h, dt * 1e3, w * h / dt / 1e6))
