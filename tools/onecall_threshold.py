#!/usr/bin/env python3
"""One image at a time: host prepare serial vs parallel, and jda_decode_to_host with the pre-scan on the host (parallel) or on the device
(lab build: JDA_ONECALL_DEVICE_PRESCAN_BYTES moves the switch).  Usage (GPU box): JDA_LIBRARY=ab/lib_lab.so python tools/onecall_threshold.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegdec_amd as J
from bench import cached_jpeg
ctx = J.Context(0)
def t(f, n):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
for (w, h, n) in ((1280, 720, 200), (1920, 1080, 100), (2560, 1440, 60), (3072, 2048, 40), (4096, 4096, 20)):
    jp = cached_jpeg(w, h, "4:2:0", 1234)
    ser = t(lambda: J.PreparedImage(jp, flags=J.PREPARE_SERIAL_PRESCAN).close(), n)
    par = t(lambda: J.PreparedImage(jp).close(), n)
    out = [None]
    def dec():
        rc, out[0], g = J.decode_to_host(ctx, jp, J.RGB565_LE, 0, out=out[0])
    d = t(dec, n)
    print("%dx%d (%d KB): prepare serial %.3f ms, parallel %.3f ms; decode_to_host %.3f ms (device pre-scan from %s bytes)" % (w, h, len(jp) >> 10, ser, par, d, os.environ.get("JDA_ONECALL_DEVICE_PRESCAN_BYTES", "131072")), flush=True)
