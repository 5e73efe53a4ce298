"""probe: which HIP call fails in the one-image path (round 3)"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jpegdec_amd as J
from tests.ref_fixtures import ref_jpeg
ctx = J.Context(0)
for name, pt, opt in (("tulips", 2, 0), ("demo", 2, 2), ("demo", 0, 0), ("tulips", 1, 8), ("zebra", 2, 2), ("perf", 2, 0), ("squirrel_dither", 2, 0)):
    j = ref_jpeg(name)
    try:
        rc, got, g = J.decode_to_host(ctx, j, pt, opt)
        print(name, pt, opt, "rc", flush=True) or print("  ", rc, "hip:", (ctx.lib.jda_last_hip_error(ctx.handle) or b"").decode())
    except Exception as e:
        print(name, pt, opt, "EXC", e, "hip:", (ctx.lib.jda_last_hip_error(ctx.handle) or b"").decode())
