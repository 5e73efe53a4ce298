#!/usr/bin/env python3
"""Corrupted streams through the HOST SIMULATOR of the device pre-scan (tests/hostsim: the kernels' per-lane logic compiled for the
CPU; test infrastructure, the oracle is the checker) -- no GPU needed, so it can run for as long as one likes:
python tools/cpu_fuzz_hostsim.py [streams per base file] [seed].  Every stream the front end accepts goes through the segment walk in
RECORD mode (as the pipeline runs it); an index the walk's checks accept must be the serial pre-scan's (positions, flags, flagged
entries, predictors, truncation count), the picture and the status the oracle's."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jpegdec_amd as J  # noqa: E402
from oracle.loader import OracleDecoder  # noqa: E402
from tests.cases import SYNTH_CASES, jpeg_for  # noqa: E402

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libjda_hostsim.so"))
lib.hostsim_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
oracle = OracleDecoder()
max_px = int(os.environ.get("FUZZ_MAX_PIXELS", 640 * 368))
bases = [n for n in sorted(SYNTH_CASES) if SYNTH_CASES[n]["width"] * SYNTH_CASES[n]["height"] <= max_px]
try:
    from tests.ref_fixtures import GOOD, ref_jpeg
    extra = [("ref:" + n) for n in GOOD]
except Exception:
    extra = []
total = used = agree = 0
lib.hostsim_set_device_prescan(2)
for name in list(bases) + extra:
    try:
        base = bytearray(ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name))
    except Exception:
        continue
    if len(base) > 400000:
        continue
    sos = bytes(base).index(b"\xff\xda")
    for it in range(n_per):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            lo = sos + 14 if it % 2 else len(b) - max((len(b) - sos) // 10, 40)
            b[int(rng.integers(lo, len(b) - 2))] = int(rng.integers(0, 256))
        jb = bytes(b)
        try:
            p = J.PreparedImage(jb)
        except J.JdaError:
            continue
        idx, nok = p.block_index()
        ooc = (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan())
        p.close()
        if ooc:
            continue
        rc, want, err = oracle.decode_canvas(jb, J.RGB8888, 0)
        got = np.full_like(want, 0x33)
        inf, cx, cy, mw, mh, bpp, sh = oracle.canvas_geometry(jb, J.RGB8888, 0)
        hrc = lib.hostsim_decode(jb, len(jb), J.RGB8888, 0, got.ctypes.data_as(C.c_void_p), got.shape[1], cx * mw, cy * mh)
        total += 1
        bad = None
        if (rc == 1) != (hrc == 0):
            bad = "status: oracle %d (%d), simulator %d" % (rc, err, hrc)
        elif lib.hostsim_prescan_used() and lib.hostsim_index_equal() != 1:
            bad = "index accepted by the walk differs from the serial one"
        elif rc == 1 and not np.array_equal(got, want):
            bad = "picture differs (%d bytes)" % int(np.count_nonzero(got != want))
        if bad:
            fn = "/tmp/hostsim_fuzz_%s_%d.jpg" % (name.replace(":", "_"), it)
            open(fn, "wb").write(jb)
            print("MISMATCH %s #%d: %s -> %s" % (name, it, bad, fn), flush=True)
        used += 1 if lib.hostsim_prescan_used() else 0
        agree += 1 if rc == 1 else 0
    print("%s: %d streams so far, %d indexed by the walk, %d decoded" % (name, total, used, agree), flush=True)
lib.hostsim_set_device_prescan(0)
print("streams %d, indexed by the walk %d, decoded by the oracle %d" % (total, used, agree))
