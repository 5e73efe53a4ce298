#!/usr/bin/env python3
"""Restart-marker damage through the HOST SIMULATOR of the device pre-scan (no GPU; the oracle is the checker): markers deleted, doubled,
renumbered, inserted in the middle of an interval, moved by a few bytes, DRI changed, the stream cut -- alone and together with byte
damage.  The reference counts MCUs and never looks for markers (jpeg.inl:5337-5348); the segment walk follows them: every stream it
accepts must come out with the serial pre-scan's index, every picture and status as the oracle's.
python tools/cpu_fuzz_markers.py [streams per base file] [seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jpegdec_amd as J  # noqa: E402
from oracle.loader import OracleDecoder  # noqa: E402
from tests.cases import SYNTH_CASES, jpeg_for  # noqa: E402
from tests.ref_fixtures import ref_jpeg  # noqa: E402

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libjda_hostsim.so"))
lib.hostsim_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
oracle = OracleDecoder()
bases = [n for n in sorted(SYNTH_CASES) if "rst" in n] + ["ref:tulips"]


def markers(j, sos):
    out, i = [], sos + 2
    while i < len(j) - 1:
        if j[i] == 0xFF and 0xD0 <= j[i + 1] <= 0xD7:
            out.append(i); i += 2
        else:
            i += 1
    return out


def damage(base, sos, rng):
    b = bytearray(base)
    ms = markers(b, sos)
    kind = int(rng.integers(0, 8))
    if not ms:
        kind = 7
    m = ms[int(rng.integers(0, len(ms)))] if ms else 0
    if kind == 0:                                   # a marker gone
        del b[m:m + 2]
    elif kind == 1:                                 # a marker twice
        b[m:m] = b[m:m + 2]
    elif kind == 2:                                 # renumbered
        b[m + 1] = 0xD0 + int(rng.integers(0, 8))
    elif kind == 3:                                 # one more, somewhere in the entropy-coded data
        at = int(rng.integers(sos + 14, len(b) - 2))
        b[at:at] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
    elif kind == 4:                                 # moved by a few bytes
        mk = bytes(b[m:m + 2]); del b[m:m + 2]
        at = max(sos + 14, min(len(b) - 2, m + int(rng.integers(-6, 7))))
        b[at:at] = mk
    elif kind == 5:                                 # another restart interval in the DRI segment
        i = bytes(b).find(b"\xff\xdd\x00\x04")
        if i >= 0:
            v = max(1, ((b[i + 4] << 8) | b[i + 5]) + int(rng.integers(-2, 3)))
            b[i + 4], b[i + 5] = v >> 8, v & 255
    elif kind == 6:                                 # the last markers' neighbourhood damaged
        lo = ms[max(0, len(ms) - 3)]
        b[int(rng.integers(lo, len(b) - 2))] = int(rng.integers(0, 256))
    if kind == 7 or rng.integers(0, 3) == 0:        # + plain byte damage
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
    return bytes(b), kind


total = used = agree = bad_n = 0
lib.hostsim_set_device_prescan(2)
for name in bases:
    base = bytearray(ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name))
    sos = bytes(base).index(b"\xff\xda")
    for it in range(n_per):
        jb, kind = damage(base, sos, rng)
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
            bad_n += 1
            fn = "/tmp/marker_fuzz_%s_%d.jpg" % (name.replace(":", "_"), it)
            open(fn, "wb").write(jb)
            print("MISMATCH %s #%d kind %d: %s -> %s" % (name, it, kind, bad, fn), flush=True)
        used += 1 if lib.hostsim_prescan_used() else 0
        agree += 1 if rc == 1 else 0
    print("%s: %d streams so far, %d indexed by the walk, %d decoded" % (name, total, used, agree), flush=True)
lib.hostsim_set_device_prescan(0)
print("streams %d, indexed by the walk %d, decoded by the oracle %d, mismatches %d" % (total, used, agree, bad_n))
