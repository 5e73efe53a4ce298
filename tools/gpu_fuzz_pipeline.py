#!/usr/bin/env python3
"""A longer run of tests/test_gpu_pipeline.py's corruption test (test infrastructure: the oracle is the checker): random byte corruptions
inside the entropy-coded data of a few files, batches of them through the device filter, segment walk and decode, every surface and status
against the oracle's.  Usage (GPU box): python tools/gpu_fuzz_pipeline.py [rounds] [seed] [mixed]
(mixed: pixel type and scale drawn per image -- the 1/4 and 1/8 kernels and the scaled colour stages see the corrupted streams too)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jpegdec_amd as J  # noqa: E402
from oracle.loader import OracleDecoder  # noqa: E402
from tests.cases import jpeg_for  # noqa: E402
from tests.test_gpu_pipeline import _check, _surfaces  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
ctx = J.Context(0)
oracle = OracleDecoder()
pipe = J.Pipeline(ctx, max_images=128, depth=3, host_threads=4)
# (round 3: + the reference's photographs -- they have the magnitude reads the reference truncates, i.e. the RECORD-mode pre-scan's
# candidates and flagged entries, with and without restart intervals)
bases = ("c420_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c422_333x217", "c420_1280x720", "gray_333x217", "c420_256x256_q98", "c440_200x120",
         "ref:zebra", "ref:st_peters", "ref:tulips", "ref:sciopero", "w16_c420_333x217_x400", "w16_gray_200x120_x400")
from tests.ref_fixtures import ref_jpeg  # noqa: E402
total = on_device = failed = 0
inflight = []
for r in range(rounds):
    jp, nm = [], []
    for name in bases:
        base = bytearray(ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name))
        sos = bytes(base).index(b"\xff\xda")
        made = 0
        while made < (8 if len(bases) == 8 else 5):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            out_of_contract = (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan())      # (ran out of data)
            p.close()
            if out_of_contract:
                continue
            jp.append(jb); nm.append("%s#%d.%d" % (name, r, made)); made += 1
    pts = [J.RGB8888] * len(jp)
    opts = [0] * len(jp)
    if len(sys.argv) > 3 and sys.argv[3] == "mixed":
        combos = ((J.RGB8888, 0), (J.RGB565_LE, J.SCALE_QUARTER), (J.GRAY8, J.SCALE_QUARTER), (J.RGB8888, J.SCALE_QUARTER), (J.RGB565_BE, J.SCALE_QUARTER),
                  (J.RGB565_BE, J.SCALE_EIGHTH), (J.GRAY8, J.SCALE_HALF), (J.RGB8888, J.SCALE_HALF))
        for i in range(len(jp)):
            pts[i], opts[i] = combos[int(rng.integers(0, len(combos)))]
    outs, metas = _surfaces(ctx, jp, pts, opts)
    inflight.append((pipe.submit(jp, outs, pts, opts), jp, pts, opts, outs, metas, nm))
    if len(inflight) == 3 or r == rounds - 1:                 # three batches in flight: the pre-scan streams really run side by side
        while inflight:
            t, jp_, pts_, opts_, outs_, metas_, nm_ = inflight.pop(0)
            st = pipe.wait(t)
            _check(ctx, oracle, jp_, pts_, opts_, outs_, metas_, st, nm_)
            total += len(jp_); failed += sum(1 for s in st if s != 0)
            for o in outs_:
                ctx.free(o[0])
on_device = pipe.stats["device_images"]
print("corrupted streams %d, indexed on the device %d, decode errors %d (as the oracle's): all surfaces and statuses equal" % (total, on_device, failed))
pipe.close(); ctx.close()
