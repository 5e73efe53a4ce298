"""N > 1 path on CPU: world_size 2 over gloo.  Each rank takes its shard of a batch of images,
runs them through the wave emulator (tests/hostsim: the kernel logic compiled for the CPU -- test
infrastructure) and the control collectives of jpegdec_amd/sharding.py are exercised: barrier,
max-over-ranks timing, sum of counters, gather of per-image digests."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["c420_333x217", "c444_333x217", "gray_333x217", "c420_16x16", "c420_1100x48"]


def _decode_digest(name):
    from oracle.loader import OracleDecoder
    from tests.cases import jpeg_for

    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libjda_hostsim.so"))
    lib.hostsim_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    jpeg = jpeg_for(name)
    pt = 3 if name.startswith("gray") else 2
    inf, cx, cy, mw, mh, bpp, sh = OracleDecoder().canvas_geometry(jpeg, pt, 0)
    out = np.zeros((cy * mh, cx * mw * bpp), dtype=np.uint8)
    assert lib.hostsim_decode(jpeg, len(jpeg), pt, 0, out.ctypes.data_as(C.c_void_p), out.shape[1], cx * mw, cy * mh) == 0
    return hashlib.sha256(out).hexdigest()[:16], inf["width"] * inf["height"]


def _worker(rank, world, port, results):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from jpegdec_amd.sharding import Group, shard_range

    g = Group(backend="gloo")
    lo, hi = shard_range(len(NAMES), g.rank, g.world)
    g.barrier()
    mine = {NAMES[i]: _decode_digest(NAMES[i]) for i in range(lo, hi)}
    g.barrier()
    elapsed = g.max(1.0 + g.rank)                      # max over ranks
    pixels = g.sum(sum(v[1] for v in mine.values()))   # whole-job work
    everyone = g.gather_objects(mine)
    if g.rank == 0:
        merged = {}
        for d in everyone:
            assert not (set(d) & set(merged)), "an image was decoded by two ranks"
            merged.update(d)
        results.put((elapsed, pixels, merged, [sorted(d) for d in everyone]))
    g.close()


def test_two_rank_sharding_over_gloo(built_checkers):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, pixels, merged, owners = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0                                              # MAX over ranks
    serial = {n: _decode_digest(n) for n in NAMES}
    assert merged == serial                                            # every image exactly once, same pixels
    assert pixels == sum(v[1] for v in serial.values())
    assert owners == [sorted(NAMES[0:3]), sorted(NAMES[3:5])]          # contiguous blocks, sizes differ by <= 1


def test_shard_ranges_cover_everything_once():
    from jpegdec_amd.sharding import owner_of, shard_range

    for n in (0, 1, 5, 8, 8192):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
                for i in range(lo, hi):
                    assert owner_of(i, n, world) == r
            assert seen == list(range(n))
