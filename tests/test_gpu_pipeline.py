"""The streamed pipeline (jda_pipeline_*): unfiltered scans to the GPU, marker filter + per-block index + decode on the device,
batches overlapped on two streams, per-image status.  Parity: every image's pixels == the oracle's, the device-made index ==
the serial host pre-scan's entry for entry (the filtered length too), a bad image does not poison its batch."""
import numpy as np
import pytest

import jpegdec_amd as J
from oracle.loader import digest
from tests.cases import PROGRESSIVE_CASES, SYNTH_CASES, jpeg_for
from tests.ref_fixtures import FAIL_IN_DECODE, GOOD, REJECTED_AT_OPEN, ref_golden, ref_jpeg

pytestmark = pytest.mark.gpu


def _surfaces(ctx, jpegs, pts, opts):
    outs, metas = [], []
    for j, pt, opt in zip(jpegs, pts, opts):
        info = J.parse(j)
        if info["status"] != 0 or info["mcu_w"] == 0:
            ptr = ctx.malloc(4096)
            outs.append((ptr, 64, 16, 16)); metas.append(None)
            continue
        ii = J.binding.ImageInfo(**{k: v for k, v in info.items() if k != "status"})
        try:
            g = J.output_geometry(ii, pt, opt)
        except J.JdaError:
            ptr = ctx.malloc(4096)
            outs.append((ptr, 64, 16, 16)); metas.append(None)
            continue
        pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
        ptr = ctx.malloc(pitch * g["canvas_h"])
        ctx.memset(ptr, 0x5a, pitch * g["canvas_h"])
        outs.append((ptr, pitch, g["canvas_w"], g["canvas_h"])); metas.append((g, pitch))
    return outs, metas


def _check(ctx, oracle, jpegs, pts, opts, outs, metas, status, names):
    for j, pt, opt, o, m, st, nm in zip(jpegs, pts, opts, outs, metas, status, names):
        try:
            orc, want, err = oracle.decode_canvas(j, pt, opt)
        except Exception:
            orc, want, err = -1, None, 2
        if orc == 1:
            assert st == 0, (nm, pt, opt, st)
            g, pitch = m
            got = ctx.to_host(o[0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]]
            assert np.array_equal(got, want), (nm, pt, opt, int(np.count_nonzero(got != want)))
        else:
            assert st != 0, (nm, pt, opt)


def test_pipeline_mixed_batches_bit_exact(gpu_ctx, oracle):
    """three batches in flight over every synthetic case (all layouts, restart intervals, progressive = host path) and mixed
    output formats; all the reference's fixtures incl. the corrupt ones in a fourth"""
    names = sorted(SYNTH_CASES) + sorted(PROGRESSIVE_CASES)
    jp = [jpeg_for(n) for n in names]
    modes = [(J.RGB8888, 0), (J.RGB565_LE, 0), (J.GRAY8, 0), (J.RGB565_BE, J.SCALE_HALF), (J.RGB8888, J.SCALE_QUARTER), (J.RGB565_LE, J.SCALE_EIGHTH)]
    pipe = J.Pipeline(gpu_ctx, max_images=64, depth=3, host_threads=4)
    batches = []
    for b in range(3):
        pts, opts = [], []
        for i, n in enumerate(names):
            pt, opt = modes[(i + b) % len(modes)]
            if n.startswith("gray") and pt == J.RGB8888:
                pt = J.RGB565_LE
            if n.startswith("c440") and pt == J.RGB8888 and (opt & 4):
                opt = 0
            if n.startswith("p"):                          # progressive: what the reference itself can do (tests/cases.py)
                pt, opt = (J.RGB565_LE, 0) if not n.startswith("pgray") else (J.GRAY8, 0)
            pts.append(pt); opts.append(opt)
        outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
        t = pipe.submit(jp, outs, pts, opts)
        batches.append((t, pts, opts, outs, metas))
    for t, pts, opts, outs, metas in batches:
        st = pipe.wait(t)
        _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, names)
        for o in outs:
            gpu_ctx.free(o[0])
    s = pipe.stats
    assert s["images"] == 3 * len(names) and s["failed_images"] == 0
    assert s["host_path_images"] == 3 * len(PROGRESSIVE_CASES), s      # everything else took the device path
    pipe.close()


def test_pipeline_reference_fixtures_and_bad_images(gpu_ctx, oracle):
    """tulips .. perf + corrupt1-5 + the truncated thumb_test in ONE batch: 9 decode with the real reference's hashes, the
    corrupt ones get their own status (JPEG_DECODE_ERROR), nobody else notices.  The device-made index of every good fixture
    equals the serial pre-scan's entry for entry, the filtered length too."""
    names = list(GOOD) + list(REJECTED_AT_OPEN) + list(FAIL_IN_DECODE)
    jp = [ref_jpeg(n) for n in names]
    pipe = J.Pipeline(gpu_ctx, max_images=32, depth=2)
    for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, J.SCALE_HALF), (J.GRAY8, 0)):
        outs, metas = _surfaces(gpu_ctx, jp, [pt] * len(jp), [opt] * len(jp))
        t = pipe.submit(jp, outs, [pt] * len(jp), [opt] * len(jp))
        st = pipe.wait(t)
        for n, s_, o, m in zip(names, st, outs, metas):
            if n in GOOD:
                fr = ref_golden()[n]["frames"]["%d:%d" % (pt, opt)]
                g, pitch = m
                got = gpu_ctx.to_host(o[0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)
                assert s_ == 0 and digest(got[: fr["h"], : fr["w"] * fr["bpp"]]) == fr["sha"], (n, pt, opt, s_)
            else:
                assert s_ == 2, (n, s_)                       # JPEG_DECODE_ERROR: rejected header or bad MCU
        if pt == J.RGB8888:
            for i, n in enumerate(names):
                if n not in GOOD or n == "corrupt5":
                    continue
                h = J.PreparedImage(jp[i])
                idx, dc, flen = pipe.read_index(t, i, h.n_blocks)
                assert flen == len(h.scan()), n
                assert J.index_equivalent(idx, h.block_index()[0]) and np.array_equal(dc, h.block_dc()), n
                h.close()
        for o in outs:
            gpu_ctx.free(o[0])
    s = pipe.stats
    assert s["failed_images"] == 3 * 5 and s["device_images"] >= 3 * 7
    pipe.close()


def test_pipeline_one_corrupt_image_in_sixteen(gpu_ctx, oracle):
    """VERDICT r1 task 7: one corrupt image in a 16-image upload -- 15 decode, 1 flagged"""
    good = jpeg_for("c420_333x217")
    bad = bytearray(good)
    sos = bytes(bad).index(b"\xff\xda")
    for k in range(40, 48):
        bad[sos + 14 + 700 + k] = 0xFF if k & 1 else 0x00      # stuffing bytes and stray markers in the middle of the scan
    bad = bytes(bad[: sos + 14 + 1500])                        # ... and the stream ends early
    jp = [good] * 16
    jp[5] = bad
    pipe = J.Pipeline(gpu_ctx, max_images=16, depth=1)
    outs, metas = _surfaces(gpu_ctx, jp, [J.RGB8888] * 16, [0] * 16)
    st = pipe.wait(pipe.submit(jp, outs, [J.RGB8888] * 16, [0] * 16))
    assert [s == 0 for s in st] == [i != 5 for i in range(16)], st
    assert st[5] == 2
    orc, want, _ = oracle.decode_canvas(good, J.RGB8888, 0)
    for i in range(16):
        if i == 5:
            continue
        g, pitch = metas[i]
        got = gpu_ctx.to_host(outs[i][0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]]
        assert np.array_equal(got, want), i
    for o in outs:
        gpu_ctx.free(o[0])
    pipe.close()


def test_pipeline_at_bench_size(gpu_ctx, ref_scalar):
    """two 4096x4096 images of the bench workload through the pipeline: frames == the real reference's"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = [os.path.join(root, "bench_cache", "synth_4096x4096_420_q85_s%d.jpg" % s) for s in (1234, 1235)]
    if not all(os.path.exists(p) for p in paths):
        pytest.skip("bench_cache inputs absent")
    jp = [open(p, "rb").read() for p in paths]
    pipe = J.Pipeline(gpu_ctx, max_images=8, depth=2)
    outs, metas = _surfaces(gpu_ctx, jp, [J.RGB8888] * 2, [0] * 2)
    st = pipe.wait(pipe.submit(jp, outs, [J.RGB8888] * 2, [0] * 2))
    assert st == [0, 0]
    for j, o, (g, pitch) in zip(jp, outs, metas):
        got = gpu_ctx.to_host(o[0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[: g["out_h"], : g["out_w"] * 4]
        want = ref_scalar.decode_cb(j, J.RGB8888, 0)["canvas"][: g["out_h"], : g["out_w"] * 4]
        assert np.array_equal(got, want)
        gpu_ctx.free(o[0])
    assert pipe.stats["device_images"] == 2
    pipe.close()


def _with_dri(jpeg: bytes, f):
    i = jpeg.index(b"\xff\xdd\x00\x04")
    v = (jpeg[i + 4] << 8) | jpeg[i + 5]
    nv = f(v)
    return jpeg[: i + 4] + bytes([nv >> 8, nv & 255]) + jpeg[i + 6:]


def test_pipeline_restart_streams_take_the_segment_walk(gpu_ctx, oracle):
    """Streams with restart intervals go through the same speculative segment walk as those without (one lane per 256 bytes, not
    one per interval): index and DC predictors == the serial pre-scan's entry for entry, pixels == the oracle's; a file whose
    DRI does not match where its markers are (same number of intervals, other places -- or another number) is noticed on the
    device and decoded through the serial pre-scan, like the reference (which counts MCUs) would."""
    names = [n for n in sorted(SYNTH_CASES) if "rst" in n]
    assert len(names) >= 6
    jp = [jpeg_for(n) for n in names]
    odd = [_with_dri(jpeg_for("c420_640x368_rstrow"), lambda v: v + 1), _with_dri(jpeg_for("c420_512x256_q98_rstrow"), lambda v: v + 1),
           _with_dri(jpeg_for("c420_640x368_rstrow"), lambda v: v * 2), _with_dri(jpeg_for("c444_384x192_q100_rst7"), lambda v: v - 1)]
    allj = jp + odd
    nm = names + ["odd%d" % i for i in range(len(odd))]
    pts = [J.GRAY8 if n.startswith("gray") else J.RGB8888 for n in nm]
    opts = [0] * len(allj)
    pipe = J.Pipeline(gpu_ctx, max_images=32, depth=2, host_threads=2)
    outs, metas = _surfaces(gpu_ctx, allj, pts, opts)
    t = pipe.submit(allj, outs, pts, opts)
    st = pipe.wait(t)
    _check(gpu_ctx, oracle, allj, pts, opts, outs, metas, st, nm)
    for i, n in enumerate(names):
        h = J.PreparedImage(jp[i])
        idx, dc, flen = pipe.read_index(t, i, h.n_blocks)
        assert flen == len(h.scan()), n
        assert J.index_equivalent(idx, h.block_index()[0]) and np.array_equal(dc, h.block_dc()), n
        h.close()
    s = pipe.stats
    assert s["device_images"] == len(names), s                 # every well-formed restart stream stayed on the device
    assert s["host_path_images"] == len(odd), s
    for o in outs:
        gpu_ctx.free(o[0])
    pipe.close()


@pytest.mark.parametrize("mixed", [False, True])
def test_pipeline_corrupted_scans_decode_like_the_oracle(gpu_ctx, oracle, mixed):
    """Random byte corruptions inside the entropy-coded data (the reference's fuzz idea, MacOS/JPEGDEC_Test/main.cpp:262-300), a whole
    batch of them through the device filter, segment walk and decode: the oracle's garbage bit for bit, failure on the same
    streams -- whether the device's index stood or the image went back to the serial pre-scan.
    mixed: pixel type and scale drawn per image.  (The decode is launched before the pre-scan's verdict is read: a kernel must be safe on
    the index of a stream the walk gave up on -- the 1/4-scale kernel, which reads the scan where the entries point, once was not.)"""
    rng = np.random.default_rng(5)
    jp, nm = [], []
    for name in ("c420_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "c422_333x217"):
        base = bytearray(jpeg_for(name))
        sos = bytes(base).index(b"\xff\xda")
        made = 0
        while made < 14:
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
            jp.append(jb); nm.append("%s#%d" % (name, made)); made += 1
    pts = [J.RGB8888] * len(jp)
    opts = [0] * len(jp)
    if mixed:
        combos = ((J.RGB565_LE, J.SCALE_QUARTER), (J.GRAY8, J.SCALE_QUARTER), (J.RGB8888, J.SCALE_QUARTER), (J.RGB565_BE, J.SCALE_EIGHTH),
                  (J.GRAY8, J.SCALE_HALF), (J.RGB8888, J.SCALE_HALF))
        for i in range(len(jp)):
            pts[i], opts[i] = combos[int(rng.integers(0, len(combos)))]
    pipe = J.Pipeline(gpu_ctx, max_images=64, depth=2, host_threads=4)
    outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
    st = pipe.wait(pipe.submit(jp, outs, pts, opts))
    _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, nm)
    s = pipe.stats
    assert s["device_images"] >= len(jp) // 4, s               # many corruptions leave a stream the device can still index
    for o in outs:
        gpu_ctx.free(o[0])
    pipe.close()


def test_pipeline_damaged_restart_markers(gpu_ctx, oracle):
    """The reference counts MCUs and never looks for restart markers (jpeg.inl:5337-5348); the device's walk follows them.  Markers
    deleted, doubled, renumbered, inserted in the middle of an interval, moved by a few bytes, another DRI value, a damaged last
    interval (tests/test_hostsim.py's regression: one byte in front of the last marker): every surface and status the oracle's --
    whether the device's index stood or the image went back to the serial pre-scan."""
    rng = np.random.default_rng(9)
    jp, nm = [], []
    fix = bytearray(jpeg_for("c444_384x192_q100_rst7")); fix[121832] = 0x2a
    jp.append(bytes(fix)); nm.append("one byte in front of the last marker")
    for name in ("c420_640x368_rstrow", "c444_384x192_q100_rst7", "c440_300x64_rst5", "gray_64x64_rst3"):
        base = bytearray(jpeg_for(name))
        sos = bytes(base).index(b"\xff\xda")
        ms = [i for i in range(sos + 2, len(base) - 1) if base[i] == 0xFF and 0xD0 <= base[i + 1] <= 0xD7]
        made = it = 0
        while made < 14 and it < 200:
            it += 1
            b = bytearray(base)
            m = ms[int(rng.integers(0, len(ms)))]
            kind = it % 7
            if kind == 0: del b[m:m + 2]
            elif kind == 1: b[m:m] = b[m:m + 2]
            elif kind == 2: b[m + 1] = 0xD0 + int(rng.integers(0, 8))
            elif kind == 3:
                at = int(rng.integers(sos + 14, len(b) - 2)); b[at:at] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
            elif kind == 4:
                mk = bytes(b[m:m + 2]); del b[m:m + 2]
                at = max(sos + 14, min(len(b) - 2, m + int(rng.integers(-6, 7)))); b[at:at] = mk
            elif kind == 5:
                i = bytes(b).find(b"\xff\xdd\x00\x04")
                v = max(1, ((b[i + 4] << 8) | b[i + 5]) + int(rng.integers(-2, 3))); b[i + 4], b[i + 5] = v >> 8, v & 255
            else:
                b[int(rng.integers(ms[max(0, len(ms) - 3)], len(b) - 2))] = int(rng.integers(0, 256))
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
            jp.append(jb); nm.append("%s#%d kind %d" % (name, made, kind)); made += 1
    pts = [J.RGB8888] * len(jp)
    opts = [0] * len(jp)
    pipe = J.Pipeline(gpu_ctx, max_images=64, depth=2, host_threads=4)
    outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
    st = pipe.wait(pipe.submit(jp, outs, pts, opts))
    _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, nm)
    assert pipe.stats["host_path_images"] >= 1                    # (the regression case at least went back to the serial pre-scan)
    for o in outs:
        gpu_ctx.free(o[0])
    pipe.close()


def test_pipeline_on_the_densest_streams(gpu_ctx, oracle):
    """Flat images: some 390 block starts per 256-byte segment (every block its two shortest codes) -- the RECORD-mode pre-scan's record
    slots hold them, with and without restart intervals, in 4:2:0 and 4:4:4; device index == the serial one, pixels == the oracle's."""
    from jpegdec_amd.synth import encode_jpeg_custom
    flat = np.full((256, 640, 3), 117, np.uint8)
    flat[100:140, 300:360] = 30
    jp = [encode_jpeg_custom(flat, 90, hv, restart_interval=ri) for hv, ri in (((2, 2), 0), ((1, 1), 0), ((2, 2), 7), ((2, 1), 0))]
    names = ["flat420", "flat444", "flat420_rst7", "flat422"]
    pts, opts = [J.RGB8888] * len(jp), [0] * len(jp)
    pipe = J.Pipeline(gpu_ctx, max_images=8, depth=2, host_threads=2)
    outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
    t = pipe.submit(jp, outs, pts, opts)
    st = pipe.wait(t)
    _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, names)
    for i, n in enumerate(names):
        h = J.PreparedImage(jp[i])
        idx, dc, flen = pipe.read_index(t, i, h.n_blocks)
        assert J.index_equivalent(idx, h.block_index()[0]) and np.array_equal(dc, h.block_dc()), n
        assert h.n_blocks * 256 // max(len(h.scan()), 1) > 150, n
        h.close()
    assert pipe.stats["device_images"] == len(jp) and pipe.stats["host_path_images"] == 0
    pipe.close()
    for o in outs:
        gpu_ctx.free(o[0])


def test_pipeline_table_assignments_of_the_components(gpu_ctx, oracle):
    """Components that share a DC table and not their AC table, crossed assignments (DC 0 with AC 1): the walk's DC entries take the
    first AC symbol along only where one AC table follows them (jda_wt_dc_follow); device index == the serial one, pixels == the oracle's."""
    from jpegdec_amd.synth import encode_jpeg_custom
    rng = np.random.default_rng(5)
    img = np.clip(rng.normal(128, 40, (200, 328, 3)) + np.linspace(0, 60, 328)[None, :, None], 0, 255).astype(np.uint8)
    ids = [((0, 0), (0, 1), (1, 1)), ((0, 1), (1, 0), (1, 0)), ((1, 1), (0, 0), (0, 1)), ((0, 0), (1, 1), (1, 1))]
    jp = [encode_jpeg_custom(img, 80, (2, 2) if i % 2 == 0 else (1, 1), table_ids=t) for i, t in enumerate(ids)]
    names = ["ids_%d" % i for i in range(len(jp))]
    pts, opts = [J.RGB8888] * len(jp), [0] * len(jp)
    pipe = J.Pipeline(gpu_ctx, max_images=8, depth=2, host_threads=2)
    outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
    t = pipe.submit(jp, outs, pts, opts)
    st = pipe.wait(t)
    _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, names)
    for i, n in enumerate(names):
        h = J.PreparedImage(jp[i])
        idx, dc, flen = pipe.read_index(t, i, h.n_blocks)
        assert J.index_equivalent(idx, h.block_index()[0]) and np.array_equal(dc, h.block_dc()), n
        h.close()
    assert pipe.stats["device_images"] == len(jp) and pipe.stats["host_path_images"] == 0
    pipe.close()
    for o in outs:
        gpu_ctx.free(o[0])


def test_pipeline_random_shapes_qualities_and_restart_intervals(gpu_ctx, oracle):
    """A sweep the fixed cases do not cover: 60 files of random size (1-700 pixels a side, ragged edges), layout, quality 20-100 and
    restart interval through ONE pipeline in two batches, four pixel types and scales mixed: statuses and pixels == the oracle's, and the
    device-made index == the serial one for every file the device indexed."""
    from jpegdec_amd.synth import encode_jpeg_custom, value_noise_image
    rng = np.random.default_rng(20260925)
    layouts = [(2, 2), (1, 1), (2, 1), (1, 2)]
    jp, names = [], []
    for i in range(60):
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        hv = layouts[int(rng.integers(0, 4))]
        q = int(rng.integers(20, 101))
        ri = 0 if rng.random() < 0.5 else int(rng.integers(1, 40))
        jp.append(encode_jpeg_custom(value_noise_image(w, h, 3, 1000 + i), q, hv, restart_interval=ri))
        names.append("%dx%d_%d%d_q%d_ri%d" % (w, h, hv[0], hv[1], q, ri))
    modes = [(J.RGB8888, 0), (J.RGB565_LE, J.SCALE_HALF), (J.GRAY8, 0), (J.RGB565_BE, 0), (J.RGB8888, J.SCALE_EIGHTH), (J.RGB565_LE, J.SCALE_QUARTER)]
    pipe = J.Pipeline(gpu_ctx, max_images=64, depth=2, host_threads=4)
    inflight = []
    for b in range(2):
        pts, opts = [], []
        for i, n in enumerate(names):
            pt, opt = modes[(i + b) % len(modes)]
            if "_12_" in n and pt == J.RGB8888 and (opt & 4):          # (4:4:0 -> RGB8888 at 1/4: the reference writes through a stray pointer)
                opt = 0
            pts.append(pt); opts.append(opt)
        outs, metas = _surfaces(gpu_ctx, jp, pts, opts)
        inflight.append((pipe.submit(jp, outs, pts, opts), pts, opts, outs, metas))
    indexed = 0
    for t, pts, opts, outs, metas in inflight:
        st = pipe.wait(t)
        _check(gpu_ctx, oracle, jp, pts, opts, outs, metas, st, names)
        for i, n in enumerate(names):
            h = J.PreparedImage(jp[i])
            try:
                idx, dc, flen = pipe.read_index(t, i, h.n_blocks)
            except J.JdaError:
                h.close()
                continue                                               # (the serial path took it: one restart interval, ...)
            if flen == len(h.scan()) and idx[-1] != 0:
                assert J.index_equivalent(idx, h.block_index()[0]) and np.array_equal(dc, h.block_dc()), n
                indexed += 1
            h.close()
        for o in outs:
            gpu_ctx.free(o[0])
    assert indexed >= 60
    pipe.close()


def test_pipeline_page_locked_input_is_read_where_it_lies(gpu_ctx, oracle):
    """JDA_SUBMIT_PINNED_INPUT: the files lie in page-locked memory (J.PinnedFiles = jda_host_alloc) and the copy engine takes the
    large ones (>= 128 KB of scan) from there, the small ones still go through the mirror -- a batch that mixes both, a damaged large
    file (its redo through the serial path reads the caller's buffer again) and the reference's fixtures: pixels, statuses and h2d
    byte counts as with pageable input."""
    rng = np.random.default_rng(5)
    from jpegdec_amd.synth import synth_jpeg
    big = [synth_jpeg(1600, 1200, "4:2:0", seed=70 + k, quality=92) for k in range(3)]      # ~ 400-500 KB each: direct
    assert all(len(b) > (160 << 10) for b in big)
    bad = bytearray(big[0])
    at = len(bad) // 2
    bad[at:at + 64] = bytes(rng.integers(0, 255, 64, dtype=np.uint8))
    small = [jpeg_for(n) for n in sorted(SYNTH_CASES)[:8]] + [ref_jpeg(n) for n in ("tulips", "zebra")]
    files = [big[0], small[0], big[1], bytes(bad)] + small[1:] + [big[2]]
    names = ["big0", "s0", "big1", "bad"] + ["s%d" % i for i in range(1, len(small))] + ["big2"]
    pts = [J.RGB8888 if not (J.parse(f)["subsample"] == 0) else J.GRAY8 for f in files]
    opts = [0] * len(files)
    pinned = J.PinnedFiles(files)
    # one page-locked object of exactly the file's size per file (jda_host_alloc and jda_host_register in turn): a copy command
    # covers bytes of ONE of them, from inside the file to its last byte
    separate = J.PinnedFiles(files, separate=True)
    assert any(ln % 16 for ln in separate.lens)
    pipe = J.Pipeline(gpu_ctx, max_images=len(files), depth=2, host_threads=2)
    results, h2d = [], []
    for mode in ("pageable", "pinned", "pinned", "separate", "separate", "unknown"):
        outs, metas = _surfaces(gpu_ctx, files, pts, opts)
        before = pipe.stats["h2d_bytes"]
        if mode == "pageable":
            t = pipe.submit(files, outs, pts, opts)
        elif mode == "unknown":      # the flag over memory the library was never told of: such files take the mirror
            t = pipe.submit_packed(pipe.pack(files, outs, pts, opts), J.SUBMIT_PINNED_INPUT)
        else:
            t = pipe.submit_packed(pipe.pack_pinned(pinned if mode == "pinned" else separate, list(range(len(files))), outs, pts, opts), J.SUBMIT_PINNED_INPUT)
        st = pipe.wait(t)
        _check(gpu_ctx, oracle, files, pts, opts, outs, metas, st, names)
        results.append(st)
        h2d.append(pipe.stats["h2d_bytes"] - before)
        for o in outs:
            gpu_ctx.free(o[0])
    assert all(r == results[0] for r in results)
    assert h2d[5] == h2d[0] and abs(h2d[3] - h2d[0]) < 4096 * len(files), h2d
    pipe.close()
    pinned.close()
    separate.close()
