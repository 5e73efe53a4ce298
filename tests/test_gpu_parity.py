"""GPU parity tests: the HIP path, called through the C-ABI, must be bit-exact to the oracle.

Checker = oracle/liboracle.so (our restatement, itself pinned to the real reference by
tests/test_oracle_*.py) and, where it travelled along, oracle/_ref (the real reference's scalar
build).  Nothing here reads /root/reference."""
import numpy as np
import pytest

import jpegdec_amd as J
from tests.cases import SYNTH_CASES, all_modes, jpeg_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_decode_bit_exact_all_modes(name, gpu_ctx, oracle):
    jpeg = jpeg_for(name)
    for pt, opt in all_modes(name):
        rc, got, g = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
        assert rc == 0, (name, pt, opt, rc)
        orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert orc == 1, (name, pt, opt, err)
        assert got.shape == want.shape
        assert np.array_equal(got, want), "%s pt=%d opt=%d: %d differing bytes" % (
            name, pt, opt, int(np.count_nonzero(got != want)))


@pytest.mark.parametrize("name", ["p420_200x120", "p444_333x217", "p422_640x368", "pgray_100x100", "p420_1280x720_q95"])
def test_progressive_thumbnails(name, gpu_ctx, oracle):
    """a progressive file is decoded from its first (DC) scan as a 1/8 thumbnail (1/2 when asked for):
    DC symbols only, differences shifted by Al, every block through the DC bypass -- bit-exact with the oracle
    (which tests/test_progressive.py pins to the real reference).  What the reference cannot do is refused."""
    from tests.cases import progressive_modes
    jpeg = jpeg_for(name)
    for pt, opt in progressive_modes(name):
        rc, got, g = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
        assert rc == 0, (name, pt, opt, rc)
        orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert orc == 1
        assert got.shape == want.shape and np.array_equal(got, want), (name, pt, opt)
    if not name.startswith("pgray"):
        with pytest.raises(J.JdaError) as e:
            J.decode_to_host(gpu_ctx, jpeg, J.GRAY8, 0)
        assert e.value.code == 3


@pytest.mark.parametrize("w,h", [(2304, 24), (2048, 8), (4160, 16), (1600, 16)])
def test_wide_gray_thumbnails(w, h, gpu_ctx, oracle):
    """1/8 and 1/4 of gray files wide enough for the kernels' whole-tile paths (jda_dc_thumbnail: four whole tiles side by side as packed
    16-bit products, a dword per lane; jda_quarter_tiles: two lanes share a store), with the ragged tiles at the right edge beside them,
    and the same files cut short by a damaged byte (the MCUs behind the bad one stay as they were, jpeg.inl:5354-5356)."""
    from jpegdec_amd.synth import synth_jpeg
    jpeg = synth_jpeg(w, h, "gray", seed=w)
    bad = bytearray(jpeg)
    sos = jpeg.index(b"\xff\xda")
    bad[sos + 14 + (len(jpeg) - sos) * 2 // 3] ^= 0x5a
    for jb in (jpeg, bytes(bad)):
        for pt, opt in ((J.GRAY8, J.SCALE_EIGHTH), (J.GRAY8, J.SCALE_QUARTER), (J.RGB565_LE, J.SCALE_EIGHTH), (J.RGB565_BE, J.SCALE_QUARTER)):
            orc, want, err = oracle.decode_canvas(jb, pt, opt)
            rc, got, g = J.decode_to_host(gpu_ctx, jb, pt, opt)
            assert (rc == 0) == (orc == 1), (w, h, pt, opt, rc, orc, err)
            if orc == 1:
                assert np.array_equal(got, want), (w, h, pt, opt, int(np.count_nonzero(got != want)))


def test_matches_real_reference_when_present(gpu_ctx, ref_scalar):
    """Same comparison against the unmodified reference (scalar integer build) if oracle/_ref travelled."""
    for name in ("c420_333x217", "c444_256x256_q100_opt", "gray_64x64_rst3", "c420_1280x720"):
        jpeg = jpeg_for(name)
        for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, 0), (J.RGB565_BE, J.SCALE_HALF), (J.GRAY8, J.SCALE_QUARTER),
                        (J.RGB8888, J.SCALE_EIGHTH)):
            if name.startswith("gray") and pt == J.RGB8888:
                continue
            rc, got, g = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
            assert rc == 0
            r = ref_scalar.decode_cb(jpeg, pt, opt)
            assert r["rc"] == 1
            want = r["canvas"][: g["out_h"], : g["canvas_w"] * g["bpp"]]
            assert np.array_equal(got[: g["out_h"]], want), (name, pt, opt)


def test_batch_of_mixed_images_resident(gpu_ctx, oracle):
    """One launch plan over several resident images of different shapes and output formats."""
    # (every kernel variant the runtime routes to gets at least one image: the general kernel and the plain-case ones for
    # RGB8888 / RGB565 / 8-bit gray, several launch lists per decode)
    names = ["c420_333x217", "c444_333x217", "gray_333x217", "c420_1100x48", "c420_640x368_rstrow",
             "c420_333x217", "c420_1280x720", "c444_333x217", "gray_1100x24", "c444_600x16", "c422_333x217", "c420_1100x48", "c440_200x120"]
    pts = [J.RGB8888, J.RGB565_LE, J.GRAY8, J.RGB8888, J.RGB565_BE,
           J.RGB565_LE, J.GRAY8, J.RGB565_LE, J.GRAY8, J.RGB8888, J.RGB8888, J.GRAY8, J.GRAY8]
    opts = [0, J.SCALE_HALF, 0, J.SCALE_QUARTER, 0,
            0, 0, 0, 0, 0, 0, J.SCALE_HALF, J.SCALE_HALF]
    prepared = [J.PreparedImage(jpeg_for(n)) for n in names]
    dev = [J.DeviceImage(gpu_ctx, p) for p in prepared]
    outs, ptrs, geos = [], [], []
    for p, pt, opt in zip(prepared, pts, opts):
        g = p.geometry(pt, opt)
        pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
        ptr = gpu_ctx.malloc(pitch * g["canvas_h"])
        gpu_ctx.memset(ptr, 0, pitch * g["canvas_h"])
        outs.append((ptr, pitch, g["canvas_w"], g["canvas_h"]))
        ptrs.append(ptr)
        geos.append((g, pitch))
    batch = J.Batch(gpu_ctx, dev, outs, pts, opts)
    batch.decode()
    batch.decode()          # idempotent: decoding twice into the same surface changes nothing
    gpu_ctx.sync()
    for n, pt, opt, ptr, (g, pitch) in zip(names, pts, opts, ptrs, geos):
        got = gpu_ctx.to_host(ptr, pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]]
        rc, want, _ = oracle.decode_canvas(jpeg_for(n), pt, opt)
        assert rc == 1 and np.array_equal(got, want), n
    batch.close()
    for d in dev:
        d.close()
    for ptr in ptrs:
        gpu_ctx.free(ptr)


def test_many_small_images_in_one_launch(gpu_ctx, oracle):
    """Hundreds of small images with different Huffman / quantiser tables in one launch: a workgroup's run of tiles then
    spans many images, its wavefronts draw tiles across image boundaries (tables restaged by whichever wavefront drew the
    new image's first tile, wavefronts without tiles keeping the barriers company) -- every image must still come out
    exactly, in the general kernel (mixed formats) and in the plain-case variant (RGB8888 full size)."""
    names420 = ["c420_16x16", "c420_250x250_q10", "c420_333x217", "c420_256x256_q98", "c420_1100x48"]
    names444 = ["c444_8x8_q30", "c444_600x16", "c444_256x256_q100_opt", "c444_333x217"]
    for names, variants in ((names420, "plain"), (names420, "mixed"), (names444, "plain"), (names420, "runs")):
        n = 240
        seq = [names[(i * 7 + i // 5) % len(names)] for i in range(n)]
        if variants == "runs":
            # runs of images with EQUAL tables (the same file, and different files of one quality): one table generation to the kernel
            # (jda_batch_create_strips), between runs a new one
            seq, k = [], 0
            while len(seq) < n:
                seq += [names[(k * 3) % len(names)]] * (1 + (k * 5) % 7); k += 1
            seq = seq[:n]
        pts = [J.RGB8888 if variants in ("plain", "runs") else (J.RGB8888, J.RGB565_LE, J.RGB565_BE)[i % 3] for i in range(n)]
        opts = [0 if variants in ("plain", "runs") else (0, J.SCALE_HALF, 0, J.SCALE_EIGHTH)[i % 4] for i in range(n)]
        prep = {nm: J.PreparedImage(jpeg_for(nm)) for nm in names}
        devs = {nm: J.DeviceImage(gpu_ctx, prep[nm]) for nm in names}
        outs, ptrs, geos = [], [], []
        for nm, pt, opt in zip(seq, pts, opts):
            g = prep[nm].geometry(pt, opt)
            pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
            ptr = gpu_ctx.malloc(pitch * g["canvas_h"])
            outs.append((ptr, pitch, g["canvas_w"], g["canvas_h"]))
            ptrs.append(ptr); geos.append((g, pitch))
        batch = J.Batch(gpu_ctx, [devs[nm] for nm in seq], outs, pts, opts)
        batch.decode(); gpu_ctx.sync()
        want_cache = {}
        for i, (nm, pt, opt, ptr, (g, pitch)) in enumerate(zip(seq, pts, opts, ptrs, geos)):
            got = gpu_ctx.to_host(ptr, pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]]
            key = (nm, pt, opt)
            if key not in want_cache:
                rc, want, _ = oracle.decode_canvas(jpeg_for(nm), pt, opt)
                assert rc == 1
                want_cache[key] = want
            assert np.array_equal(got, want_cache[key]), (variants, i, nm, pt, opt)
        batch.close()
        for ptr in ptrs:
            gpu_ctx.free(ptr)
        for d in devs.values():
            d.close()


def test_clip_to_image_size(gpu_ctx, oracle):
    """width_px / rows clip: writing only W x H pixels leaves the rest of the surface untouched."""
    jpeg = jpeg_for("c420_333x217")
    p = J.PreparedImage(jpeg)
    d = J.DeviceImage(gpu_ctx, p)
    g = p.geometry(J.RGB8888, 0)
    pitch = (g["canvas_w"] * 4 + 15) & ~15
    ptr = gpu_ctx.malloc(pitch * g["canvas_h"])
    gpu_ctx.memset(ptr, 0x5A, pitch * g["canvas_h"])
    b = J.Batch(gpu_ctx, [d], [(ptr, pitch, g["out_w"], g["out_h"])], [J.RGB8888], [0])
    b.decode()
    gpu_ctx.sync()
    got = gpu_ctx.to_host(ptr, pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)
    rc, want, _ = oracle.decode_canvas(jpeg, J.RGB8888, 0)
    assert np.array_equal(got[: g["out_h"], : g["out_w"] * 4], want[: g["out_h"], : g["out_w"] * 4])
    assert (got[g["out_h"]:, :] == 0x5A).all() and (got[:, g["out_w"] * 4:] == 0x5A).all()
    b.close(); d.close(); gpu_ctx.free(ptr)


@pytest.mark.parametrize("name", ["c420_640x368_rstrow", "gray_64x64_rst3", "c444_384x192_q100_rst7", "c420_512x256_q98_rstrow",
                                  "c422_1100x24_rstrow", "c440_300x64_rst5"])
def test_restart_marker_fast_path(name, gpu_ctx, oracle):
    """JDA_PREPARE_DEVICE_PRESCAN leaves the Huffman pre-scan of a stream with restart markers to
    the GPU (jda_upload: the segment walk's RST flavour -- interval ends found by position, sums restart behind
    them, every marker checked against the MCU count).  Same pixels as the oracle, and the path really was taken."""
    import jpegdec_amd as J
    jpeg = jpeg_for(name)
    prep = J.PreparedImage(jpeg, device_prescan=True)
    assert prep.prescan_pending
    dimg = J.DeviceImage(gpu_ctx, prep)
    assert dimg.prescan_on_device and not prep.prescan_pending
    for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, 0), (J.GRAY8, J.SCALE_HALF)):
        if name.startswith("gray") and pt == J.RGB8888:
            continue
        rc, want, _ = oracle.decode_canvas(jpeg, pt, opt)
        g = prep.geometry(pt, opt)
        pitch = (want.shape[1] + 15) // 16 * 16
        out = gpu_ctx.malloc(pitch * want.shape[0])
        b = J.Batch(gpu_ctx, [dimg], [(out, pitch, g["canvas_w"], g["canvas_h"])], [pt], [opt])
        b.decode(); gpu_ctx.sync()
        got = gpu_ctx.to_host(out, pitch * want.shape[0]).reshape(want.shape[0], pitch)[:, : want.shape[1]]
        assert np.array_equal(got, want), (name, pt, opt)
        gpu_ctx.free(out)
    # several images at once (jda_upload_batch): markers and no markers mixed
    preps = [J.PreparedImage(jpeg_for(nm), device_prescan=True) for nm in (name, "c420_333x217", name)]
    dimgs = J.upload_batch(gpu_ctx, preps)
    assert [d.prescan_on_device for d in dimgs] == [True, True, True]     # (the marker-less one by the segment walk, 8f N2)
    for p_, d_, nm in zip(preps, dimgs, (name, "c420_333x217", name)):
        pt = J.GRAY8 if nm.startswith("gray") else J.RGB8888
        rc, want, _ = oracle.decode_canvas(jpeg_for(nm), pt, 0)
        g = p_.geometry(pt, 0)
        pitch = (want.shape[1] + 15) // 16 * 16
        out = gpu_ctx.malloc(pitch * want.shape[0])
        b = J.Batch(gpu_ctx, [d_], [(out, pitch, g["canvas_w"], g["canvas_h"])], [pt], [0])
        b.decode(); gpu_ctx.sync()
        got = gpu_ctx.to_host(out, pitch * want.shape[0]).reshape(want.shape[0], pitch)[:, : want.shape[1]]
        assert np.array_equal(got, want), (nm, "batch upload")
        gpu_ctx.free(out)


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "gray_333x217", "c420_256x256_q98", "c444_256x256_q100_opt",
                                  "c420_1100x48", "gray_1600x16", "c420_16x16", "c444_8x8_q30", "c420_1280x720", "c422_333x217",
                                  "c440_200x120", "c420_250x250_q10"])
def test_markerless_device_prescan(name, gpu_ctx, oracle):
    """JDA_PREPARE_DEVICE_PRESCAN on a stream WITHOUT restart markers: the per-block index is made on the GPU
    by the segment walk (jda_segscan: speculative rounds, count, write).  The index in HBM must equal the serial host
    pre-scan's byte for byte (reader phase of every block, DC predictors, closing entry), and the pixels the oracle's."""
    import jpegdec_amd as J
    jpeg = jpeg_for(name)
    prep = J.PreparedImage(jpeg, device_prescan=True)
    assert prep.prescan_pending
    dimg = J.DeviceImage(gpu_ctx, prep)
    assert dimg.prescan_on_device and not prep.prescan_pending
    assert 1 <= gpu_ctx.lib.jda_last_prescan_rounds(gpu_ctx.handle) <= 24
    host = J.PreparedImage(jpeg)                       # serial pre-scan
    want_idx, nok = host.block_index()
    got_idx, got_dc = dimg.read_index()
    assert J.index_equivalent(got_idx, want_idx)      # (bit position + flag of every block, the whole entry of a flagged one)
    assert np.array_equal(got_dc, host.block_dc())
    for pt, opt in ((J.RGB8888, 0), (J.RGB565_LE, J.SCALE_HALF), (J.GRAY8, J.SCALE_EIGHTH)):
        if name.startswith("gray") and pt == J.RGB8888:
            continue
        rc, want, _ = oracle.decode_canvas(jpeg, pt, opt)
        g = prep.geometry(pt, opt)
        pitch = (want.shape[1] + 15) // 16 * 16
        out = gpu_ctx.malloc(pitch * want.shape[0])
        b = J.Batch(gpu_ctx, [dimg], [(out, pitch, g["canvas_w"], g["canvas_h"])], [pt], [opt])
        b.decode(); gpu_ctx.sync()
        got = gpu_ctx.to_host(out, pitch * want.shape[0]).reshape(want.shape[0], pitch)[:, : want.shape[1]]
        assert np.array_equal(got, want), (name, pt, opt)
        b.close(); gpu_ctx.free(out)
    dimg.close()


def test_markerless_device_prescan_batch_and_fallback(gpu_ctx, oracle):
    """Many images of different sizes in one jda_upload_batch (one launch per pass for all of them), and corrupted streams
    among them: those must come out exactly as through the serial pre-scan (the device walk hands them back)."""
    import jpegdec_amd as J
    names = ["c420_1280x720", "c444_8x8_q30", "gray_1600x16", "c420_333x217", "c444_256x256_q100_opt", "c420_16x16"]
    jpegs = [jpeg_for(n) for n in names]
    base = bytearray(jpeg_for("c420_333x217"))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(5)
    for it in range(10):                                 # corrupted variants
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
        try:
            J.PreparedImage(bytes(b)).close()
        except J.JdaError:
            continue
        jpegs.append(bytes(b))
    preps = [J.PreparedImage(j, device_prescan=True) for j in jpegs]
    dimgs = J.upload_batch(gpu_ctx, preps)
    assert all(d.prescan_on_device for d in dimgs[: len(names)])
    for j, p_, d_ in zip(jpegs, preps, dimgs):
        host = J.PreparedImage(j)
        want_idx, nok = host.block_index()
        got_idx, got_dc = d_.read_index()
        nb = nok * p_.info.blocks_per_mcu
        assert J.index_equivalent(got_idx[:nb], want_idx[:nb])
        assert np.array_equal(got_dc[:nb], host.block_dc()[:nb])
        assert d_.n_mcus_ok == nok
        d_.close(); host.close()


def test_corrupted_scans_on_the_gpu(gpu_ctx, oracle):
    """A few corrupted streams through the real kernels (the CPU suite runs many more through the wave emulator):
    same verdict as the oracle, same pixels where it decodes, and nothing hangs."""
    checked = 0
    for name in ("c420_333x217", "c422_333x217", "c420_640x368_rstrow"):
        base = bytearray(jpeg_for(name))
        sos = bytes(base).index(b"\xff\xda")
        rng = np.random.default_rng(31)
        for it in range(12):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
            jb = bytes(b)
            try:
                p = J.PreparedImage(jb)
            except J.JdaError:
                continue
            idx, nok = p.block_index()
            over_read = (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan())
            rc, got, g = J.decode_to_host(gpu_ctx, jb, J.RGB8888, 0)
            orc, want, err = oracle.decode_canvas(jb, J.RGB8888, 0)
            assert rc in (0, 2)
            if not over_read:
                assert (rc == 0) == (orc == 1), (name, it, rc, orc, err)
                if orc == 1:
                    assert np.array_equal(got, want), (name, it)
                    checked += 1
    assert checked >= 5


def test_marker_filter_on_the_gpu(gpu_ctx, oracle):
    """JPEGFilter (jpeg.inl:1431-1540) as a scan of two-state transition functions on the GPU (jda_filter_scan): byte for byte
    the host filter's output, restart-marker offsets included -- on real scans, and on synthetic byte streams dense with FF
    runs of every length and phase (also across the 16-byte thread and 16 KB step boundaries, and at the very end)."""
    import jpegdec_amd as J
    for name in ("c420_333x217", "c420_640x368_rstrow", "c444_384x192_q100_rst7", "gray_64x64_rst3", "c420_1280x720"):
        jpeg = jpeg_for(name)
        p = J.PreparedImage(jpeg)
        raw = jpeg[p.info.scan_offset:]
        got, rpos, nr = J.filter_on_device(gpu_ctx, raw)
        assert got == p.scan().tobytes() == oracle.filter(raw), name
        assert np.array_equal(rpos, _host_restart_positions(raw)), name
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(0, 70000)) if trial else 0
        raw = rng.integers(0, 256, n, dtype=np.uint8)
        # salt with FF runs, FF 00, RSTn markers; some exactly at 16-byte / 16 KB boundaries
        for _ in range(n // 37):
            at = int(rng.integers(0, max(n - 8, 1)))
            kind = int(rng.integers(0, 5))
            run = int(rng.integers(1, 7))
            raw[at:at + run] = 0xFF
            if kind == 0 and at + run < n: raw[at + run] = 0x00
            if kind == 1 and at + run < n: raw[at + run] = 0xD0 + int(rng.integers(0, 8))
        for edge in (15, 16, 16383, 16384, 32767, n - 1, n - 2):
            if 0 <= edge < n and rng.integers(0, 2): raw[edge] = 0xFF
        rawb = raw.tobytes()
        got, rpos, nr = J.filter_on_device(gpu_ctx, rawb)
        assert got == oracle.filter(rawb), (trial, n)
        assert np.array_equal(rpos, _host_restart_positions(rawb)), (trial, n)
    # runs of FF as long as a thread's sixteen bytes, a wavefront's 1,024, a workgroup's 16 KB and more, at every alignment class, of
    # either parity, followed by 00 / a marker / a plain byte / the end: a thread (wavefront, chunk) of nothing but FF passes the state
    # on flipped by its parity -- the kernels take its incoming state from the nearest thread in front that holds another byte
    for trial, (lead, run, tail) in enumerate([(0, 16, b"\x00"), (5, 16, b"\xd3"), (16, 17, b"\x41"), (3, 32, b"\x00"), (11, 33, b"\xd0\x12"),
                                               (0, 1024, b"\x00"), (7, 1025, b"\x00"), (1000, 2049, b"\xd7"), (16383, 17, b"\x00"),
                                               (9, 16384, b"\x00"), (16384, 16385, b"\x41"), (100, 32769, b"\xd1\x00"), (0, 40000, b""),
                                               (0, 40001, b""), (15, 1, b""), (16, 1, b""), (0, 0, b"\xff"), (16383, 1, b"\x00\xff"), (16384, 15, b"")]):
        rawb = bytes(rng.integers(0, 255, lead, dtype=np.uint8)) + b"\xff" * run + tail + bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * (1 if tail else 0)
        got, rpos, nr = J.filter_on_device(gpu_ctx, rawb)
        assert got == oracle.filter(rawb), ("long run", trial, lead, run)
        assert np.array_equal(rpos, _host_restart_positions(rawb)), ("long run", trial, lead, run)
    for trial in range(60):                                       # many runs of sixteen and more in one buffer, runs meeting chunk edges
        n = int(rng.integers(20000, 90000))
        raw = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(int(rng.integers(3, 40))):
            run = int(rng.choice([16, 17, 31, 32, 48, 63, 64, 65, 128, 1023, 1024, 1025, 4096, 16384, 16385]))
            at = int(rng.choice([rng.integers(0, n), 16384 - run // 2, 16384 - run, 32768 - 1, 16384])) % max(n - 1, 1)
            raw[at:at + run] = 0xFF
            end = min(at + run, n - 1)
            raw[end] = int(rng.choice([0x00, 0xD0 + int(rng.integers(0, 8)), 0x37, 0xFF]))
        rawb = raw.tobytes()
        got, rpos, nr = J.filter_on_device(gpu_ctx, rawb)
        assert got == oracle.filter(rawb), ("many runs", trial, n)
        assert np.array_equal(rpos, _host_restart_positions(rawb)), ("many runs", trial, n)


def _host_restart_positions(raw: bytes):
    """what jda_prepare's filter records: the filtered offset of every RSTn marker, preceded by 0"""
    pos, o, i, n = [0], 0, 0, len(raw)
    while i < n:
        if raw[i] != 0xFF:
            o += 1; i += 1
        else:
            if i + 1 < n and raw[i + 1] == 0:
                o += 1
            elif i + 1 < n and (raw[i + 1] & 0xF8) == 0xD0:
                pos.append(o)
            i += 2
    return np.array(pos, np.uint32)


@pytest.mark.parametrize("luma_hv", [(2, 2), (1, 1), (2, 1), (1, 2)])
def test_duplicate_eob_code_general_reader(luma_hv, gpu_ctx, oracle):
    """An AC table that codes the end-of-block symbol twice (malformed DHT; the reference's per-code LUTs decode it):
    P1's one-compare EOB test cannot be used, the host flags the image (JDA_DESC_GENERAL_P1) and the kernels take their
    general bit reader.  Same bytes as the oracle (pinned to the real reference on these files by
    tests/test_oracle_vs_ref.py); a device pre-scan makes the same index."""
    from jpegdec_amd.synth import encode_jpeg_custom, value_noise_image
    jpeg = encode_jpeg_custom(value_noise_image(333, 217, 3, 78), 85, luma_hv, dup_eob=True)
    prep = J.PreparedImage(jpeg)
    assert prep.general_p1()
    prep.close()
    for pt, opt in ((J.RGB8888, 0), (J.RGB565_BE, 0), (J.RGB8888, J.SCALE_HALF), (J.GRAY8, J.SCALE_QUARTER), (J.RGB565_LE, J.SCALE_EIGHTH)):
        if luma_hv == (1, 2) and pt == J.RGB8888 and opt == J.SCALE_QUARTER:
            continue
        rc, got, g = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
        assert rc == 0
        orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert orc == 1 and np.array_equal(got, want), (luma_hv, pt, opt)
    dev = J.PreparedImage(jpeg, device_prescan=True)
    host = J.PreparedImage(jpeg)
    dimg = J.DeviceImage(gpu_ctx, dev)
    assert J.index_equivalent(dimg.read_index()[0], host.block_index()[0])
    dimg.close(); dev.close(); host.close()


def test_one_bad_file_does_not_poison_a_resident_batch(gpu_ctx, oracle):
    """VERDICT r1 task 7 on the resident path: jda_prepare_batch -> jda_upload_batch_ex -> jda_batch_create over 16 files of
    which one is rejected by the parser and one has a corrupt scan: 14 decode bit-exact, the hole and the partial decode are
    flagged per image (jda_batch_get_status), nobody else notices."""
    good = jpeg_for("c420_333x217")
    broken_scan = bytearray(good)
    sos = bytes(broken_scan).index(b"\xff\xda")
    broken_scan = bytes(broken_scan[: sos + 14 + 900])            # the stream ends early: bad MCU somewhere in the middle
    files = [good] * 16
    files[3] = good[:100]                                          # not a JPEG any more (JPEG_INVALID_FILE)
    files[9] = broken_scan
    prepared, errs = J.prepare_batch(files, threads=4, strict=False)
    assert [p is None for p in prepared] == [i == 3 for i in range(16)] and errs[3] == 4
    dev = J.upload_batch(gpu_ctx, prepared)
    assert [d is None for d in dev] == [i == 3 for i in range(16)]
    g = prepared[0].geometry(J.RGB8888, 0)
    pitch = (g["canvas_w"] * 4 + 15) & ~15
    base = gpu_ctx.malloc(pitch * g["canvas_h"] * 16)
    gpu_ctx.memset(base, 0, pitch * g["canvas_h"] * 16)
    outs = [(base + i * pitch * g["canvas_h"], pitch, g["canvas_w"], g["canvas_h"]) for i in range(16)]
    batch = J.Batch(gpu_ctx, dev, outs, [J.RGB8888] * 16, [0] * 16)
    batch.decode(); gpu_ctx.sync()
    assert batch.status() == [1 if i == 3 else (2 if i == 9 else 0) for i in range(16)]
    rc, want, _ = oracle.decode_canvas(good, J.RGB8888, 0)
    for i in range(16):
        got = gpu_ctx.to_host(outs[i][0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * 4]
        if i not in (3, 9):
            assert np.array_equal(got, want), i
    nok = prepared[9].block_index()[1]
    assert 0 < nok < prepared[9].n_mcus
    got = gpu_ctx.to_host(outs[9][0], pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * 4]
    rows = (nok // prepared[9].info.mcus_x) * 16
    assert np.array_equal(got[:rows], want[:rows])                 # the MCU rows in front of the bad one are the good image's
    batch.close()
    for d in dev:
        if d is not None:
            d.close()
    gpu_ctx.free(base)


def test_p1_in_chunks_is_bit_exact(gpu_ctx, oracle):
    """P1's chunked mode (jda_p1c_*: the lanes of a wavefront share a tile's long blocks through the index's continuation entries,
    one every 8 AC symbols) asked for on every image (JDA_PREPARE_CONT_ALWAYS; by default the serial pre-scan writes entries between
    56 and 112 bits of scan per block): every synthetic case and the reference's photographs, to RGB8888 (4:2:0 and 4:4:4 have a
    chunked kernel) and to the formats that keep their kernels -- bit-exact with the oracle.  The entries themselves: every one
    names its block and a place inside it."""
    from tests.ref_fixtures import GOOD, ref_jpeg
    cases = [(n, jpeg_for(n)) for n in sorted(SYNTH_CASES)] + [("ref:" + n, ref_jpeg(n)) for n in GOOD]
    n_entries = 0
    before = J.kernel_launch_counts()
    for name, jpeg in cases:
        p = J.PreparedImage(jpeg, flags=J.PREPARE_CONT_ALWAYS)
        first, ent = p.block_cont()
        assert first[0] == 0 and first[-1] == len(ent) and np.all(np.diff(first.astype(np.int64)) >= 0)
        owner = np.repeat(np.arange(p.n_blocks), np.diff(first.astype(np.int64)))
        assert np.array_equal((ent >> 18) & 127, owner & 127) and np.all(((ent >> 12) & 63) >= 9), name   # (eight symbols in: coefficient 9 at the least)
        n_entries += len(ent)
        gray = p.info.ncomp == 1
        for pt, opt in ((J.RGB565_BE, 0), (J.GRAY8, 0)) if gray else ((J.RGB8888, 0), (J.RGB565_BE, 0), (J.RGB8888, J.SCALE_HALF)):
            st, got, g = J.decode_resident(gpu_ctx, p, pt, opt)
            orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
            assert (st == 0) == (orc == 1), (name, pt, opt, st, orc)
            if orc == 1:
                assert np.array_equal(got, want), (name, pt, opt, int(np.count_nonzero(got != want)))
        p.close()
    assert n_entries > 50000
    after = J.kernel_launch_counts()
    chunked = [k for k in after if "persistent" in k and k.endswith("Li1EEvPK12jda_dev_descPK9jda_stripjj") and after[k] > before.get(k, 0)]
    assert len(chunked) == 2, (chunked, sorted(after))           # the 4:2:0 and the 4:4:4 kernel both ran
