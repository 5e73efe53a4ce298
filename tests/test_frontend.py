"""Host front end of the product (jpegdec_amd/csrc/jda_frontend.cpp, via the C-ABI; no GPU calls):
parse, Huffman LUTs, scan filter, quant prescale, serial pre-scan index, draw plan -- each against
the oracle's independent restatement of the same reference stage."""
import ctypes as C
import os

import numpy as np
import pytest

import jpegdec_amd as J
from jpegdec_amd.binding import ImageInfo, load_library
from oracle.loader import USES_DMA
from tests.cases import SYNTH_CASES, jpeg_for


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_prepare_matches_oracle_stage_by_stage(name, oracle, product_lib):
    jpeg = jpeg_for(name)
    p = J.PreparedImage(jpeg, flags=J.PREPARE_SERIAL_PRESCAN)      # (the serial pre-scan: the reference reader's phase in EVERY entry; the interval-parallel one: below)
    oi = oracle.info(jpeg)
    assert (p.info.width, p.info.height, p.info.ncomp, p.info.subsample, p.info.restart_interval, p.info.scan_offset) == (
        oi["width"], oi["height"], oi["ncomp"], oi["subsample"], oi["restart_interval"], oi["scan_offset"])
    # filtered scan == JPEGFilter over the whole scan
    assert p.scan().tobytes() == oracle.filter(jpeg[oi["scan_offset"]:])
    # table blob: DC LUTs | AC LUTs | prescaled quant | zigzag
    t = p.tables()
    rc, dc, ac = oracle.huff_tables(jpeg)
    assert rc == 1
    assert np.array_equal(t[:2048], dc)
    assert np.array_equal(t[2048:10240].view(np.uint16), ac)
    rc, q = oracle.quant_tables(jpeg)
    assert np.array_equal(t[10240:10752].view(np.int16).reshape(4, 64), q)
    assert list(t[10752:10816]) == [oracle.lib.orc_zigzag_to_natural(k) for k in range(64)]
    # pre-scan index (format 2) == the oracle's bit-reader phase at every block's first AC symbol + the block's own DC value
    n, coefs, flags, state, dcp = oracle.entropy(jpeg, 0)
    idx, nok = p.block_index()
    assert nok == p.n_mcus and n == len(flags) == p.n_blocks
    # (an entry holds the reader behind the refill at the top of the AC loop, jpeg.inl:2225-2230: offset <= 47; bit 6 flags a block
    # with a truncated magnitude read)
    wpos, woff = oracle.blk_ac_state[:, 0].astype(np.int64), oracle.blk_ac_state[:, 1].astype(np.int64)
    assert woff.max() <= 47
    assert np.array_equal(idx[:-1] >> 7, wpos) and np.array_equal(idx[:-1] & 63, woff)
    assert np.array_equal(p.block_dc(), coefs[:, 0])
    # .. and that DC value is the predictor the next block of its component is entered with (what format 1 stored)
    bpm = p.info.blocks_per_mcu
    comp = np.array([0 if b < bpm - (2 if p.info.ncomp == 3 else 0) else b - (bpm - 2) + 1 for b in range(bpm)])
    if p.info.restart_interval == 0:
        for c in range(3 if p.info.ncomp == 3 else 1):
            sel = np.flatnonzero(np.tile(comp, p.n_mcus) == c)
            assert np.array_equal(oracle.blk_pred[sel][1:], coefs[sel, 0][:-1].astype(np.int32))
    n_flagged = int(np.count_nonzero(idx[:-1] & 64))
    assert n_flagged <= p.truncation_events() and (n_flagged > 0) == (p.truncation_events() > 0)
    p.close()


def test_truncation_events_are_counted(product_lib):
    """high-quality streams hit the un-refilled magnitude read; smooth ones do not."""
    assert J.PreparedImage(jpeg_for("c444_256x256_q100_opt")).truncation_events() > 0


def test_reject_rules(product_lib):
    good = jpeg_for("c420_16x16")
    assert J.parse(good[:100])["status"] == 4                      # JPEG_INVALID_FILE: < 256 bytes (jpeg.inl:1598)
    assert J.parse(b"\x00" * 400)["status"] == 4                   # no SOI (jpeg.inl:1604)
    sof = good.index(b"\xff\xc0")
    bad = bytearray(good); bad[sof + 1] = 0xC1
    assert J.parse(bytes(bad))["status"] == 3                      # SOF1 -> JPEG_UNSUPPORTED_FEATURE (jpeg.inl:1649)
    bad = bytearray(good); bad[sof + 1] = 0xC2
    with pytest.raises(J.JdaError) as e:
        J.PreparedImage(bytes(bad))                                # progressive: off this path
    assert e.value.code == 3
    bad = bytearray(good); bad[sof + 11] = 0x41                    # Y sampling 4x1: reference divides by zero
    with pytest.raises(J.JdaError):
        J.PreparedImage(bytes(bad))
    with pytest.raises(J.JdaError):
        J.PreparedImage(good[: good.index(b"\xff\xda")])           # no SOS -> JPEG_DECODE_ERROR


def test_truncated_scan_reports_partial_index(product_lib):
    jpeg = jpeg_for("c420_333x217")
    cut = jpeg[: len(jpeg) // 2] + b"\xff\xd9" + b"\x00" * 300
    p = J.PreparedImage(cut)
    idx, nok = p.block_index()
    assert 0 < nok <= p.n_mcus


def test_fuzz_does_not_crash(product_lib):
    """The reference's own fuzz loop (MacOS/JPEGDEC_Test/main.cpp:262-281): invert each of the
    first bytes of the file; the front end must fail cleanly or produce an index, never crash."""
    jpeg = bytearray(jpeg_for("c420_333x217"))
    for i in range(0, min(len(jpeg), 700)):
        b = bytearray(jpeg); b[i] ^= 0xFF
        try:
            J.PreparedImage(bytes(b)).close()
        except J.JdaError:
            pass
    rng = np.random.default_rng(5)
    for _ in range(200):
        b = bytearray(jpeg)
        for _ in range(2):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            J.PreparedImage(bytes(b)).close()
        except J.JdaError:
            pass


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "gray_333x217", "c420_1100x48", "c420_16x16"])
def test_draw_plan_matches_oracle(name, oracle, product_lib):
    jpeg = jpeg_for(name)
    info = ImageInfo()
    assert load_library().jda_parse(jpeg, len(jpeg), C.byref(info)) == 0
    for pt in (0, 2, 3):
        for opt in (0, 2, 4, 8, USES_DMA):
            for mm in (0, 3):
                got = J.draw_plan(info, pt, opt & ~USES_DMA, mm, bool(opt & USES_DMA))
                want = oracle.draw_plan(jpeg, pt, opt & ~USES_DMA, mm, bool(opt & USES_DMA))
                assert np.array_equal(got, want), (pt, opt, mm)


def test_output_geometry(product_lib):
    p = J.PreparedImage(jpeg_for("c420_333x217"))
    assert p.geometry(J.RGB8888, 0) == dict(bpp=4, out_w=333, out_h=217, canvas_w=336, canvas_h=224)
    assert p.geometry(J.RGB565_LE, J.SCALE_HALF) == dict(bpp=2, out_w=167, out_h=109, canvas_w=168, canvas_h=112)
    assert p.geometry(J.RGB8888, J.LUMA_ONLY | J.SCALE_EIGHTH) == dict(bpp=1, out_w=42, out_h=28, canvas_w=42, canvas_h=28)


@pytest.mark.needs_reference
def test_cropped_draw_plan_matches_the_reference(ref_scalar):
    """jda_crop_round + jda_draw_plan_ex against the JPEGDRAW log of the reference run with setCropArea."""
    import ctypes as C
    import jpegdec_amd as J
    from jpegdec_amd.binding import ImageInfo, load_library
    checked = 0
    for name in ["c420_333x217", "c444_333x217", "gray_333x217", "c420_640x368_rstrow", "c420_1280x720"]:
        jpeg = jpeg_for(name)
        info = ImageInfo()
        assert load_library().jda_parse(jpeg, len(jpeg), C.byref(info)) == 0
        for crop in [(50, 50, 125, 170), (0, 0, 64, 64), (100, 20, 200, 100), (16, 16, 16, 16), (300, 200, 400, 300)]:
            rounded = J.crop_round(info, *crop)
            for pt in (0, 2, 3):
                for opt in (0, 128):
                    r = ref_scalar.decode_cb(jpeg, pt, opt, crop=crop, want_log=True)
                    plan = J.draw_plan_ex(info, pt, opt, uses_dma=bool(opt & 128), crop=rounded)
                    if r["rc"] != 1:
                        continue            # crop below the last MCU row: reference decodes past the scan and fails
                    assert np.array_equal(r["log"], plan[:, :6]), (name, crop, pt, opt)
                    checked += 1
    assert checked > 50


@pytest.mark.needs_reference
def test_exif_thumbnail_metadata_matches_the_reference(ref_scalar):
    from jpegdec_amd.synth import synth_jpeg
    from tests.exif_util import with_exif_thumbnail
    main, th = synth_jpeg(160, 120, "4:2:0", seed=3), synth_jpeg(64, 48, "4:2:0", seed=4)
    for be in (False, True):
        for dims in (True, False):
            for orient in (1, 6, 8):
                j = with_exif_thumbnail(main, th, 64, 48, orientation=orient, big_endian=be, with_dims=dims)
                info = ImageInfo()
                assert load_library().jda_parse(j, len(j), C.byref(info)) == 0
                r = ref_scalar.info(j)
                assert (info.has_thumb, info.thumb_w, info.thumb_h, info.orientation) == (r["hasthumb"], r["thumbw"], r["thumbh"], r["orientation"])
                assert j[info.thumb_offset:info.thumb_offset + 2] == b"\xff\xd8"
    plain = ImageInfo()
    assert load_library().jda_parse(main, len(main), C.byref(plain)) == 0 and plain.has_thumb == 0


def _dht_segments(jpeg):
    """offsets of the DHT table headers: (offset of the Tc/Th byte) for every table of every DHT segment"""
    out, at = [], 2
    while at + 4 <= len(jpeg):
        marker, seglen = jpeg[at + 1], (jpeg[at + 2] << 8) | jpeg[at + 3]
        if marker == 0xDA:
            break
        if marker == 0xC4:
            q, end = at + 4, at + 2 + seglen
            while q + 17 <= end:
                out.append(q)
                q += 17 + sum(jpeg[q + 1: q + 17])
        at += 2 + seglen
    return out


def test_oversubscribed_dht_is_rejected():
    """a DHT whose counts no prefix code can have (more codes of a length than there are bit patterns) made
    build_ac_lut / build_dc_lut write far outside the LUT.  Such a table is refused (JDA_UNSUPPORTED_FEATURE, the code
    JPEGMakeHuffTables failures map to, jpeg.inl:1771-1775) -- for every table of the file, DC and AC."""
    base = jpeg_for("c420_333x217")
    tabs = _dht_segments(base)
    assert len(tabs) == 4
    for q in tabs:
        for length, count in ((1, 255), (1, 3), (2, 5), (3, 9)):
            b = bytearray(base)
            total = sum(b[q + 1: q + 17])
            if count > total:
                continue
            # keep the total (the symbol bytes stay where they are): move `count` codes to `length`
            counts = [0] * 16
            counts[length - 1] = count
            counts[15] = total - count
            b[q + 1: q + 17] = bytes(counts)
            with pytest.raises(J.JdaError) as e:
                J.PreparedImage(bytes(b))
            assert e.value.code in (2, 3), (q, length, count, e.value.code)


def test_frontend_fuzz_under_asan_ubsan():
    """Header mutations (DHT / DQT / SOF / SOS / APPn bytes, truncations) and scan corruptions through jda_parse, jda_prepare_ex,
    the draw plan and the crop rounding, compiled with -fsanitize=address,undefined: any out-of-bounds access or UB aborts."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "frontfuzz"], cwd=root, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    exe = os.path.join(root, "tests", "fuzz", "frontend_fuzz")
    for seed, rel in enumerate(("golden/ref/tulips.jpg", "golden/ref/thumb_test.jpg", "golden/c444_8x8_q30.jpg", "golden/c420_16x16.jpg",
                                "golden/gray_64x64_rst3.jpg", "golden/c440_300x64_rst5.jpg", "golden/p420_200x120.jpg", "golden/ref/corrupt5.jpg",
                                "golden/c420_640x368_rstrow.jpg", "golden/c444_384x192_q100_rst7.jpg")):      # (the last two: the interval-parallel pre-scan under the sanitizers)
        p = subprocess.run([exe, os.path.join(root, "tests", rel), "2500", str(seed + 1)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert p.returncode == 0, (rel, p.stdout[-3000:])


def test_chunk_parallel_prescan_at_chunk_sizes_no_thread_count_produces():
    """host_prescan_chunks cuts the scan by the number of threads: a test machine sees one or two chunk sizes.  tests/fuzz/chunk_equiv.cpp
    (jda_frontend.cpp with the chunk size exposed, ASan + UBSan) runs chunks of 512 bytes .. 16 KB over streams without restart markers and
    corrupted copies of them against the serial pre-scan: same verdict, same index (jda_index_equivalent), DC values, truncation count,
    continuation entries.  (Small chunks are where a walker from the guess meets codes that do not exist -- it steps on a bit and keeps
    looking -- and where the splice has the least room.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "chunkequiv"], cwd=root, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    files = ["c420_1280x720.jpg", "c420_256x256_q98.jpg", "c420_333x217.jpg", "c422_333x217.jpg", "c440_200x120.jpg", "c444_256x256_q100_opt.jpg", "c444_333x217.jpg",
             "gray_333x217.jpg", "w16_c420_333x217_x400.jpg", "ref/zebra.jpg", "ref/st_peters.jpg", "ref/sciopero.jpg"]
    p = subprocess.run([os.path.join(root, "tests", "fuzz", "chunk_equiv"), "40", "7"] + files, cwd=os.path.join(root, "tests", "golden"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    words = p.stdout.strip().split()
    taken, threads = int(words[-12]), int(words[-2])
    assert taken > 1500 or threads == 1, p.stdout  # (the chunk-parallel pre-scan really made most of them -- where the machine has CPUs for helper threads)
    # an allocation that fails inside a job of the helper pool (std::bad_alloc thrown at the n-th allocation point, on whatever thread gets
    # there): the job ends like one that gave up -- same index as the serial pre-scan's, the pool usable afterwards.  Streams without
    # restart markers above (the chunk walkers), with them here (the interval walkers)
    assert "allocation failures injected and survived" in p.stdout and (threads == 1 or int(p.stdout.split("chunk_equiv: ")[-2].split()[0]) > 20), p.stdout
    q = subprocess.run([os.path.join(root, "tests", "fuzz", "chunk_equiv"), "0", "9", "c420_640x368_rstrow.jpg", "c444_384x192_q100_rst7.jpg", "ref/tulips.jpg"],
                       cwd=os.path.join(root, "tests", "golden"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert q.returncode == 0 and "allocation failures injected and survived" in q.stdout, q.stdout[-3000:]


def test_interval_parallel_prescan_under_thread_sanitizer():
    """The helper threads of the interval-parallel host pre-scan (jda_frontend.cpp: RstPool, rst_worker) under ThreadSanitizer: restart
    streams and corrupted copies of them (an interval that runs past its bytes must not touch what another thread writes)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "fronttsan"], cwd=root, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    exe = os.path.join(root, "tests", "fuzz", "frontend_tsan")
    for seed, rel in enumerate(("golden/ref/tulips.jpg", "golden/c444_384x192_q100_rst7.jpg", "golden/c420_640x368_rstrow.jpg", "golden/c420_1280x720.jpg", "golden/ref/zebra.jpg")):
        p = subprocess.run([exe, os.path.join(root, "tests", rel), "300", str(seed + 3)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert p.returncode == 0 and "ThreadSanitizer" not in p.stdout, (rel, p.stdout[-3000:])


def test_cropped_draw_plan_depends_on_the_x_offset(ref_scalar):
    """jpeg.inl:5328 compares jd.x (iXOffset included) with iCropX + iCropCX: the strip widths of a cropped decode change with the
    x passed to decode().  jda_draw_plan_at == the real reference's JPEGDRAW sequence for a grid of crops x offsets x pixel types."""
    lib = load_library()
    checked = 0
    for name in ("c420_640x368_rstrow", "c444_333x217", "gray_333x217", "c422_333x217"):
        jpeg = jpeg_for(name)
        p = J.PreparedImage(jpeg)
        for crop in ((16, 32, 100, 60), (50, 50, 125, 170), (0, 0, 64, 64), (100, 20, 200, 100), (8, 8, 300, 40)):
            for xoff in ((0, 3, 16) if crop[2] == 300 else (0, 3, 16, 61)):      # (the reference crashes on (8, 8, 300, 40) at x = 61)
                for pt in (0, 2, 3):
                    if name.startswith("gray") and pt == 2:
                        continue
                    b = ref_scalar.decode_cb(jpeg, pt, 0, crop=crop, want_log=True, xoff=xoff, yoff=5)
                    if b["rc"] != 1:
                        continue
                    c = (C.c_int32 * 4)(*J.crop_round(p.info, *crop))
                    rects = np.zeros((4096, 8), np.int32)
                    n = lib.jda_draw_plan_at(C.byref(p.info), pt, 0, 0, 0, c, xoff, rects.ctypes.data_as(C.c_void_p), 4096)
                    got = rects[:n, :6].copy()
                    got[:, 0] += xoff
                    got[:, 1] += 5
                    assert np.array_equal(got, b["log"]), (name, crop, xoff, pt)
                    checked += 1
        p.close()
    assert checked > 100


def test_word_precision_dqt_cases(hostsim, oracle):
    """jpeg.inl:1742-1750: the reference takes 16-bit quantisers as they come.  The w16_* cases (uniform noise, quantisers x 400 / x 3000
    at word precision) leave the range the kernels' 24-bit multiplier covers -- jda_image_fast_mul == 0 for each of them, i.e. they
    are the inputs of the 32-bit-multiply column stage --, every other synthetic case stays inside it; the oracle decodes them."""
    from tests.cases import SYNTH_CASES, WORD_DQT_CASES, jpeg_for
    assert len(WORD_DQT_CASES) == 5
    for name in sorted(SYNTH_CASES):
        jpeg = jpeg_for(name)
        assert hostsim.hostsim_fast_mul(jpeg, len(jpeg)) == (0 if name in WORD_DQT_CASES else 1), name
    for name in WORD_DQT_CASES:
        jpeg = jpeg_for(name)
        dqt = jpeg.index(b"\xff\xdb")
        assert jpeg[dqt + 4] >> 4 == 1                      # Pq = 1
        rc, want, err = oracle.decode_canvas(jpeg, 2 if "gray" not in name else 0, 0)
        assert rc == 1 and len(set(want.ravel().tolist())) > 8   # (not one flat colour: the IDCT's every path sees values)


def test_interval_parallel_prescan_equals_the_serial_one():
    """SURVEY 8f N1 / N2 on the host: a stream with restart intervals is pre-scanned interval by interval on helper threads
    (host_prescan_intervals), one without in chunks walked from a guess (host_prescan_chunks); the reference reader's phase settled
    afterwards through the chain of its refill points.
    Against the serial pre-scan (JDA_PREPARE_SERIAL_PRESCAN) on every restart stream of the suite, the reference's fixtures with DRI
    (tulips 7 truncated reads, croptest 39, demo 75, perf.jpg 1,793) and dense high-quality streams: the same index in the sense of
    jda_index_equivalent (positions and flags of every block, flagged entries identical), DC values, truncation count, continuation
    entries; and on 700 corrupted copies the same verdict (where the parallel walk gives up, the serial pre-scan decides)."""
    import jpegdec_amd as J
    from jpegdec_amd.synth import synth_jpeg
    from tests.cases import SYNTH_CASES, jpeg_for
    from tests.ref_fixtures import GOOD, ref_jpeg
    cases = [(n, jpeg_for(n)) for n in sorted(SYNTH_CASES)] + [("ref:" + n, ref_jpeg(n)) for n in GOOD]
    cases += [("1080p_rstrow", synth_jpeg(1920, 1080, "4:2:0", seed=3, restart_rows=1)), ("c444_q100_rst9", synth_jpeg(640, 480, "4:4:4", seed=4, quality=100, restart_blocks=9)),
              ("c420_q98_rst5", synth_jpeg(800, 608, "4:2:0", seed=5, quality=98, restart_blocks=5)), ("gray_q95_rst2", synth_jpeg(512, 512, "gray", seed=6, quality=95, restart_blocks=2))]
    # .. and WITHOUT restart markers: chunks of the scan walked from a guess, the true path spliced in front of where each walker fell
    # into step (host_prescan_chunks)
    cases += [("1080p", synth_jpeg(1920, 1080, "4:2:0", seed=3)), ("c444_q100", synth_jpeg(640, 480, "4:4:4", seed=4, quality=100)), ("c420_q98", synth_jpeg(800, 608, "4:2:0", seed=5, quality=98)),
              ("gray_q95", synth_jpeg(512, 512, "gray", seed=6, quality=95)), ("c422", synth_jpeg(1024, 768, "4:2:2", seed=7)), ("noise_q90", synth_jpeg(640, 480, "4:2:0", seed=8, quality=90, noise=True))]

    def same(jpeg, flags, what):
        try:
            a = J.PreparedImage(jpeg, flags=flags | J.PREPARE_SERIAL_PRESCAN)
        except J.JdaError as ea:
            with pytest.raises(J.JdaError) as eb:
                J.PreparedImage(jpeg, flags=flags | J.PREPARE_PARALLEL_PRESCAN)
            assert eb.value.code == ea.code, what
            return 0
        b = J.PreparedImage(jpeg, flags=flags | J.PREPARE_PARALLEL_PRESCAN)     # (the helper threads whatever the file's size)
        (ia, na), (ib, nb) = a.block_index(), b.block_index()
        assert na == nb, what
        if na == a.n_mcus:
            assert J.index_equivalent(ia, ib), what
            assert np.array_equal(a.block_dc(), b.block_dc()) and a.truncation_events() == b.truncation_events(), what
            (fa, ea_), (fb, eb_) = a.block_cont(), b.block_cont()
            assert np.array_equal(fa, fb) and np.array_equal(ea_, eb_), what
        else:                                   # the serial pre-scan made both: identical
            assert np.array_equal(ia, ib) and np.array_equal(a.block_dc(), b.block_dc()), what
        n = a.truncation_events()
        a.close(); b.close()
        return n

    trunc = 0
    for name, jpeg in cases:
        for flags in (0, J.PREPARE_CONT_ALWAYS):
            trunc += same(jpeg, flags, name)
    assert trunc > 3000                       # (the phase chain really decided: perf.jpg alone has 1,793 truncated reads)
    rng = np.random.default_rng(77)
    for name in ("c420_640x368_rstrow", "c444_384x192_q100_rst7", "c420_512x256_q98_rstrow", "c420_1280x720", "c444_256x256_q100_opt", "w16_c420_333x217_x400", "ref:zebra"):
        jpeg = ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name)
        sos = jpeg.index(b"\xff\xda") + 14
        for it in range(100):
            bad = bytearray(jpeg)
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(sos, len(bad) - 2))
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    bad[at] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    bad[at] = int(rng.integers(0, 256))
                else:
                    del bad[at:at + int(rng.integers(1, 40))]
            same(bytes(bad), 0, (name, it))


def test_prepare_in_a_forked_child_after_the_helper_threads_exist():
    """The helper threads of the parallel host pre-scans do not exist in a fork()ed child: the child pre-scans on its own thread (same index)
    and exits normally (the pool is never destroyed: a destructor joining threads the child does not have would hang it)."""
    import subprocess
    import sys
    code = """
import os, sys
sys.path.insert(0, %r)
import jpegdec_amd as J
from tests.ref_fixtures import ref_jpeg
j = ref_jpeg('tulips')
a = J.PreparedImage(j, flags=J.PREPARE_PARALLEL_PRESCAN)
ia = a.block_index()[0]
pid = os.fork()
if pid == 0:
    b = J.PreparedImage(j, flags=J.PREPARE_PARALLEL_PRESCAN)
    ok = J.index_equivalent(ia, b.block_index()[0])
    sys.exit(0 if ok else 3)             # (a normal exit: static destructors run)
_, st = os.waitpid(pid, 0)
print('child', st)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "child 0" in r.stdout, r.stdout[-2000:]


def test_host_prescan_helpers_can_be_switched_off():
    """jda_set_host_prescan_helpers(0): every pre-scan on its caller's thread, whatever the flags ask for -- and the same index (the
    serial pre-scan's own).  Restoring the default brings the helpers back (where the machine has the CPUs for them)."""
    import jpegdec_amd as J
    from jpegdec_amd.synth import synth_jpeg
    lib = J.load_library()
    jpeg = synth_jpeg(1920, 1080, "4:2:0", seed=3)
    ref = J.PreparedImage(jpeg, flags=J.PREPARE_SERIAL_PRESCAN)
    was = lib.jda_set_host_prescan_helpers(0)
    try:
        assert was == -1
        a = J.PreparedImage(jpeg, flags=J.PREPARE_PARALLEL_PRESCAN)
        assert np.array_equal(a.block_index()[0], ref.block_index()[0]) and np.array_equal(a.block_dc(), ref.block_dc())      # identical, not merely equivalent: the serial pre-scan made it
        a.close()
        assert lib.jda_set_host_prescan_helpers(1) == 0
        b = J.PreparedImage(jpeg, flags=J.PREPARE_PARALLEL_PRESCAN)
        assert J.index_equivalent(b.block_index()[0], ref.block_index()[0]) and np.array_equal(b.block_dc(), ref.block_dc())
        b.close()
    finally:
        lib.jda_set_host_prescan_helpers(-1)
    c = J.PreparedImage(jpeg, flags=J.PREPARE_PARALLEL_PRESCAN)
    assert J.index_equivalent(c.block_index()[0], ref.block_index()[0]) and np.array_equal(c.block_dc(), ref.block_dc())
    c.close(); ref.close()
