"""The reference's class API (openFLASH / setPixelType / decode / JPEG_DRAW_CALLBACK /
setFramebuffer) on the GPU path.  The SAME driver (oracle/ref_shim.cpp) is built once against the
real reference and once against include/JPEGDEC.h + libjpegdec_amd.so; outputs, draw-callback
sequences and error codes must agree.  Mirrors the reference's own tests 1, 9 and the perf loop
(MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp:74-105, 218-234)."""
import os

import numpy as np
import pytest

from oracle.loader import GRAY8, RGB565_LE, RGB8888, SCALE_HALF, SCALE_QUARTER, USES_DMA, RefDecoder
from tests.cases import jpeg_for

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product_class(gpu_ctx):
    import subprocess
    subprocess.run(["make", "classshim"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "gray_333x217", "c420_1100x48", "c420_640x368_rstrow"])
def test_draw_callbacks_deliver_the_oracle_frame(name, product_class, oracle):
    jpeg = jpeg_for(name)
    inf = product_class.info(jpeg)
    oi = oracle.info(jpeg)
    assert inf["ok"] == 1 and (inf["width"], inf["height"], inf["subsample"]) == (oi["width"], oi["height"], oi["subsample"])
    for pt, opt in ((RGB565_LE, 0), (RGB8888, 0), (GRAY8, 0), (RGB565_LE, SCALE_HALF), (GRAY8, SCALE_QUARTER)):
        if name.startswith("gray") and pt == RGB8888:
            continue
        r = product_class.decode_cb(jpeg, pt, opt, want_log=True)
        assert r["rc"] == 1 and r["last_error"] == 0
        rc, want, _ = oracle.decode_canvas(jpeg, pt, opt)
        sh = r["scale_shift"]
        h = (inf["height"] + (1 << sh) - 1) >> sh
        assert np.array_equal(r["canvas"][:h, : want.shape[1]], want[:h]), (name, pt, opt)
        assert np.array_equal(r["log"], oracle.draw_plan(jpeg, pt, opt)), (name, pt, opt)   # same JPEGDRAW sequence


@pytest.mark.parametrize("name", ["p420_200x120", "p444_333x217", "pgray_100x100"])
def test_progressive_file_is_a_dc_thumbnail(name, product_class, oracle):
    """SURVEY 8f N4 through the class: getJPEGType() says progressive, decode() delivers the 1/8 thumbnail of the first
    (DC) scan with the reference's JPEGDRAW sequence (jpeg.inl:4964-4966)."""
    from tests.cases import progressive_modes
    jpeg = jpeg_for(name)
    inf = product_class.info(jpeg)
    assert inf["ok"] == 1 and inf["jpegtype"] == 1
    for pt, opt in progressive_modes(name):
        r = product_class.decode_cb(jpeg, pt, opt, want_log=True)
        assert r["rc"] == 1 and r["last_error"] == 0, (name, pt, opt, r["last_error"])
        rc, want, _ = oracle.decode_canvas(jpeg, pt, opt)
        sh = r["scale_shift"]
        h = (inf["height"] + (1 << sh) - 1) >> sh
        assert np.array_equal(r["canvas"][:h, : want.shape[1]], want[:h]), (name, pt, opt)
        assert np.array_equal(r["log"], oracle.draw_plan(jpeg, pt, opt)), (name, pt, opt)


def test_same_behaviour_as_the_real_reference(product_class, ref_scalar):
    jpeg = jpeg_for("c420_333x217")
    for pt, opt, mm in ((RGB565_LE, 0, 0), (RGB8888, 0, 3), (GRAY8, SCALE_HALF, 0), (RGB565_LE, USES_DMA, 0)):
        a = product_class.decode_cb(jpeg, pt, opt, max_mcus=mm, want_log=True, xoff=5, yoff=7)
        b = ref_scalar.decode_cb(jpeg, pt, opt, max_mcus=mm, want_log=True, xoff=5, yoff=7)
        assert a["rc"] == b["rc"] == 1
        assert np.array_equal(a["log"], b["log"])
        assert a["dma_reuse"] == b["dma_reuse"]                       # reference test 9: DMA ping-pong
        h = a["canvas"].shape[0] - 16
        assert np.array_equal(a["canvas"][:h], b["canvas"][:h])
    # early exit: callback returns 0 after 3 strips (jpeg.inl:5325) -- decode still returns 1
    a = product_class.decode_cb(jpeg, RGB565_LE, 0, stop_after=3)
    b = ref_scalar.decode_cb(jpeg, RGB565_LE, 0, stop_after=3)
    assert (a["rc"], a["n_calls"]) == (b["rc"], b["n_calls"]) == (1, 3)


def test_framebuffer_mode(product_class, oracle):
    jpeg = jpeg_for("c420_640x368_rstrow")            # width is an MCU multiple: framebuffer == canvas
    rc, fb = product_class.decode_fb(jpeg, RGB8888, 0)
    assert rc == 1
    orc, want, _ = oracle.decode_canvas(jpeg, RGB8888, 0)
    assert np.array_equal(fb[: want.size].reshape(want.shape), want)


def test_error_codes(product_class):
    good = jpeg_for("c420_16x16")
    assert product_class.info(good[:100])["lasterror"] == 4          # JPEG_INVALID_FILE
    bad = bytearray(good); bad[good.index(b"\xff\xc0") + 1] = 0xC1
    assert product_class.info(bytes(bad))["lasterror"] == 3          # JPEG_UNSUPPORTED_FEATURE


CROPS = [(50, 50, 125, 170), (0, 0, 64, 64), (100, 20, 200, 100), (16, 16, 16, 16), (0, 180, 64, 30)]


@pytest.mark.parametrize("name", ["c420_640x368_rstrow", "c444_600x16", "c420_1280x720", "gray_333x217"])
def test_crop_area_matches_the_reference(name, product_class, ref_scalar):
    """setCropArea (jpeg.inl:682-727) + the skip/draw bookkeeping of :5111, :5134-5137, :5300-5336 --
    the reference's own test 2 (MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp) checks the drawn union only;
    here every strip and every pixel is compared with the reference itself."""
    jpeg = jpeg_for(name)
    checked = 0
    for crop in CROPS:
        for pt, opt in ((RGB565_LE, 0), (RGB8888, 0), (GRAY8, 0)):
            if name.startswith("gray") and pt == RGB8888:
                continue                              # reference writes 16-bit pixels at 32 bpp here
            b = ref_scalar.decode_cb(jpeg, pt, opt, crop=crop, want_log=True)
            a = product_class.decode_cb(jpeg, pt, opt, crop=crop, want_log=True)
            if b["rc"] != 1:
                # crop reaching below the last MCU row: the reference runs off the scan and fails
                assert (a["rc"], a["last_error"]) == (b["rc"], b["last_error"]) == (0, 2), (name, crop)
                continue
            assert a["rc"] == 1 and a["last_error"] == 0
            assert np.array_equal(a["log"], b["log"]), (name, crop, pt)
            assert np.array_equal(a["canvas"], b["canvas"]), (name, crop, pt)
            checked += 1
    assert checked >= 6


@pytest.mark.parametrize("name,crop", [("c420_333x217", None), ("c444_333x217", None), ("gray_333x217", None),
                                       ("c420_640x368_rstrow", (50, 50, 125, 170)), ("c420_1280x720", (100, 20, 200, 100)),
                                       ("c444_600x16", (16, 0, 64, 16)), ("gray_1600x16", (16, 0, 64, 16))])
def test_framebuffer_ragged_and_cropped(name, crop, product_class, ref_scalar):
    """Framebuffer mode with a width that is not an MCU multiple, or with a crop: MCUs past the pitch wrap
    into the next buffer row or are clipped exactly as the reference's JPEGPutMCU* variants leave them
    (jpeg.inl:5114-5124; see the comment in jpegdec_amd/csrc/JPEGDEC.cpp)."""
    import ctypes as C
    import jpegdec_amd as J
    from jpegdec_amd.binding import ImageInfo, load_library
    from oracle.loader import SCALE_EIGHTH
    jpeg = jpeg_for(name)
    inf = ref_scalar.info(jpeg)
    info = ImageInfo(); load_library().jda_parse(jpeg, len(jpeg), C.byref(info))
    checked = 0
    for opt in (0, SCALE_HALF, SCALE_QUARTER, SCALE_EIGHTH):
        for pt in (RGB565_LE, RGB8888, GRAY8):
            if inf["subsample"] == 0 and pt == RGB8888:
                continue                              # grayscale + RGB8888: reference writes 16-bit pixels
            if name == "c444_333x217" and opt == 0 and pt != GRAY8:
                continue                              # reference bug in the clipped last MCU (jpeg.inl:3521-3557), documented divergence
            rb, fb_ref = ref_scalar.decode_fb(jpeg, pt, opt, crop=crop)
            ra, fb_got = product_class.decode_fb(jpeg, pt, opt, crop=crop)
            assert ra == rb == 1
            bpp = {RGB565_LE: 2, RGB8888: 4, GRAY8: 1}[pt]
            sh = {0: 0, SCALE_HALF: 1, SCALE_QUARTER: 2, SCALE_EIGHTH: 3}[opt]
            mh = info.mcu_h >> sh
            cx_, cy_, cw_, chh = J.crop_round(info, *crop) if crop else (0, 0, inf["width"], inf["height"])
            rows_mcu = min((cy_ + chh + info.mcu_h - 1) // info.mcu_h, info.mcus_y)
            kept = [y for y in range(rows_mcu) if y * mh >= cy_]
            if not kept:
                assert not fb_got.any() and not fb_ref.any()
                continue
            n = (kept[-1] * mh - cy_ + mh) * cw_ * bpp
            assert np.array_equal(fb_got[:n], fb_ref[:n]), (name, crop, pt, opt)
            checked += 1
    assert checked >= 6


def test_exif_thumbnail(product_class, ref_scalar):
    """hasThumb / getThumbWidth / getThumbHeight and decode(JPEG_EXIF_THUMBNAIL) (jpeg.inl:1654-1678, 4967-4976):
    the embedded JPEG is decoded instead of the main image; without the IFD1 size tags the reference refuses."""
    from jpegdec_amd.synth import synth_jpeg
    from tests.exif_util import with_exif_thumbnail
    main, th = synth_jpeg(320, 240, "4:2:0", seed=3), synth_jpeg(64, 48, "4:2:2", seed=4)
    for be in (False, True):
        j = with_exif_thumbnail(main, th, 64, 48, big_endian=be)
        a, b = product_class.info(j), ref_scalar.info(j)
        assert (a["hasthumb"], a["thumbw"], a["thumbh"], a["orientation"]) == (b["hasthumb"], b["thumbw"], b["thumbh"], b["orientation"]) == (1, 64, 48, 6)
        for pt, opt in ((RGB8888, 32), (RGB565_LE, 32 | SCALE_HALF), (GRAY8, 32)):
            ra = product_class.decode_cb(j, pt, opt, want_log=True)
            rb = ref_scalar.decode_cb(j, pt, opt, want_log=True)
            assert ra["rc"] == rb["rc"] == 1
            assert np.array_equal(ra["log"], rb["log"])
            assert np.array_equal(ra["canvas"][:48], rb["canvas"][:48]), (be, pt, opt)
    j = with_exif_thumbnail(main, th, 64, 48, with_dims=False)
    ra, rb = product_class.decode_cb(j, RGB8888, 32), ref_scalar.decode_cb(j, RGB8888, 32)
    assert (ra["rc"], ra["last_error"]) == (rb["rc"], rb["last_error"]) == (0, 1)
    # the main image still decodes as before
    ra, rb = product_class.decode_cb(j, RGB8888, 0), ref_scalar.decode_cb(j, RGB8888, 0)
    assert ra["rc"] == rb["rc"] == 1 and np.array_equal(ra["canvas"][:240], rb["canvas"][:240])


def test_exif_thumbnail_of_the_other_kind_than_its_main_image(product_class, ref_scalar):
    """jpeg.inl:4964-4976: the reference decides on JPEG_SCALE_EIGHTH (progressive = DC-only thumbnail) from the MAIN image's mode
    before it parses the EXIF thumbnail, and then decodes the thumbnail with that option: a progressive main file with a baseline
    thumbnail decodes the thumbnail at 1/8.  Same draw sequence and pixels.  Refused with JPEG_UNSUPPORTED_FEATURE:
    the other way round -- a progressive thumbnail in a baseline file at full size: the reference runs its baseline decoder over
    progressive scan data (garbage; a segmentation fault with JPEG_SCALE_HALF -> GRAY8) -- and a baseline thumbnail of a progressive
    file at 1/2: two scale bits on a baseline image, the reference's IDCT then runs over coefficients earlier blocks left behind."""
    from jpegdec_amd.synth import synth_jpeg
    from tests.cases import jpeg_for
    from tests.exif_util import with_exif_thumbnail
    base_main, base_thumb = synth_jpeg(320, 240, "4:2:0", seed=3), synth_jpeg(64, 48, "4:2:0", seed=4)
    prog_main, prog_thumb = jpeg_for("p420_200x120"), synth_jpeg(64, 48, "4:2:0", seed=5, progressive=True)
    checked, wrong = 0, []
    for nm, main, th in (("prog+base", prog_main, base_thumb), ("prog+prog", prog_main, prog_thumb)):
        j = with_exif_thumbnail(main, th, 64, 48)
        a, b = product_class.info(j), ref_scalar.info(j)
        assert (a["hasthumb"], a["thumbw"], a["thumbh"]) == (b["hasthumb"], b["thumbw"], b["thumbh"])
        for pt, opt in ((RGB8888, 32), (RGB565_LE, 32), (GRAY8, 32 | SCALE_HALF)):
            if nm == "prog+prog" and (opt & SCALE_HALF):
                continue                                          # (the reference dies of a segmentation fault on this one)
            if nm == "prog+base" and (opt & SCALE_HALF):
                ra = product_class.decode_cb(j, pt, opt)
                assert (ra["rc"], ra["last_error"]) == (0, 3)
                continue
            ra = product_class.decode_cb(j, pt, opt, want_log=True)
            rb = ref_scalar.decode_cb(j, pt, opt, want_log=True)
            if (ra["rc"], ra["last_error"]) != (rb["rc"], rb["last_error"]):
                wrong.append((nm, pt, opt, "rc", ra["rc"], ra["last_error"], rb["rc"], rb["last_error"]))
            elif rb["rc"] == 1:
                if not np.array_equal(ra["log"], rb["log"]):
                    wrong.append((nm, pt, opt, "log", ra["log"][:2].tolist(), rb["log"][:2].tolist()))
                elif not np.array_equal(ra["canvas"][:48], rb["canvas"][:48]):
                    wrong.append((nm, pt, opt, "pixels"))
                checked += 1
    assert not wrong, wrong
    assert checked >= 3
    j = with_exif_thumbnail(base_main, prog_thumb, 64, 48)
    ra = product_class.decode_cb(j, RGB8888, 32)
    assert (ra["rc"], ra["last_error"]) == (0, 3)                 # JPEG_UNSUPPORTED_FEATURE


def test_objects_on_four_threads_decode_concurrently(product_class):
    """One JPEGDEC object per thread is the reference's threading model; every thread gets its own device context
    (stream, staging buffer, block pool), so four threads must beat one -- and deliver the same pixels."""
    import threading
    import time
    from oracle.loader import digest
    from tests.ref_fixtures import ref_golden, ref_jpeg

    jpeg = ref_jpeg("tulips")
    want = ref_golden()["tulips"]["frames"]["0:0"]["sha"]
    n_each = 30

    def work(out):
        ok = 0
        for _ in range(n_each):
            r = product_class.decode_cb(jpeg, RGB565_LE, 0)
            ok += int(r["rc"] == 1 and digest(r["canvas"][:480, : 640 * 2]) == want)
        out.append(ok)

    one = []
    work(one)                                             # (first use: context creation, code objects)
    res = []
    th = [threading.Thread(target=work, args=(res,)) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert one == [n_each] and res == [n_each] * 4        # every decode right, on every thread
    # the rate: C loops of open + decode (no-op draw callback) + close, one object per thread (the shim's bench entry: hashing canvases
    # in Python holds the interpreter's lock for longer than a decode takes, and threads that queue for it measure the interpreter)
    reps = 300
    product_class.bench([jpeg], RGB565_LE, 0, reps=50, threads=4)
    r1 = product_class.bench([jpeg], RGB565_LE, 0, reps=reps, threads=1)
    r4 = product_class.bench([jpeg], RGB565_LE, 0, reps=reps, threads=4)
    assert r1["failures"] == 0 and r4["failures"] == 0
    speedup = (4 * reps / r4["seconds"]) / (reps / r1["seconds"])
    print("class decode: 1 thread %.3f ms / image, 4 threads %.2fx the throughput" % (r1["seconds"] / reps * 1e3, speedup))
    # (one thread's decode has the pre-scan's helper threads to itself; of four at once one has them and three pre-scan on their own
    # thread: 2x is about what there is to get, and a loaded box takes its share)
    assert speedup > 1.15, speedup


def test_framebuffer_direct_copy_and_device_selection(product_class, ref_scalar):
    """Framebuffer mode with an MCU-multiple width takes the direct D2H path (no intermediate canvas): same bytes as the
    real reference, for every pixel type."""
    from tests.ref_fixtures import ref_jpeg
    for name, w, h in (("tulips", 640, 480), ("zebra", 320, 240)):
        for pt, bpp in ((RGB565_LE, 2), (RGB8888, 4), (GRAY8, 1)):
            rc_a, fa = product_class.decode_fb(ref_jpeg(name), pt, 0)
            rc_b, fb = ref_scalar.decode_fb(ref_jpeg(name), pt, 0)
            assert rc_a == rc_b == 1 and np.array_equal(fa[: w * h * bpp], fb[: w * h * bpp]), (name, pt)


def test_framebuffer_keeps_what_a_bad_stream_does_not_reach(product_class, oracle):
    """Framebuffer mode on a stream with a bad MCU: the reference returns at that MCU (jpeg.inl:5354-5356) and whatever the caller's
    buffer held behind it stays -- the direct copy-back path must not lay the canvas's zero fill over it (round 2 did)."""
    import jpegdec_amd as J
    from jpegdec_amd.synth import synth_jpeg
    base = bytearray(synth_jpeg(320, 240, "4:2:0", seed=21, quality=85))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(8)
    done = 0
    for _ in range(400):
        b = bytearray(base)
        q = int(rng.integers(sos + 14 + (len(b) - sos) // 4, len(b) - 8 - (len(b) - sos) // 4))
        b[q:q + 6] = b"\xff\x00\xff\x00\xff\x00"       # 24 one bits in the filtered stream: no Annex K code is all ones
        jb = bytes(b)
        try:
            p = J.PreparedImage(jb)
        except J.JdaError:
            continue
        nok, total = p.block_index()[1], p.n_mcus
        p.close()
        if not (total // 8 < nok < total - total // 8):
            continue
        rc, fb = product_class.decode_fb(jb, RGB8888, 0, fill=0x5A)
        assert rc == 0 and product_class.last_error == 2                     # JPEG_DECODE_ERROR
        orc, want, err = oracle.decode_canvas(jb, RGB8888, 0)               # (MCUs before the bad one; zeros behind)
        got = fb[: 320 * 240 * 4].reshape(240, 320 * 4)
        mx = 320 // 16
        for m in range(total):
            y, x = (m // mx) * 16, (m % mx) * 16
            blk = got[y:y + 16, x * 4:(x + 16) * 4]
            if m < nok:
                assert np.array_equal(blk, want[y:y + 16, x * 4:(x + 16) * 4]), (m, nok)
            else:
                assert np.all(blk == 0x5A), (m, nok)
        done += 1
        if done == 3:
            break
    assert done == 3


def test_large_images_leave_the_device_strip_major(product_class, ref_scalar):
    """tests/strip_major_cases.py on the GPU: the kernels write a surface of 2 MB and more as the reference's JPEGDRAW strips (tiles cut
    at strip edges, every strip's pixels contiguous), the class hands out pointers into the page-locked copy of it -- every strip and
    every pixel the unmodified reference's."""
    from tests.strip_major_cases import check_strip_major_decodes
    check_strip_major_decodes(product_class, ref_scalar)
