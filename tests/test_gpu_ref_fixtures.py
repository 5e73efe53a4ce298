"""The reference's OWN fixtures through the HIP kernels (VERDICT r1 task 1): test_images/{tulips,zebra,st_peters,sciopero,
thumb_test}.h, examples/crop_area/croptest.h, MacOS/JPEGDEC_Test/JPEGDEC_Test/corrupt1-5.h, demo.jpg, perf.jpg,
squirrel_dither.jpg -- camera / libjpeg quantisers, 1.4-4.1 bit/px (zebra, croptest, sciopero and st_peters sit past the
960-byte LDS window: the general bit reader), hundreds of window-truncation events (perf.jpg: 1,793).

Checkers: the hashes the REAL reference produced (tests/golden/ref/ref_golden.json), the oracle restatement (canvas
byte for byte, MCU padding included) and, where it travelled, oracle/_ref itself.  Everything goes through the C-ABI
(ctypes) or through the drop-in class (tests/libjpegdec_class_shim.so = oracle/ref_shim.cpp built against
include/JPEGDEC.h + libjpegdec_amd.so).  Nothing here reads /root/reference."""
import os
import subprocess

import numpy as np
import pytest

import jpegdec_amd as J
from oracle.loader import EXIF_THUMBNAIL, GRAY8, RGB565_BE, RGB565_LE, RGB8888, USES_DMA, RefDecoder, digest
from tests.ref_fixtures import BIG, CROP_INO, FAIL_IN_DECODE, GOOD, REJECTED_AT_OPEN, frame_of, modes_of, ref_golden, ref_jpeg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product_class(gpu_ctx):
    subprocess.run(["make", "classshim"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))


@pytest.mark.parametrize("name", GOOD + BIG)
def test_fixture_all_modes_bit_exact(name, gpu_ctx, oracle):
    """every pixel type x option through jda_decode_to_host: canvas == oracle canvas, frame hash == the real reference's"""
    jpeg = ref_jpeg(name)
    g = ref_golden()[name]
    for pt, opt in modes_of(name):
        fr = g["frames"]["%d:%d" % (pt, opt)]
        rc, got, geo = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
        assert rc == 0, (name, pt, opt, rc)
        assert digest(got[: fr["h"], : fr["w"] * fr["bpp"]]) == fr["sha"], (name, pt, opt)
        orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert orc == 1 and got.shape == want.shape
        assert np.array_equal(got, want), "%s pt=%d opt=%d: %d differing bytes" % (name, pt, opt, int(np.count_nonzero(got != want)))


@pytest.mark.parametrize("device_prescan", [False, True])
def test_fixtures_as_one_resident_batch(device_prescan, gpu_ctx):
    """All decodable fixtures resident in HBM, one launch plan (jda_upload_batch / jda_batch_create / jda_batch_decode),
    with the per-block index made by the serial host pre-scan and by the device pre-scan: the device index equals the host's
    entry for entry (reader phase, truncation flags, DC predictors) and the frames carry the real reference's hashes."""
    names = list(GOOD + BIG)
    modes = [(RGB8888, 0), (RGB565_LE, 0), (GRAY8, 0), (RGB565_BE, J.SCALE_HALF), (RGB8888, J.SCALE_QUARTER), (GRAY8, J.SCALE_EIGHTH)]
    host = [J.PreparedImage(ref_jpeg(n)) for n in names]
    prepared = J.prepare_batch([ref_jpeg(n) for n in names], device_prescan=device_prescan, threads=4)
    dev = J.upload_batch(gpu_ctx, prepared)
    if device_prescan:
        for n, d, h in zip(names, dev, host):
            # (corrupt5 is 48 MCUs followed by 40 KB of unrelated bytes: the segment walk hands it to the serial pre-scan)
            assert d.prescan_on_device or n == "corrupt5", n
            idx, dc = d.read_index()
            assert J.index_equivalent(idx, h.block_index()[0]), n
            assert np.array_equal(dc, h.block_dc()), n
    for pt, opt in modes:
        sel = [i for i, n in enumerate(names) if "%d:%d" % (pt, opt) in ref_golden()[n]["frames"]]
        outs, ptrs, geos = [], [], []
        for i in sel:
            geo = host[i].geometry(pt, opt)
            pitch = (geo["canvas_w"] * geo["bpp"] + 15) & ~15
            ptr = gpu_ctx.malloc(pitch * geo["canvas_h"])
            outs.append((ptr, pitch, geo["canvas_w"], geo["canvas_h"]))
            ptrs.append(ptr); geos.append((geo, pitch))
        batch = J.Batch(gpu_ctx, [dev[i] for i in sel], outs, [pt] * len(sel), [opt] * len(sel))
        batch.decode()
        gpu_ctx.sync()
        for i, ptr, (geo, pitch) in zip(sel, ptrs, geos):
            fr = ref_golden()[names[i]]["frames"]["%d:%d" % (pt, opt)]
            got = gpu_ctx.to_host(ptr, pitch * geo["canvas_h"]).reshape(geo["canvas_h"], pitch)
            assert digest(got[: fr["h"], : fr["w"] * fr["bpp"]]) == fr["sha"], (names[i], pt, opt, device_prescan)
        batch.close()
        for ptr in ptrs:
            gpu_ctx.free(ptr)
    for d in dev:
        d.close()
    for p in prepared + host:
        p.close()


@pytest.mark.parametrize("name", GOOD)
def test_class_draw_callbacks_equal_reference(name, product_class):
    """openFLASH / setPixelType / decode / JPEG_DRAW_CALLBACK on every fixture: frame and JPEGDRAW sequence == the real reference's"""
    jpeg = ref_jpeg(name)
    g = ref_golden()[name]
    for pt, opt in ((RGB565_LE, 0), (RGB8888, 0), (GRAY8, 0), (RGB565_BE, J.SCALE_HALF), (RGB8888, J.SCALE_QUARTER), (RGB565_LE, J.SCALE_EIGHTH), (RGB565_LE, J.LUMA_ONLY)):
        fr = g["frames"]["%d:%d" % (pt, opt)]
        r = product_class.decode_cb(jpeg, pt, opt, want_log=True)
        assert (r["rc"], r["last_error"], r["n_calls"]) == (1, 0, fr["draw_calls"]), (name, pt, opt)
        frame, w, h = frame_of(r)
        assert digest(frame) == fr["sha"] and digest(r["log"]) == fr["log_sha"], (name, pt, opt)


def test_reference_test_1_and_9_tulips(product_class):
    """main.cpp:74-105 (the drawn extent equals the image size) and :218-234 (JPEG_USES_DMA toggles the buffer every callback)"""
    jpeg = ref_jpeg("tulips")
    r = product_class.decode_cb(jpeg, RGB565_LE, 0, want_log=True, used_only=True)
    log = r["log"]
    x1, y1 = log[:, 0].min(), log[:, 1].min()
    x2, y2 = (log[:, 0] + log[:, 4] - 1).max(), (log[:, 1] + log[:, 3] - 1).max()
    assert (1 + x2 - x1, 1 + y2 - y1) == (640, 480)
    r = product_class.decode_cb(jpeg, RGB565_LE, USES_DMA)
    assert r["rc"] == 1 and r["dma_reuse"] == 0


def test_reference_test_2_crop_tulips(product_class):
    """main.cpp:107-137: setCropArea(50, 50, 125, 170) gets MCU-adjusted; the drawn extent equals the adjusted rectangle"""
    jpeg = ref_jpeg("tulips")
    p = J.PreparedImage(jpeg)
    cx, cy, cw, ch = J.crop_round(p.info, 50, 50, 125, 170)
    assert (cx, cy, cw, ch) == (48, 48, 128, 176)
    r = product_class.decode_cb(jpeg, RGB565_LE, 0, crop=(50, 50, 125, 170), want_log=True)
    assert r["rc"] == 1
    log = r["log"]
    w = 1 + (log[:, 0] + log[:, 4] - 1).max() - log[:, 0].min()
    h = 1 + (log[:, 1] + log[:, 3] - 1).max() - log[:, 1].min()
    assert (w, h) == (cw, ch)


def test_crop_area_ino_rectangle(product_class):
    """examples/crop_area/crop_area.ino:92 -- setCropArea(120, 65, 119, 110) on croptest: strips and pixels == the real reference's"""
    jpeg = ref_jpeg("croptest")
    gold = ref_golden()["croptest"]["crop_ino"]
    for pt in (RGB565_LE, RGB565_BE, RGB8888, GRAY8):
        r = product_class.decode_cb(jpeg, pt, 0, crop=CROP_INO, want_log=True)
        gd = gold[str(pt)]
        assert (r["rc"], r["last_error"], r["n_calls"]) == (gd["rc"], gd["err"], gd["draw_calls"])
        assert digest(r["canvas"]) == gd["sha"] and digest(r["log"]) == gd["log_sha"], pt


def test_reference_test_10_exif_thumbnail(product_class):
    """main.cpp:236-260: thumb_test has an EXIF thumbnail; decode(JPEG_EXIF_THUMBNAIL) leaves the object describing a 320x240 image.
    Pixels and JPEGDRAW sequence of the thumbnail == the real reference's, at every scale."""
    jpeg = ref_jpeg("thumb_test")
    inf = product_class.info(jpeg)
    assert inf["ok"] == 1 and inf["hasthumb"] == 1 and (inf["thumbw"], inf["thumbh"]) == (320, 240)
    gold = ref_golden()["thumb_test"]["exif_thumbnail"]
    for key, gd in sorted(gold.items()):
        pt, opt = (int(v) for v in key.split(":"))
        r = product_class.decode_cb(jpeg, pt, opt | EXIF_THUMBNAIL, canvas_shape=(256, 336), want_log=True)
        assert (r["rc"], r["last_error"], r["n_calls"]) == (gd["rc"], gd["err"], gd["draw_calls"]), key
        assert list(r["size_after"]) == gd["size_after"] == [320, 240]
        assert digest(r["canvas"]) == gd["sha"] and digest(r["log"]) == gd["log_sha"], key


@pytest.mark.parametrize("name", REJECTED_AT_OPEN)
def test_reference_tests_4_7_rejected_at_open(name, product_class, gpu_ctx):
    """main.cpp:166-175, :199-207: corrupt1 / FPE1 -- openFLASH fails with the reference's error code, nothing crashes"""
    g = ref_golden()[name]
    inf = product_class.info(ref_jpeg(name))
    assert (inf["ok"], inf["lasterror"]) == (g["info"]["ok"], g["info"]["lasterror"]) == (0, 2)
    with pytest.raises(J.JdaError) as e:
        J.decode_to_host(gpu_ctx, ref_jpeg(name), RGB8888, 0)
    assert e.value.code == 2


@pytest.mark.parametrize("name", FAIL_IN_DECODE)
def test_reference_tests_5_6_fail_in_decode(name, product_class, gpu_ctx, oracle):
    """main.cpp:176-198: corrupt2 / corrupt3 (and the truncated main image of thumb_test) open fine and fail inside decode()
    with JPEG_DECODE_ERROR after delivering the strips in front of the bad MCU -- the same verdict here, the same pixels in
    every MCU this path decodes before its bad MCU, and the same JPEGDRAW records for the strips it delivers.  (The reference
    goes on into stale file-buffer bytes where the stream has run out, so it may deliver more strips: DESIGN.md 3.)"""
    jpeg = ref_jpeg(name)
    g = ref_golden()[name]
    p = J.PreparedImage(jpeg)
    nok = p.block_index()[1]
    for pt, opt in modes_of(name)[:6]:
        fr = g["frames"]["%d:%d" % (pt, opt)]
        rc, got, geo = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
        assert rc == 2 and (fr["rc"], fr["err"]) == (0, 2), (name, pt, opt, rc)
        r = product_class.decode_cb(jpeg, pt, opt, want_log=True)
        assert (r["rc"], r["last_error"]) == (0, 2)
        assert r["n_calls"] <= fr["draw_calls"]
        # pixels of the MCUs in front of the bad one: the oracle restatement (pinned to the real reference) decodes the same stream
        orc, want, err = oracle.decode_canvas(jpeg, pt, opt)
        assert (orc, err) == (0, 2)
        sh = geo["canvas_h"] // p.info.mcus_y                       # MCU height in output rows
        full_rows = (nok // p.info.mcus_x) * sh
        assert np.array_equal(got[:full_rows], want[:full_rows]), (name, pt, opt)
        mw = geo["canvas_w"] // p.info.mcus_x * geo["bpp"]
        part = (nok % p.info.mcus_x) * mw
        assert np.array_equal(got[full_rows: full_rows + sh, :part], want[full_rows: full_rows + sh, :part]), (name, pt, opt)
    p.close()


def test_fixtures_equal_the_real_reference_when_present(product_class, ref_scalar):
    """the same driver against oracle/_ref itself (where the prebuilt reference travelled): frames, logs, crops, offsets"""
    for name in ("tulips", "zebra", "croptest", "perf"):
        jpeg = ref_jpeg(name)
        for pt, opt, crop, mm in ((RGB565_LE, 0, None, 0), (RGB8888, 0, (16, 32, 100, 60), 0), (GRAY8, J.SCALE_HALF, None, 5), (RGB565_BE, USES_DMA, None, 0)):
            a = product_class.decode_cb(jpeg, pt, opt, crop=crop, max_mcus=mm, want_log=True, xoff=3, yoff=9)
            b = ref_scalar.decode_cb(jpeg, pt, opt, crop=crop, max_mcus=mm, want_log=True, xoff=3, yoff=9)
            assert a["rc"] == b["rc"] == 1 and np.array_equal(a["log"], b["log"]) and a["dma_reuse"] == b["dma_reuse"], (name, pt, opt)
            h = a["canvas"].shape[0] - 16
            assert np.array_equal(a["canvas"][:h], b["canvas"][:h]), (name, pt, opt)
    # framebuffer mode (jpeg.inl:5114-5124) on the fixtures whose width is an MCU multiple
    for name in ("tulips", "zebra", "croptest", "st_peters"):
        for pt in (RGB565_LE, RGB8888, GRAY8):
            rc_a, fa = product_class.decode_fb(ref_jpeg(name), pt, 0)
            rc_b, fb = ref_scalar.decode_fb(ref_jpeg(name), pt, 0)
            inf = ref_golden()[name]["info"]
            n = inf["width"] * inf["height"] * {RGB565_LE: 2, RGB8888: 4, GRAY8: 1}[pt]
            assert rc_a == rc_b == 1 and np.array_equal(fa[:n], fb[:n]), (name, pt)


def test_reference_test_program(gpu_ctx):
    """tests/ref_main/jpegtest_amd.cpp: the reference's own test program (MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp:74-300, tests
    1, 2, 4-10 and both fuzz loops; test 3 is a CPU timing comparison) restated against include/JPEGDEC.h + libjpegdec_amd.so"""
    subprocess.run(["make", "jpegtest"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    p = subprocess.run([os.path.join(ROOT, "tests", "ref_main", "jpegtest_amd"), os.path.join(ROOT, "tests", "golden", "ref")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "Total tests: 11, 11 passed, 0 failed" in p.stdout, p.stdout[-3000:]


def test_crop_aware_decode_launches_only_the_crop(gpu_ctx):
    """VERDICT r1 task 8: crop_area.ino's rectangle on croptest -- the launch plan holds the tiles of the kept MCUs only
    (<= 40 % of the image's), the pixels of the rectangle are those of the full decode, nothing outside it is written."""
    jpeg = ref_jpeg("croptest")
    p = J.PreparedImage(jpeg)
    cx, cy, cw, ch = J.crop_round(p.info, *CROP_INO)
    assert (cx, cy, cw, ch) == (112, 64, 128, 112)
    mw, mh = p.info.mcu_w, p.info.mcu_h
    xs = [x for x in range(p.info.mcus_x) if not (x * mw < cx or x * mw > cx + cw)]
    rect = (xs[0], (cy + mh - 1) // mh, xs[-1] + 1, min(p.info.mcus_y, (cy + ch + mh - 1) // mh))
    for pt in (RGB8888, RGB565_LE, GRAY8):
        for opt, sh in ((0, 0), (J.SCALE_HALF, 1), (J.SCALE_QUARTER, 2), (J.SCALE_EIGHTH, 3)):      # (the 1/4 and 1/8 kernels take the same tile lists)
            rc0, full, g = J.decode_to_host(gpu_ctx, jpeg, pt, opt)
            rc1, part, g1, tiles = J.binding.decode_to_host_rect(gpu_ctx, jpeg, pt, opt, rect)
            assert rc0 == rc1 == 0 and tiles[1] == 30 and tiles[0] <= 0.4 * tiles[1], tiles
            bpp = g["bpp"]
            ys, xb = slice(rect[1] * (mh >> sh), rect[3] * (mh >> sh)), slice(rect[0] * (mw >> sh) * bpp, rect[2] * (mw >> sh) * bpp)
            assert np.array_equal(part[ys, xb], full[ys, xb]), (pt, opt)
            outside = part.copy()
            outside[ys, xb] = 0
            assert not outside.any(), (pt, opt)
    p.close()
    # a gray file wide enough for whole tiles that start off a tile boundary (and off a 4-pixel one): the 1/4-scale kernel's shared store
    # and the DC thumbnail kernel's packed path must hand over to their general paths
    from jpegdec_amd.synth import synth_jpeg
    wide = synth_jpeg(2304, 40, "gray", seed=77)
    for x0 in (1, 2, 64, 67):
        rect = (x0, 1, x0 + 200, 4)
        for pt, opt, sh in ((GRAY8, J.SCALE_QUARTER, 2), (GRAY8, J.SCALE_EIGHTH, 3), (RGB565_LE, J.SCALE_QUARTER, 2), (GRAY8, 0, 0)):
            rc0, full, g = J.decode_to_host(gpu_ctx, wide, pt, opt)
            rc1, part, g1, tiles = J.binding.decode_to_host_rect(gpu_ctx, wide, pt, opt, rect)
            assert rc0 == rc1 == 0
            bpp = g["bpp"]
            ys, xb = slice(rect[1] * (8 >> sh), rect[3] * (8 >> sh)), slice(rect[0] * (8 >> sh) * bpp, rect[2] * (8 >> sh) * bpp)
            assert np.array_equal(part[ys, xb], full[ys, xb]), (x0, pt, opt)
            outside = part.copy()
            outside[ys, xb] = 0
            assert not outside.any(), (x0, pt, opt)
