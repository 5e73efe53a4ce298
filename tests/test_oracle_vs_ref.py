"""The oracle restatement against the real reference itself (oracle/_ref, built from
/root/reference by oracle/Makefile).  Skipped where the prebuilt reference is absent."""
import glob
import os
import re

import numpy as np
import pytest

from oracle.loader import GRAY8, RGB8888, USES_DMA, load_c_array_header
from tests.cases import OPTIONS, PIXEL_TYPES, SYNTH_CASES, jpeg_for

REF_ROOT = "/root/reference"


def _compare_all_modes(jpeg, oracle, ref):
    inf = ref.info(jpeg)
    for pt in PIXEL_TYPES:
        for opt in OPTIONS:
            if inf["subsample"] == 0 and pt == RGB8888:
                continue
            if inf["subsample"] == 0x12 and pt == RGB8888 and (opt & 4):
                continue                          # reference UB (jpeg.inl:4620 writes through the address of a local)
            r = ref.decode_cb(jpeg, pt, opt)
            rc, canvas, err = oracle.decode_canvas(jpeg, pt, opt)
            assert r["rc"] == 1 and rc == 1
            sh = r["scale_shift"]
            h = (inf["height"] + (1 << sh) - 1) >> sh      # rows below the image are clipped by the draw callback
            want = r["canvas"][:h, : canvas.shape[1]]
            assert np.array_equal(canvas[:h], want), (pt, opt)
            assert np.array_equal(oracle.draw_plan(jpeg, pt, opt),
                                  ref.decode_cb(jpeg, pt, opt, want_log=True)["log"]), (pt, opt)


@pytest.mark.parametrize("name", sorted(SYNTH_CASES))
def test_oracle_equals_reference_synthetic(name, oracle, ref_scalar):
    _compare_all_modes(jpeg_for(name), oracle, ref_scalar)


@pytest.mark.needs_reference
@pytest.mark.parametrize("image", ["tulips", "zebra", "st_peters", "sciopero"])
def test_oracle_equals_reference_on_its_own_test_images(image, oracle, ref_scalar):
    path = os.path.join(REF_ROOT, "test_images", image + ".h")
    if not os.path.exists(path):
        pytest.skip("reference tree absent")
    _compare_all_modes(load_c_array_header(path), oracle, ref_scalar)


def test_draw_plan_options(oracle, ref_scalar):
    jpeg = jpeg_for("c420_333x217")
    for pt in (0, 2, 3):
        for opt in (0, 2, USES_DMA):
            for mm in (0, 3):
                want = ref_scalar.decode_cb(jpeg, pt, opt, max_mcus=mm, want_log=True)["log"]
                got = oracle.draw_plan(jpeg, pt, opt, max_mcus=mm, uses_dma=bool(opt & USES_DMA))
                assert np.array_equal(got, want), (pt, opt, mm)


@pytest.mark.needs_reference
def test_closed_form_tables_equal_reference_literals(oracle):
    """ucRangeTable / usGrayTo565 / usRangeTableR,G,B / iScaleBits / cZigZag2 as literals in jpeg.inl."""
    src = os.path.join(REF_ROOT, "src", "jpeg.inl")
    if not os.path.exists(src):
        pytest.skip("reference tree absent")
    txt = open(src, errors="replace").read()

    def arr(name):
        m = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", txt, re.S)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        return [int(v, 0) for v in re.findall(r"0[xX][0-9a-fA-F]+|\d+", body)]

    L = oracle.lib
    assert arr("ucRangeTable") == [L.orc_range_limit(i) for i in range(1024)]
    assert arr("usGrayTo565") == [L.orc_gray565(i) for i in range(256)]
    for comp, nm in enumerate(("usRangeTableR", "usRangeTableG", "usRangeTableB")):
        assert arr(nm) == [L.orc_range565(comp, i) for i in range(1024)], nm
    assert arr("iScaleBits") == [L.orc_aan_scale(i) for i in range(64)]
    assert arr("cZigZag2") == [L.orc_zigzag_to_natural(i) for i in range(64)]


@pytest.mark.parametrize("name", ["c420_333x217", "c444_333x217", "gray_333x217", "c422_333x217", "c440_200x120"])
def test_oracle_equals_reference_on_corrupted_scans(name, oracle, ref_scalar):
    """Random byte corruptions inside the entropy-coded data: as long as the stream does not run out of bits the
    oracle must reproduce the reference's output (garbage included) and its verdict.  (A stream that over-reads
    makes the reference decode stale bytes of its 2 KiB file buffer: out of contract)"""
    import jpegdec_amd as J
    base = bytearray(jpeg_for(name))
    sos = bytes(base).index(b"\xff\xda")
    rng = np.random.default_rng(23)
    checked = 0
    for it in range(60):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
        jb = bytes(b)
        try:
            p = J.PreparedImage(jb)
        except J.JdaError:
            continue
        idx, nok = p.block_index()
        if (int(idx[-1]) >> 7) + ((int(idx[-1]) & 127) + 7) // 8 > len(p.scan()):
            continue
        pt = 3 if name.startswith("gray") else 2
        r = ref_scalar.decode_cb(jb, pt, 0)
        rc, canvas, err = oracle.decode_canvas(jb, pt, 0)
        assert (r["rc"] == 1) == (rc == 1), (name, it, r["rc"], r["last_error"], rc, err)
        if rc == 1:
            h = ref_scalar.info(jb)["height"]
            assert np.array_equal(canvas[:h], r["canvas"][:h, : canvas.shape[1]]), (name, it)
            checked += 1
    assert checked >= 10


@pytest.mark.parametrize("luma_hv", [(2, 2), (1, 1), (2, 1), (1, 2)])
def test_oracle_equals_reference_duplicate_eob_code(luma_hv, oracle, ref_scalar):
    """A DHT that codes the end-of-block symbol twice: the reference's per-code LUTs decode it; so must the oracle."""
    from jpegdec_amd.synth import encode_jpeg_custom, value_noise_image
    _compare_all_modes(encode_jpeg_custom(value_noise_image(333, 217, 3, 78), 85, luma_hv, dup_eob=True), oracle, ref_scalar)
