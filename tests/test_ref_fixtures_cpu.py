"""The reference's own fixtures (tests/golden/ref/) on the CPU side: the oracle restatement and the kernel logic
(tests/hostsim) against the hashes the REAL reference produced for them (tests/golden/ref/ref_golden.json), and the
front end's verdict on the corrupt files (MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp:166-216).  The GPU runs of the same
fixtures are in tests/test_gpu_ref_fixtures.py."""
import ctypes as C

import numpy as np
import pytest

import jpegdec_amd as J
from oracle.loader import digest
from tests.ref_fixtures import BIG, FAIL_IN_DECODE, GOOD, REJECTED_AT_OPEN, modes_of, ref_golden, ref_jpeg


@pytest.mark.parametrize("name", sorted(ref_golden()))
def test_fixture_bytes_are_the_reference_vectors(name):
    g = ref_golden()[name]
    assert len(ref_jpeg(name)) == g["jpeg_len"] and digest(ref_jpeg(name)) == g["jpeg_sha"]


@pytest.mark.parametrize("name", GOOD + BIG)
def test_oracle_restatement_equals_reference_hashes(name, oracle):
    jpeg = ref_jpeg(name)
    g = ref_golden()[name]
    for pt, opt in modes_of(name):
        fr = g["frames"]["%d:%d" % (pt, opt)]
        rc, canvas, err = oracle.decode_canvas(jpeg, pt, opt)
        assert (rc, err) == (fr["rc"], fr["err"]) == (1, 0)
        assert digest(canvas[: fr["h"], : fr["w"] * fr["bpp"]]) == fr["sha"], (name, pt, opt)
        assert digest(oracle.draw_plan(jpeg, pt, opt)) == fr["log_sha"], (name, pt, opt)


@pytest.mark.parametrize("name", GOOD)
def test_kernel_logic_equals_reference_hashes(name, hostsim):
    """The per-lane kernel code (jda_device_core.h through the CPU wave emulator) on real photographs: camera quantisers,
    1.4-4.1 bit/px, DRI, hundreds of window-truncation events (perf.jpg: 1,793) -- frame hash == the real reference's."""
    jpeg = ref_jpeg(name)
    g = ref_golden()[name]
    inf = g["info"]
    p = J.PreparedImage(jpeg)
    for pt, opt in ((2, 0), (0, 0), (1, 2), (3, 0), (2, 4), (3, 8), (0, 64)):
        fr = g["frames"]["%d:%d" % (pt, opt)]
        geo = p.geometry(pt, opt)
        got = np.full((geo["canvas_h"], geo["canvas_w"] * geo["bpp"]), 0x33, np.uint8)
        assert hostsim.hostsim_decode(jpeg, len(jpeg), pt, opt, got.ctypes.data_as(C.c_void_p), got.shape[1],
                                      p.info.mcus_x * p.info.mcu_w, p.info.mcus_y * p.info.mcu_h) == 0
        assert digest(got[: fr["h"], : fr["w"] * fr["bpp"]]) == fr["sha"], (name, pt, opt)
    if name == "perf":
        assert p.truncation_events() > 1000
    assert inf["width"] == p.info.width and inf["height"] == p.info.height and inf["subsample"] == p.info.subsample
    p.close()


@pytest.mark.parametrize("name", REJECTED_AT_OPEN)
def test_header_rejected_like_the_reference(name):
    """corrupt1 (invalid header offsets) and corrupt4 ("FPE1": sampling factors 5x3) fail in JPEGParseInfo with JPEG_DECODE_ERROR."""
    g = ref_golden()[name]
    assert g["info"]["ok"] == 0
    assert J.parse(ref_jpeg(name))["status"] == g["info"]["lasterror"] == 2


@pytest.mark.parametrize("name", FAIL_IN_DECODE)
def test_decode_failure_verdict(name, oracle):
    """corrupt2 / corrupt3 / the truncated main image of thumb_test open fine and fail inside decode() with JPEG_DECODE_ERROR:
    the pre-scan must stop short of the last MCU (that is what turns into JPEG_DECODE_ERROR on the GPU path)."""
    g = ref_golden()[name]
    assert g["info"]["ok"] == 1 and all((f["rc"], f["err"]) == (0, 2) for f in g["frames"].values())
    p = J.PreparedImage(ref_jpeg(name))
    idx, nok = p.block_index()
    assert nok < p.n_mcus
    assert (p.info.width, p.info.height) == (g["info"]["width"], g["info"]["height"])
    p.close()
