"""Sequences of calls on ONE JPEGDEC object -- setPixelType / setMaxOutputSize / setCropArea / decode / getters / close + reopen --
replayed through the product's class and compared, value by value, with what the unmodified reference recorded for the same sequence
(tests/golden/script_walks.json, made by tests/golden/make_script_walk_golden.py where /root/reference exists): what one call leaves
behind for the next (crop, pixel type, the object describing the EXIF thumbnail after it was decoded, error codes)."""
import json
import os

import pytest

from oracle.loader import RefDecoder
from tests.cases import jpeg_for
from tests.ref_fixtures import ref_jpeg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def product_class(gpu_ctx):
    import subprocess
    subprocess.run(["make", "classshim"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))


def test_call_sequences_leave_the_state_the_reference_leaves(product_class):
    from tests.walks import check_script_walks
    check_script_walks(product_class)


def test_a_second_decode_equals_a_first_one(product_class):
    """Not the reference's behaviour -- there a second decode() on an open object fails or decodes from the wrong place (the file position
    is not rewound) -- but the product's: decode() leaves nothing behind, so the same object decodes again, and what it delivers is what a
    fresh object delivers.  (JPEG_EXIF_THUMBNAIL is the exception by design: the object describes the thumbnail afterwards, jpeg.inl:4967-4976.)"""
    import numpy as np
    rng = np.random.default_rng(3)
    names = ["c420_333x217", "c444_333x217", "gray_333x217", "c422_333x217", "c440_200x120", "c420_640x368_rstrow", "p420_200x120", "pgray_100x100",
             "ref:tulips", "ref:sciopero", "ref:corrupt3", "ref:thumb_test"]
    checked = 0
    for name in names:
        jpeg = ref_jpeg(name[4:]) if name.startswith("ref:") else jpeg_for(name)
        gray, prog = name.startswith(("gray", "pgray")), name.startswith("p")
        for _ in range(6):
            def one():
                pt = int(rng.integers(0, 4))
                if (gray and pt == 2) or (prog and not gray and pt == 3):
                    pt = 0
                opt = int((0, 2, 4, 8)[int(rng.integers(0, 4))])
                if prog and opt == 4:
                    opt = 2
                if name == "c440_200x120" and pt == 2 and opt == 4:
                    opt = 0
                return [[1, pt, 0, 0, 0], [2, int(rng.integers(1, 40)), 0, 0, 0], [4, int(rng.integers(0, 30)), int(rng.integers(0, 20)), opt, 0], [5, 0, 0, 0, 0]]
            a, b = one(), one()
            both = product_class.run_script(jpeg, a + b)
            alone = product_class.run_script(jpeg, b)
            assert both[-len(alone) + 1:] == alone[1:], (name, a, b, both, alone)
            checked += 1
    assert checked >= 60


def test_a_strip_wider_than_the_reference_s_buffer(product_class):
    """sciopero is 300 pixels wide (18.75 MCUs): with a decode x offset the reference widens the last strip of every MCU row to what is
    left of the row (jpeg.inl:5328-5335) -- 240 pixels x 16 rows of RGB8888 into usPixels[2048], i.e. over the tables behind it.  The
    product hands the callback the strips of that plan from a buffer of their size (it used to write them into a 4 KB one)."""
    jpeg = ref_jpeg("sciopero")
    for pt in (2, 0, 3):
        for opt in (128, 0):
            out = product_class.run_script(jpeg, [[1, pt, 0, 0, 0], [4, 45, 2, opt, 0], [4, 45, 2, opt, 0], [5, 0, 0, 0, 0]])
            assert out[0] == 1 and out[1:3] == [1, 0] and out[6:8] == [1, 0] and out[1:6] == out[6:11], out
