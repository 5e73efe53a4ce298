"""Random walks over the public API, replayed through the product's JPEGDEC class and compared with what the unmodified reference did
with the same calls (tests/walks.py; tests/golden/api_walks.json, made by tests/golden/make_api_walk_golden.py where /root/reference
exists; every walk ran in a process of its own there because some combinations crash the reference -- those are not compared)."""
import os

import pytest

from oracle.loader import RefDecoder
from tests.walks import check_api_walks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product_class(gpu_ctx):
    import subprocess
    subprocess.run(["make", "classshim"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return RefDecoder(False, path=os.path.join(ROOT, "tests", "libjpegdec_class_shim.so"))


def test_api_walks_match_the_reference(product_class):
    check_api_walks(product_class)
