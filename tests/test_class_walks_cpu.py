"""The class's HOST logic without a GPU: jpegdec_amd/csrc/JPEGDEC.cpp + the host front end over a CPU stand-in for the four device entry
points (tests/class_cpu/stub_runtime.cpp: pixels from the oracle's restatement, MCU counts from the product's serial pre-scan), driven by
the same shim as the reference (oracle/ref_shim.cpp) -- option handling, crop rounding, the JPEGDRAW replay, framebuffer wrap / clip, EXIF
thumbnails, error codes, state between calls -- against the 1,000 walks recorded from the unmodified reference; and the same walks once
more under AddressSanitizer / UBSan (a strip wider than the buffer behind it was a heap overflow the GPU walks found in round 2)."""
import json
import os
import subprocess

import pytest

from oracle.loader import RefDecoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def class_cpu():
    subprocess.run(["make", "classcpu"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return RefDecoder(False, path=os.path.join(ROOT, "tests", "class_cpu", "libjpegdec_class_cpu.so"))


def test_api_walks_through_the_class_on_the_cpu(class_cpu):
    from tests.walks import check_api_walks
    check_api_walks(class_cpu)


def test_call_sequences_through_the_class_on_the_cpu(class_cpu):
    from tests.walks import check_script_walks
    check_script_walks(class_cpu)


def test_walks_under_asan_ubsan(class_cpu):
    """every recorded walk (the ones that kill the reference too) through the sanitizer build: tests/class_cpu/walks_main.cpp"""
    from tests.cases import jpeg_for
    from tests.ref_fixtures import ref_jpeg
    import tempfile
    A = json.load(open(os.path.join(ROOT, "tests", "golden", "api_walks.json")))["walks"]
    S = json.load(open(os.path.join(ROOT, "tests", "golden", "script_walks.json")))["scripts"]
    with tempfile.TemporaryDirectory() as td:
        names = sorted({x["walk"]["image"] for x in A} | {x["script"]["image"] for x in S})
        for k, n in enumerate(names):
            open(os.path.join(td, "img%d.jpg" % k), "wb").write(ref_jpeg(n[4:]) if n.startswith("ref:") else jpeg_for(n))
        lines = []
        for x in A:
            w = x["walk"]
            c = w["crop"] or [-1, -1, -1, -1]
            lines.append("W %d %d %d %d %d %d %d %d %d %d %d" % (names.index(w["image"]), 1 if w.get("fb") else 0, w["pixel_type"], w["options"], w["max_mcus"], w["xoff"], w["yoff"], *c))
        for x in S:
            sc = x["script"]
            lines.append("S %d %d %s" % (names.index(sc["image"]), len(sc["ops"]), " ".join(str(v) for op in sc["ops"] for v in op)))
        open(os.path.join(td, "walks.txt"), "w").write("\n".join(lines) + "\n")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([os.path.join(ROOT, "tests", "class_cpu", "walks_asan"), td, str(len(names))], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
        assert "walks done" in r.stdout and int(r.stdout.split("walks done")[0].split()[-1]) == len(lines), r.stdout[-300:]


def test_large_images_leave_the_device_strip_major(class_cpu):
    """tests/strip_major_cases.py over the stand-in device: the class's side of the strip-major path (plan regularity, strip addresses,
    the bands' replay) against the unmodified reference."""
    from oracle.loader import ref_available
    from tests.strip_major_cases import check_strip_major_decodes
    if not ref_available(False):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    check_strip_major_decodes(class_cpu, RefDecoder(False))
