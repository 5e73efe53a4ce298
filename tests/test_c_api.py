"""The C flavour of the reference API (JPEG_openFile / JPEG_getWidth / JPEG_decode / JPEG_close, reference
src/JPEGDEC.h:288-309) used from a plain C program (tests/capi_c/c_user.c, compiled with the C compiler) the way
linux/examples/c_cmdline/main.c uses the reference.  Opening only parses, so that part runs everywhere;
decoding needs the GPU and must fail loudly without one."""
import os
import subprocess

import numpy as np
import pytest

from tests.cases import jpeg_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "capi_c", "c_user")


@pytest.fixture(scope="module")
def c_user(tmp_path_factory):
    subprocess.run(["make", "cuser"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return EXE


def _write(tmp_path, name):
    p = tmp_path / (name + ".jpg")
    p.write_bytes(jpeg_for(name))
    return str(p)


def test_c_program_opens_and_reports_geometry(c_user, tmp_path, oracle):
    for name in ("c420_333x217", "c422_333x217", "gray_333x217", "c444_8x8_q30"):
        r = subprocess.run([c_user, _write(tmp_path, name)], capture_output=True, text=True)
        inf = oracle.info(jpeg_for(name))
        assert r.returncode == 0
        w_h, sub, bpp = r.stdout.split()
        assert w_h == "%dx%d" % (inf["width"], inf["height"]) and int(sub) == inf["subsample"]
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"\x00" * 400)
    assert subprocess.run([c_user, str(bad)], capture_output=True, text=True).returncode == 101


def test_c_program_decode_without_gpu_fails_loudly(c_user, tmp_path):
    import jpegdec_amd as J
    if J.load_library().jda_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([c_user, _write(tmp_path, "c420_333x217"), str(tmp_path / "o.rgba")], capture_output=True, text=True)
    assert r.returncode == 6          # JPEG_ERROR_NO_DEVICE: there is no CPU decode path


@pytest.mark.gpu
def test_c_program_decodes_the_oracle_frame(c_user, tmp_path, oracle, gpu_ctx):
    for name in ("c420_333x217", "c422_333x217", "c440_200x120"):
        out = tmp_path / (name + ".rgba")
        r = subprocess.run([c_user, _write(tmp_path, name), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        rc, want = oracle.decode_frame(jpeg_for(name), 2, 0)
        got = np.fromfile(str(out), dtype=np.uint8).reshape(want.shape)
        assert np.array_equal(got, want), name
