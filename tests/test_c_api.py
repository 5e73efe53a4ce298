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


NODE_EXE = os.path.join(ROOT, "tests", "capi_c", "node_user")


@pytest.fixture(scope="module")
def node_user():
    subprocess.run(["make", "nodeuser"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return NODE_EXE


def test_node_shard_rule_is_the_ranks_rule(product_lib):
    """jda_node deals a list out in contiguous blocks whose sizes differ by at most one: the same rule as sharding.shard_range
    (bench.py's ranks), every image owned exactly once."""
    import ctypes as C
    from jpegdec_amd.sharding import shard_range
    product_lib.jda_node_shard_of.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    product_lib.jda_node_shard_of.restype = None
    for nd in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9, 64, 8192):
            seen = 0
            for k in range(nd):
                first, count = C.c_int32(-1), C.c_int32(-1)
                product_lib.jda_node_shard_of(nd, n, k, C.byref(first), C.byref(count))
                lo, hi = shard_range(n, k, nd)
                assert (first.value, first.value + count.value) == (lo, hi)
                seen += count.value
            assert seen == n


def test_node_program_without_gpu_fails_loudly(node_user, tmp_path):
    import jpegdec_amd as J
    if J.load_library().jda_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([node_user, _write(tmp_path, "c420_333x217"), "4"], capture_output=True, text=True)
    assert r.returncode == 6 and "no node" in r.stdout          # JDA_ERROR_NO_DEVICE: there is no CPU decode path


@pytest.mark.gpu
def test_node_program_decodes_a_list_on_every_device(node_user, tmp_path, oracle, gpu_ctx):
    """A C caller shards a list over the node without Python: every image decoded (status 0), every surface's device-made
    checksum equal to the checksum of the oracle's canvas."""
    import jpegdec_amd as J
    for name, n in (("c420_333x217", 13), ("c444_333x217", 5)):
        r = subprocess.run([node_user, _write(tmp_path, name), str(n)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        f = r.stdout.split()
        assert int(f[1]) >= 1 and int(f[3]) == n and int(f[5]) == n and int(f[9]) == 1, r.stdout
        rc, want, err = oracle.decode_canvas(jpeg_for(name), J.RGB8888, 0)
        assert rc == 1 and int(f[7], 16) == J.surface_checksum_host(want), r.stdout


def test_object_semantics_of_the_reference_boundary(tmp_path):
    """tests/capi_c/semantics_user.cpp: JPEGDEC objects are copied, assigned, moved and kept in containers like the reference's plain
    struct (src/JPEGDEC.h:286); a C JPEGIMAGE needs no initialisation and no JPEG_close for RAM sources (src/JPEGDEC.cpp:232-236) --
    2,000 handles opened and never closed, every one of them live (the state is the caller's struct, as the reference's is: no table,
    no limit); what is set on one stays with it; a struct copy is not a handle; file sources own their bytes until JPEG_close."""
    subprocess.run(["make", "semuser"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = subprocess.run([os.path.join(ROOT, "tests", "capi_c", "semantics_user"), _write(tmp_path, "c420_333x217")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_node_over_eight_stub_devices():
    """tests/node_stub: jda_node.cpp (host code above the C-ABI only) linked against a stand-in for the device half with EIGHT pretend
    devices, built with ThreadSanitizer: the contiguous shard rule, ONE PERSISTENT host thread per device that makes every call of that
    device (none comes from the caller's thread: its current HIP device is never touched), statuses in list order, the submit flags
    handed through, depth, a device that refuses its block, a short list, a subset of devices.  No GPU needed."""
    subprocess.run(["make", "nodestub"], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = subprocess.run([os.path.join(ROOT, "tests", "node_stub", "node_stub_user")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
