"""pytest configuration: markers, paths, shared fixtures.

`-m "not gpu"` : oracle vs golden vectors / the real reference (when oracle/_ref exists), host
                 front end, CPU wave emulator (tests/hostsim), C-ABI symbol check, gloo sharding.
`-m gpu`       : parity tests proper -- the HIP path through the C-ABI vs the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def _make(*targets):
    subprocess.run(["make", *targets], cwd=ROOT, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.fixture(scope="session")
def built_checkers():
    """liboracle.so (+ oracle/_ref where the reference tree exists) and the wave emulator."""
    _make("oracle", "hostsim")
    return True


@pytest.fixture(scope="session")
def oracle(built_checkers):
    from oracle.loader import OracleDecoder

    return OracleDecoder()


@pytest.fixture(scope="session")
def ref_scalar(built_checkers):
    from oracle.loader import RefDecoder, ref_available

    if not ref_available(False):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    return RefDecoder(False)


@pytest.fixture(scope="session")
def product_lib():
    """libjpegdec_amd.so: built by hipcc if missing (cross-compiles without a GPU)."""
    import jpegdec_amd

    if not os.path.exists(jpegdec_amd.library_path()):
        _make("lib")
    return jpegdec_amd.load_library()


@pytest.fixture(scope="session")
def gpu_ctx(product_lib):
    import jpegdec_amd

    ctx = jpegdec_amd.Context(0)   # raises JdaError(NO_DEVICE) without a GPU: gpu tests must not silently pass
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def hostsim(built_checkers):
    """tests/hostsim: the kernels' per-lane logic compiled for the CPU (a wave emulator; test infrastructure only)."""
    import ctypes as C

    # (JDA_HOSTSIM_LIBRARY: another build of the emulator, e.g. one with -DJDA_SEG_BYTES=64u)
    lib = C.CDLL(os.environ.get("JDA_HOSTSIM_LIBRARY") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "hostsim", "libjda_hostsim.so"))
    lib.hostsim_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return lib
