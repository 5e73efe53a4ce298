"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares;
without a device the GPU entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import jpegdec_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(product_lib):
    hdr = open(os.path.join(ROOT, "include", "jpegdec_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(jda_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    for n in names:
        assert hasattr(product_lib, n), "libjpegdec_amd.so does not export " + n


def test_jpegdec_class_is_exported(product_lib):
    out = subprocess.run(["nm", "-DC", J.library_path()], stdout=subprocess.PIPE, text=True).stdout
    for m in ("JPEGDEC::openRAM", "JPEGDEC::openFLASH", "JPEGDEC::decode", "JPEGDEC::setPixelType",
              "JPEGDEC::setFramebuffer", "JPEGDEC::setMaxOutputSize", "JPEGDEC::setUserPointer",
              "JPEGDEC::setCropArea", "JPEGDEC::getCropArea", "JPEGDEC::getWidth", "JPEGDEC::getHeight",
              "JPEGDEC::getBpp", "JPEGDEC::getSubSample", "JPEGDEC::getOrientation", "JPEGDEC::getJPEGType",
              "JPEGDEC::hasThumb", "JPEGDEC::getLastError", "JPEGDEC::getPixelType", "JPEGDEC::close"):
        assert m + "(" in out, m


def test_no_device_is_an_error_not_a_fallback(product_lib):
    if product_lib.jda_device_count() > 0:
        pytest.skip("a GPU is present")
    err = C.c_int32(0)
    assert not product_lib.jda_create(0, C.byref(err))
    assert err.value == 6                                   # JDA_ERROR_NO_DEVICE
    with pytest.raises(J.JdaError):
        J.Context(0)
    assert product_lib.jda_decode_to_host(None, b"x", 1, 2, 0, None, 0, 0) == 6


def test_product_does_not_link_the_oracle(product_lib):
    out = subprocess.run(["ldd", J.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out and "jpegdec_ref" not in out
    syms = subprocess.run(["nm", "-D", J.library_path()], stdout=subprocess.PIPE, text=True).stdout
    assert "orc_" not in syms
