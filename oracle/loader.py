"""oracle/loader.py -- ctypes access to the CHECKERS (test infrastructure only).

Loads
  * oracle/_ref/libjpegdec_ref_scalar.so  -- the real reference, scalar integer path (parity oracle)
  * oracle/_ref/libjpegdec_ref_sse2.so    -- the real reference, SSE2 build (CPU timing baseline)
  * oracle/liboracle.so                   -- our C restatement (oracle/jpegdec_oracle.c)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing here reads /root/reference at run time (the .so files are prebuilt by oracle/Makefile).
"""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# pixel types / options: values are the reference's public constants (src/JPEGDEC.h:68-75,102-111)
RGB565_LE, RGB565_BE, RGB8888, GRAY8 = 0, 1, 2, 3
SCALE_HALF, SCALE_QUARTER, SCALE_EIGHTH = 2, 4, 8
EXIF_THUMBNAIL, LUMA_ONLY, USES_DMA = 32, 64, 128

BYTES_PER_PIXEL = {RGB565_LE: 2, RGB565_BE: 2, RGB8888: 4, GRAY8: 1}


def digest(buf) -> str:
    """Short content hash used for golden vectors (sha256, first 16 hex digits)."""
    import hashlib

    if isinstance(buf, np.ndarray):
        data = np.ascontiguousarray(buf).view(np.uint8)
    else:
        data = np.frombuffer(memoryview(buf), dtype=np.uint8)
    return hashlib.sha256(data).hexdigest()[:16]


def mcu_dims(subsample: int):
    """MCU width/height in pixels for the reference's ucSubSample byte (jpeg.inl:5008-5046)."""
    return {0x00: (8, 8), 0x11: (8, 8), 0x12: (8, 16), 0x21: (16, 8), 0x22: (16, 16)}[subsample]


def scale_shift(options: int) -> int:
    if options & SCALE_HALF:
        return 1
    if options & SCALE_QUARTER:
        return 2
    if options & SCALE_EIGHTH:
        return 3
    return 0


class RefDecoder:
    """The real reference (bitbank2/JPEGDEC) behind oracle/ref_shim.cpp."""

    def __init__(self, simd: bool = False, path: str = None):
        """path: load another build of the same shim (tests use this for the product's class)."""
        if path is None:
            name = "libjpegdec_ref_sse2.so" if simd else "libjpegdec_ref_scalar.so"
            path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_is_simd.restype = C.c_int
        L.ref_get_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.ref_get_info.restype = C.c_int
        L.ref_decode_cb.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_decode_cb.restype = C.c_int
        L.ref_decode_cb2.argtypes = L.ref_decode_cb.argtypes + [C.POINTER(C.c_int)]
        L.ref_decode_cb2.restype = C.c_int
        L.ref_decode_fb.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.ref_decode_fb.restype = C.c_int
        L.ref_bench.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        L.ref_bench.restype = C.c_double
        assert bool(L.ref_is_simd()) == simd

    def info(self, data: bytes):
        arr = (C.c_int * 10)()
        rc = self.lib.ref_get_info(data, len(data), arr)
        keys = ["width", "height", "subsample", "bpp", "jpegtype", "orientation", "hasthumb",
                "thumbw", "thumbh", "lasterror"]
        d = dict(zip(keys, list(arr)))
        d["ok"] = rc
        return d

    def decode_cb(self, data: bytes, pixel_type=RGB8888, options=0, max_mcus=0, xoff=0, yoff=0,
                  crop=None, used_only=False, stop_after=0, want_log=False, canvas_shape=None):
        """Callback-mode decode assembled into an MCU-padded canvas.

        Returns dict(rc, canvas (rows x pitch_bytes uint8), n_calls, dma_reuse, last_error, log)."""
        inf = self.info(data)
        if not inf["ok"]:
            return dict(rc=-1, canvas=None, n_calls=0, dma_reuse=0, last_error=inf["lasterror"], log=None)
        mw, mh = mcu_dims(inf["subsample"])
        # a progressive file gets JPEG_SCALE_EIGHTH OR-ed in (jpeg.inl:4964-4966) before the HALF / QUARTER / EIGHTH chain
        sh = scale_shift(options) if inf["jpegtype"] == 0 else scale_shift(options | SCALE_EIGHTH)
        pt = GRAY8 if (options & LUMA_ONLY and pixel_type < GRAY8) else pixel_type
        bpp = BYTES_PER_PIXEL[pt]
        if inf["subsample"] == 0 and pt == RGB8888:
            bpp = 4  # reference reports 32 bpp but writes 16-bit pixels
        cx = (inf["width"] + mw - 1) // mw
        cy = (inf["height"] + mh - 1) // mh
        if canvas_shape is None:
            rows = cy * (mh >> sh) + yoff
            cols = cx * (mw >> sh) + xoff
            # strips may overhang the padded width when iMCUCount does not divide cx: add slack
            cols += 2048
        else:
            rows, cols = canvas_shape
        canvas = np.zeros((rows, cols * bpp), dtype=np.uint8)
        maxlog = 1 << 16
        log = (C.c_int * (6 * maxlog))() if want_log else None
        n_calls = C.c_int(0)
        dma = C.c_int(0)
        err = C.c_int(0)
        croparr = (C.c_int * 4)(*crop) if crop is not None else None
        after = (C.c_int * 2)()
        rc = self.lib.ref_decode_cb2(data, len(data), pixel_type, options, max_mcus, xoff, yoff,
                                     croparr, canvas.ctypes.data_as(C.c_void_p), cols * bpp, rows,
                                     1 if used_only else 0, log, maxlog if want_log else 0, stop_after,
                                     C.byref(n_calls), C.byref(dma), C.byref(err), after)
        out_log = None
        if want_log:
            n = min(n_calls.value, maxlog)
            out_log = np.frombuffer(log, dtype=np.int32)[: 6 * n].reshape(n, 6).copy()
        return dict(rc=rc, canvas=canvas, n_calls=n_calls.value, dma_reuse=dma.value,
                    last_error=err.value, log=out_log, info=inf, bpp=bpp, scale_shift=sh,
                    size_after=(after[0], after[1]))

    def decode_frame(self, data: bytes, pixel_type=RGB8888, options=0):
        """Full frame (H>>s x W>>s pixels, tightly packed) assembled from the draw callbacks."""
        r = self.decode_cb(data, pixel_type, options, used_only=False)
        if r["rc"] != 1:
            return r["rc"], None
        inf, sh, bpp = r["info"], r["scale_shift"], r["bpp"]
        adj = (1 << sh) - 1
        w = (inf["width"] + adj) >> sh
        h = (inf["height"] + adj) >> sh
        return 1, np.ascontiguousarray(r["canvas"][:h, : w * bpp])

    def decode_fb(self, data: bytes, pixel_type=RGB8888, options=0, crop=None, fill=0):
        """Framebuffer-mode decode; returns (rc, flat buffer; pitch = W (or the cropped width) * bpp).  fill: what the buffer holds before."""
        inf = self.info(data)
        if not inf["ok"]:
            return -1, None
        mw, mh = mcu_dims(inf["subsample"])
        sh = scale_shift(options)
        pt = GRAY8 if (options & LUMA_ONLY and pixel_type < GRAY8) else pixel_type
        bpp = BYTES_PER_PIXEL[pt]
        cy = (inf["height"] + mh - 1) // mh
        rows = cy * mh + mh  # generous: reference overruns rows
        fb = np.full((rows + 8, inf["width"] * bpp + 64), fill, dtype=np.uint8).reshape(-1)
        err = C.c_int(0)
        croparr = (C.c_int * 4)(*crop) if crop is not None else None
        rc = self.lib.ref_decode_fb_crop(data, len(data), pixel_type, options, croparr,
                                         fb.ctypes.data_as(C.c_void_p), C.byref(err))
        self.last_error = err.value
        return rc, fb

    def run_script(self, data: bytes, ops, canvas_rows=1200, canvas_pitch=8192):
        """A sequence of calls on ONE object (ref_run_script in ref_shim.cpp: op, a, b, c, d per row); returns the recorded values."""
        arr = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 5))
        canvas = np.zeros((canvas_rows, canvas_pitch), dtype=np.uint8)
        out = np.zeros(4096, dtype=np.int32)
        self.lib.ref_run_script.restype = C.c_int
        n = self.lib.ref_run_script(data, len(data), arr.ctypes.data_as(C.c_void_p), arr.shape[0], canvas.ctypes.data_as(C.c_void_p),
                                    canvas_pitch, canvas_rows, out.ctypes.data_as(C.c_void_p), out.size)
        return out[: min(n, out.size)].tolist()

    def bench(self, datas, pixel_type=RGB8888, options=0, reps=1, threads=1):
        n = len(datas)
        arr = (C.c_char_p * n)(*datas)
        lens = (C.c_int * n)(*[len(d) for d in datas])
        px = C.c_longlong(0)
        fl = C.c_int(0)
        secs = self.lib.ref_bench(arr, lens, n, pixel_type, options, reps, threads,
                                  C.byref(px), C.byref(fl))
        return dict(seconds=secs, pixels=px.value, failures=fl.value)


def ref_available(simd=False) -> bool:
    name = "libjpegdec_ref_sse2.so" if simd else "libjpegdec_ref_scalar.so"
    return os.path.exists(os.path.join(HERE, "_ref", name))


def load_c_array_header(path: str) -> bytes:
    """Parse a `const uint8_t x[] = {0xff,0xd8,...};` C header into bytes (used only in the build
    container, where the reference's own test_images/*.h exist; never on the GPU box)."""
    txt = open(path, "r", errors="replace").read()
    body = txt[txt.index("{") + 1: txt.rindex("}")]
    body = re.sub(r"//[^\n]*", "", body)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    vals = re.findall(r"0[xX][0-9a-fA-F]+|\d+", body)
    return bytes(int(v, 0) & 0xFF for v in vals)


class OracleDecoder:
    """Our C restatement (oracle/jpegdec_oracle.c) -- the portable checker."""

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle liboracle.so`)")
        self.lib = C.CDLL(path)
        L = self.lib

        class Info(C.Structure):
            _fields_ = [("width", C.c_int), ("height", C.c_int), ("ncomp", C.c_int), ("subsample", C.c_int),
                        ("mode", C.c_int), ("restart_interval", C.c_int), ("quant_id", C.c_int * 4),
                        ("dc_id", C.c_int * 4), ("ac_id", C.c_int * 4), ("scan_offset", C.c_int),
                        ("error", C.c_int), ("scan_start", C.c_int), ("scan_end", C.c_int), ("approx", C.c_int)]

        self.Info = Info
        L.orc_get_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(Info)]
        L.orc_filter.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.orc_huff_tables.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_quant_tables.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        L.orc_entropy.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_entropy2.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_block_pixels.argtypes = [C.c_void_p, C.c_uint16, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_block_pixels.restype = None
        L.orc_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.POINTER(C.c_int)]
        L.orc_draw_plan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int]
        L.orc_range_limit.restype = C.c_uint8
        L.orc_range565.restype = C.c_uint16
        L.orc_gray565.restype = C.c_uint16

    def info(self, data: bytes):
        inf = self.Info()
        rc = self.lib.orc_get_info(data, len(data), C.byref(inf))
        d = {k: getattr(inf, k) for k in ("width", "height", "ncomp", "subsample", "mode",
                                           "restart_interval", "scan_offset", "error")}
        d["quant_id"] = list(inf.quant_id)
        d["dc_id"] = list(inf.dc_id)
        d["ac_id"] = list(inf.ac_id)
        d["ok"] = rc
        return d

    def filter(self, scan: bytes) -> bytes:
        out = np.zeros(len(scan) + 16, dtype=np.uint8)
        n = self.lib.orc_filter(scan, len(scan), out.ctypes.data_as(C.c_void_p))
        return out[:n].tobytes()

    def huff_tables(self, data: bytes):
        dc = np.zeros(2 * 1024, dtype=np.uint8)
        ac = np.zeros(2 * 2048, dtype=np.uint16)
        rc = self.lib.orc_huff_tables(data, len(data), dc.ctypes.data_as(C.c_void_p),
                                      ac.ctypes.data_as(C.c_void_p))
        return rc, dc, ac

    def quant_tables(self, data: bytes):
        q = np.zeros((4, 64), dtype=np.int16)
        rc = self.lib.orc_quant_tables(data, len(data), q.ctypes.data_as(C.c_void_p))
        return rc, q

    def canvas_geometry(self, data: bytes, pixel_type=RGB8888, options=0):
        inf = self.info(data)
        mw, mh = mcu_dims(inf["subsample"])
        if inf["mode"] == 0xC2:
            options |= SCALE_EIGHTH                     # a progressive file is decoded as a 1/8 thumbnail (jpeg.inl:4964-4966)
        sh = scale_shift(options)
        pt = GRAY8 if (options & LUMA_ONLY and pixel_type < GRAY8) else pixel_type
        bpp = BYTES_PER_PIXEL[pt]
        if inf["subsample"] == 0 and pt == RGB8888:
            bpp = 2  # reference writes 16-bit pixels for gray JPEG + RGB8888
        cx = (inf["width"] + mw - 1) // mw
        cy = (inf["height"] + mh - 1) // mh
        return inf, cx, cy, mw >> sh, mh >> sh, bpp, sh

    def entropy(self, data: bytes, options=0):
        inf, cx, cy, mw, mh, bpp, sh = self.canvas_geometry(data, RGB8888, 0)
        bpm = {0x00: 1, 0x11: 3, 0x22: 6, 0x12: 4, 0x21: 4}[inf["subsample"]]
        if inf["ncomp"] == 1:
            bpm = 1
        nb = cx * cy * bpm
        coefs = np.zeros((nb, 64), dtype=np.int16)
        flags = np.zeros(nb, dtype=np.uint16)
        state = np.zeros((cx * cy, 2), dtype=np.uint32)
        dcp = np.zeros((cx * cy, 3), dtype=np.int32)
        self.blk_state = np.zeros((nb, 2), dtype=np.uint32)
        self.blk_pred = np.zeros(nb, dtype=np.int32)
        self.blk_ac_state = np.zeros((nb, 2), dtype=np.uint32)      # the reader at every block's first AC symbol (the product's index format 2)
        n = self.lib.orc_entropy2(data, len(data), options, nb, coefs.ctypes.data_as(C.c_void_p),
                                  flags.ctypes.data_as(C.c_void_p), state.ctypes.data_as(C.c_void_p),
                                  dcp.ctypes.data_as(C.c_void_p), self.blk_state.ctypes.data_as(C.c_void_p),
                                  self.blk_pred.ctypes.data_as(C.c_void_p), self.blk_ac_state.ctypes.data_as(C.c_void_p))
        return n, coefs, flags, state, dcp

    def decode_canvas(self, data: bytes, pixel_type=RGB8888, options=0):
        """MCU-padded canvas (cy*mh rows x cx*mw*bpp bytes)."""
        inf, cx, cy, mw, mh, bpp, sh = self.canvas_geometry(data, pixel_type, options)
        canvas = np.zeros((cy * mh, cx * mw * bpp), dtype=np.uint8)
        err = C.c_int(0)
        rc = self.lib.orc_decode(data, len(data), pixel_type, options, canvas.ctypes.data_as(C.c_void_p),
                                 canvas.shape[1], canvas.shape[0], C.byref(err))
        return rc, canvas, err.value

    def decode_frame(self, data: bytes, pixel_type=RGB8888, options=0):
        inf, cx, cy, mw, mh, bpp, sh = self.canvas_geometry(data, pixel_type, options)
        rc, canvas, err = self.decode_canvas(data, pixel_type, options)
        if rc != 1:
            return rc, None
        adj = (1 << sh) - 1
        w = (inf["width"] + adj) >> sh
        h = (inf["height"] + adj) >> sh
        return 1, np.ascontiguousarray(canvas[:h, : w * bpp])

    def draw_plan(self, data: bytes, pixel_type=RGB8888, options=0, max_mcus=0, uses_dma=False):
        rects = np.zeros((1 << 16, 6), dtype=np.int32)
        n = self.lib.orc_draw_plan(data, len(data), pixel_type, options, max_mcus, 1 if uses_dma else 0,
                                   rects.ctypes.data_as(C.c_void_p), rects.shape[0])
        return rects[: max(n, 0)].copy()


def oracle_available() -> bool:
    return os.path.exists(os.path.join(HERE, "liboracle.so"))
