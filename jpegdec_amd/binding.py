"""ctypes binding of include/jpegdec_amd.h (the C-ABI).  Plumbing only."""
import ctypes as C
import os

import numpy as np

# constants = include/jpegdec_amd.h (= the reference's src/JPEGDEC.h values)
RGB565_LE, RGB565_BE, RGB8888, GRAY8 = 0, 1, 2, 3
SCALE_HALF, SCALE_QUARTER, SCALE_EIGHTH, LUMA_ONLY = 2, 4, 8, 64

ERROR_NAMES = {0: "JDA_SUCCESS", 1: "JDA_INVALID_PARAMETER", 2: "JDA_DECODE_ERROR",
               3: "JDA_UNSUPPORTED_FEATURE", 4: "JDA_INVALID_FILE", 5: "JDA_ERROR_MEMORY",
               6: "JDA_ERROR_NO_DEVICE", 7: "JDA_ERROR_HIP"}


class JdaError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s (%d) %s" % (ERROR_NAMES.get(code, "?"), code, what))


class ImageInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "ncomp", "subsample", "bpp", "jpeg_type", "restart_interval",
        "orientation", "mcu_w", "mcu_h", "mcus_x", "mcus_y", "scan_offset", "blocks_per_mcu",
        "has_thumb", "thumb_w", "thumb_h", "thumb_offset", "scan_start", "scan_end", "approx")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Output(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("pitch_bytes", C.c_int32), ("width_px", C.c_int32),
                ("rows", C.c_int32)]


class PipelineStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("images", "device_images", "host_path_images", "failed_images", "source_pixels",
                                          "compressed_bytes", "h2d_bytes")] + [("launches", C.c_int32), ("spec_rounds_max", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("source_pixels", C.c_int64), ("output_bytes", C.c_int64), ("scan_bytes", C.c_int64),
                ("index_bytes", C.c_int64), ("table_bytes", C.c_int64), ("n_launches", C.c_int32),
                ("n_workgroups", C.c_int32), ("tiles", C.c_int64), ("tiles_whole_images", C.c_int64)]


def library_path() -> str:
    # JDA_LIBRARY: another build of the same library (kernel A/B runs on the GPU box, tools/gpu_ab.sh)
    return os.environ.get("JDA_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjpegdec_amd.so")


_lib = None

# every symbol include/jpegdec_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_PROTOTYPES = [
    ("jda_parse", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(ImageInfo)]),
    ("jda_prepare", _P, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_prepare_ex", _P, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_set_host_prescan_helpers", C.c_int, [C.c_int32]),
    ("jda_image_prescan_pending", C.c_int, [_P]),
    ("jda_prepare_batch", C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("jda_dev_image_prescan_on_device", C.c_int, [_P]),
    ("jda_last_prescan_rounds", C.c_int, [_P]),
    ("jda_filter_on_device", C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    ("jda_effective_options", C.c_int32, [_P, C.c_int32]),
    ("jda_dev_image_read_index", C.c_int, [_P, _P, _P, _P]),
    ("jda_dev_image_mcus_ok", C.c_uint32, [_P]),
    ("jda_upload_batch", C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(_P)]),
    ("jda_image_free", None, [_P]),
    ("jda_image_get_info", C.POINTER(ImageInfo), [_P]),
    ("jda_image_scan", _P, [_P, C.POINTER(C.c_uint32)]),
    ("jda_image_block_index", _P, [_P, C.POINTER(C.c_uint32)]),
    ("jda_image_block_dc", _P, [_P]),
    ("jda_index_equivalent", C.c_int, [_P, _P, C.c_uint32]),
    ("jda_image_block_cont", _P, [_P, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    ("jda_kernel_launch_counts", C.c_int, [C.c_char_p, C.c_int]),
    ("jda_image_tables", _P, [_P, C.POINTER(C.c_uint32)]),
    ("jda_image_truncation_events", C.c_uint32, [_P]),
    ("jda_image_general_p1", C.c_uint32, [_P]),
    ("jda_output_geometry", C.c_int, [C.POINTER(ImageInfo), C.c_int32, C.c_int32] + [C.POINTER(C.c_int32)] * 5),
    ("jda_draw_plan", C.c_int, [C.POINTER(ImageInfo), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    ("jda_crop_round", None, [C.POINTER(ImageInfo)] + [C.POINTER(C.c_int32)] * 4),
    ("jda_draw_plan_ex", C.c_int, [C.POINTER(ImageInfo), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32]),
    ("jda_draw_plan_at", C.c_int, [C.POINTER(ImageInfo), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32]),
    ("jda_device_count", C.c_int, []),
    ("jda_create", _P, [C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_destroy", None, [_P]),
    ("jda_last_hip_error", C.c_char_p, [_P]),
    ("jda_stream", _P, [_P]),
    ("jda_malloc", _P, [_P, C.c_size_t]),
    ("jda_free", None, [_P, _P]),
    ("jda_memset", C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    ("jda_copy_to_host", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("jda_copy_to_device", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("jda_upload", _P, [_P, _P, C.POINTER(C.c_int32)]),
    ("jda_dev_image_free", None, [_P, _P]),
    ("jda_dev_image_bytes", C.c_size_t, [_P]),
    ("jda_batch_create", _P, [_P, C.c_int32, C.POINTER(_P), C.POINTER(Output), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("jda_batch_destroy", None, [_P, _P]),
    ("jda_batch_decode", C.c_int, [_P, _P]),
    ("jda_batch_get_stats", C.c_int, [_P, C.POINTER(BatchStats)]),
    ("jda_sync", C.c_int, [_P]),
    ("jda_timer_start", C.c_int, [_P]),
    ("jda_timer_stop", C.c_int, [_P]),
    ("jda_timer_elapsed_ms", C.c_double, [_P]),
    ("jda_decode_to_host", C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32]),
    ("jda_decode_to_host_ex", C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_pipeline_create", _P, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_pipeline_destroy", None, [_P]),
    ("jda_pipeline_submit_ex", C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(Output), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_pipeline_submit", C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(Output), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("jda_pipeline_wait", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    ("jda_pipeline_get_stats", C.c_int, [_P, C.POINTER(PipelineStats)]),
    ("jda_pipeline_read_index", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_uint32)]),
    ("jda_checksum_surfaces", C.c_int, [_P, C.c_int32, C.POINTER(Output), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    ("jda_device_pci_bus_id", C.c_int, [_P, C.c_char_p, C.c_int32]),
    ("jda_upload_batch_ex", C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("jda_batch_get_status", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("jda_decode_to_host_rect", C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _P, C.c_int32, C.c_int32,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("jda_batch_create_rect", _P, [_P, C.c_int32, C.POINTER(_P), C.POINTER(Output), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("jda_version", C.c_char_p, []),
    ("jda_host_alloc", _P, [C.c_size_t]),
    ("jda_host_free", None, [_P]),
    ("jda_host_register", C.c_int, [_P, C.c_size_t]),
    ("jda_host_unregister", C.c_int, [_P]),
]


def load_library():
    """Load libjpegdec_amd.so (built by `make lib` / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: build it with `make lib` (hipcc --offload-arch=gfx950); "
                                "jpegdec_amd has no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, res, args in _PROTOTYPES:
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def parse(jpeg: bytes) -> dict:
    info = ImageInfo()
    rc = load_library().jda_parse(jpeg, len(jpeg), C.byref(info))
    d = info.as_dict()
    d["status"] = rc
    return d


def output_geometry(info: ImageInfo, pixel_type=RGB8888, options=0):
    vals = [C.c_int32(0) for _ in range(5)]
    rc = load_library().jda_output_geometry(C.byref(info), pixel_type, options, *[C.byref(v) for v in vals])
    if rc != 0:
        raise JdaError(rc, "jda_output_geometry")
    return dict(zip(("bpp", "out_w", "out_h", "canvas_w", "canvas_h"), [v.value for v in vals]))


def draw_plan(info: ImageInfo, pixel_type=RGB8888, options=0, max_mcus=0, uses_dma=False):
    rects = np.zeros((1 << 16, 6), dtype=np.int32)
    n = load_library().jda_draw_plan(C.byref(info), pixel_type, options, max_mcus, 1 if uses_dma else 0,
                                     rects.ctypes.data_as(_P), rects.shape[0])
    return rects[: max(n, 0)].copy()


def crop_round(info: ImageInfo, x, y, w, h):
    v = [C.c_int32(a) for a in (x, y, w, h)]
    load_library().jda_crop_round(C.byref(info), *[C.byref(a) for a in v])
    return tuple(a.value for a in v)


def draw_plan_ex(info: ImageInfo, pixel_type=RGB8888, options=0, max_mcus=0, uses_dma=False, crop=None):
    rects = np.zeros((1 << 16, 8), dtype=np.int32)
    carr = (C.c_int32 * 4)(*crop) if crop is not None else None
    n = load_library().jda_draw_plan_ex(C.byref(info), pixel_type, options, max_mcus, 1 if uses_dma else 0,
                                        carr, rects.ctypes.data_as(_P), rects.shape[0])
    return rects[: max(n, 0)].copy()


PREPARE_DEVICE_PRESCAN = 1
PREPARE_CONT_ALWAYS = 2
PREPARE_CONT_NEVER = 4
PREPARE_SERIAL_PRESCAN = 8
PREPARE_PARALLEL_PRESCAN = 16


class PreparedImage:
    """Host-side result of parse + LUT build + scan filter + serial pre-scan (jda_prepare).

    device_prescan=True (jda_prepare_ex, JDA_PREPARE_DEVICE_PRESCAN): for a stream with restart markers the
    serial pre-scan is left to the GPU (done by DeviceImage / jda_upload); `prescan_pending` tells."""

    def __init__(self, jpeg: bytes, device_prescan: bool = False, _handle=None, flags: int = 0):
        """flags: further JDA_PREPARE_* bits (PREPARE_CONT_ALWAYS / PREPARE_CONT_NEVER)"""
        self.lib = load_library()
        err = C.c_int32(0)
        self.handle = _handle if _handle else self.lib.jda_prepare_ex(jpeg, len(jpeg), (PREPARE_DEVICE_PRESCAN if device_prescan else 0) | flags, C.byref(err))
        if not self.handle:
            raise JdaError(err.value, "jda_prepare")
        self.info = self.lib.jda_image_get_info(self.handle).contents

    def close(self):
        if self.handle:
            self.lib.jda_image_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_mcus(self):
        return self.info.mcus_x * self.info.mcus_y

    @property
    def prescan_pending(self) -> bool:
        return bool(self.lib.jda_image_prescan_pending(self.handle))

    def scan(self) -> np.ndarray:
        n = C.c_uint32(0)
        p = self.lib.jda_image_scan(self.handle, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    @property
    def n_blocks(self):
        return self.n_mcus * self.info.blocks_per_mcu

    def block_index(self):
        """(index[n_blocks+1] uint32 = pos<<7|off per block, n_mcus_ok)"""
        n = C.c_uint32(0)
        p = self.lib.jda_image_block_index(self.handle, C.byref(n))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(self.n_blocks + 1,)).copy()
        return arr, n.value

    def block_cont(self):
        """(cont_first[n_blocks + 1], cont[n]) -- the serial pre-scan's continuation entries (jda_image_block_cont)"""
        cf, n = _P(), C.c_uint32(0)
        p = self.lib.jda_image_block_cont(self.handle, C.byref(cf), C.byref(n))
        first = np.ctypeslib.as_array(C.cast(cf, C.POINTER(C.c_uint32)), shape=(self.n_blocks + 1,)).copy()
        ent = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
        return first, ent

    def block_dc(self) -> np.ndarray:
        p = self.lib.jda_image_block_dc(self.handle)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(self.n_blocks,)).copy()

    def tables(self) -> np.ndarray:
        n = C.c_uint32(0)
        p = self.lib.jda_image_tables(self.handle, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    def truncation_events(self) -> int:
        return self.lib.jda_image_truncation_events(self.handle)

    def general_p1(self) -> bool:
        """An AC table codes EOB twice: the kernels cannot find EOB by one compare and take their general bit reader."""
        return bool(self.lib.jda_image_general_p1(self.handle))

    def geometry(self, pixel_type=RGB8888, options=0):
        return output_geometry(self.info, pixel_type, options)


class Context:
    """One per process per GPU: HIP device + stream + timing events (jda_create)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        err = C.c_int32(0)
        self.handle = self.lib.jda_create(device, C.byref(err))
        if not self.handle:
            raise JdaError(err.value, "jda_create(device=%d): no usable HIP device -- there is no CPU fallback" % device)
        self.device = device

    def close(self):
        if self.handle:
            self.lib.jda_destroy(self.handle)
            self.handle = None

    def check(self, rc, what=""):
        if rc != 0:
            raise JdaError(rc, what + ": " + (self.lib.jda_last_hip_error(self.handle) or b"").decode())

    def malloc(self, nbytes: int) -> int:
        p = self.lib.jda_malloc(self.handle, nbytes)
        if not p:
            raise JdaError(5, "jda_malloc(%d)" % nbytes)
        return p

    def free(self, ptr):
        self.lib.jda_free(self.handle, ptr)

    def memset(self, ptr, value, nbytes):
        self.check(self.lib.jda_memset(self.handle, ptr, value, nbytes), "jda_memset")

    def to_host(self, ptr, nbytes) -> np.ndarray:
        out = np.empty(nbytes, dtype=np.uint8)
        self.check(self.lib.jda_copy_to_host(self.handle, out.ctypes.data_as(_P), ptr, nbytes), "jda_copy_to_host")
        return out

    def sync(self):
        self.check(self.lib.jda_sync(self.handle), "jda_sync")

    def checksums(self, surfaces, row_bytes):
        """jda_checksum_surfaces: surfaces = list of (device_ptr, pitch_bytes, width_px, rows); one uint64 per surface."""
        n = len(surfaces)
        outs = (Output * n)(*[Output(*o) for o in surfaces])
        rb = (C.c_int32 * n)(*row_bytes)
        res = (C.c_uint64 * n)()
        self.check(self.lib.jda_checksum_surfaces(self.handle, n, outs, rb, res), "jda_checksum_surfaces")
        return [int(v) for v in res]

    def pci_bus_id(self) -> str:
        buf = C.create_string_buffer(32)
        self.check(self.lib.jda_device_pci_bus_id(self.handle, buf, 32), "jda_device_pci_bus_id")
        return buf.value.decode()

    def timer_start(self):
        self.check(self.lib.jda_timer_start(self.handle), "jda_timer_start")

    def timer_stop(self):
        self.check(self.lib.jda_timer_stop(self.handle), "jda_timer_stop")

    def timer_elapsed_ms(self) -> float:
        return self.lib.jda_timer_elapsed_ms(self.handle)


class DeviceImage:
    """Inputs of one image resident in HBM (jda_upload)."""

    def __init__(self, ctx: Context, prepared: PreparedImage, _handle=None):
        self.ctx = ctx
        err = C.c_int32(0)
        self.handle = _handle if _handle else ctx.lib.jda_upload(ctx.handle, prepared.handle, C.byref(err))
        if not self.handle:
            raise JdaError(err.value, "jda_upload")
        self.info = ImageInfo.from_buffer_copy(prepared.info)
        self.nbytes = ctx.lib.jda_dev_image_bytes(self.handle)
        self.prescan_on_device = bool(ctx.lib.jda_dev_image_prescan_on_device(self.handle))
        self.n_mcus_ok = int(ctx.lib.jda_dev_image_mcus_ok(self.handle))

    def read_index(self):
        """(index[n_blocks + 1] uint32, dc[n_blocks] int16) as they stand in HBM"""
        nb = self.info.mcus_x * self.info.mcus_y * self.info.blocks_per_mcu
        idx = np.zeros(nb + 1, np.uint32)
        dc = np.zeros(nb, np.int16)
        rc = self.ctx.lib.jda_dev_image_read_index(self.ctx.handle, self.handle, idx.ctypes.data_as(C.c_void_p), dc.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise JdaError(rc, "jda_dev_image_read_index")
        return idx, dc

    def close(self):
        if self.handle:
            self.ctx.lib.jda_dev_image_free(self.ctx.handle, self.handle)
            self.handle = None


def index_equivalent(a, b) -> bool:
    """Do two per-block indexes (format 2) name the same decode?  An entry is (byte position << 7) | flag << 6 | bit offset: the
    bit position pos * 8 + off of the block's FIRST AC SYMBOL is what P1 starts from; the split into (pos, off) -- the reference
    reader's phase behind the refill at the top of its AC loop -- only matters for a block flagged JDA_INDEX_TRUNC (its truncated
    magnitude reads are emulated from that phase).  The serial pre-scan writes the true phase everywhere, the device pre-scan a
    canonical one ((p >> 3) << 7 | p & 7) into unflagged entries: equivalent = same bit position and flag in every block's entry,
    same entry where flagged.  The closing entry (the last one) only bounds the scan from above: the device pre-scan's lies up to
    41 bits behind the serial one's (the DC symbol the stream's padding decodes to, an interval's rounding); either order of
    the arguments is accepted."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    if a.shape != b.shape or a.ndim != 1 or a.size == 0:
        return False
    return bool(load_library().jda_index_equivalent(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.size - 1))


def prepare_batch(jpegs, device_prescan: bool = False, threads: int = 0, strict: bool = True, flags: int = 0):
    """jda_prepare_batch: the images are prepared on `threads` host threads (0 = all).  strict=False: a rejected file leaves a
    None in the list (and its error code in the second return value) instead of failing the whole batch."""
    lib = load_library()
    n = len(jpegs)
    arr = (C.c_char_p * n)(*jpegs)
    lens = (C.c_int32 * n)(*[len(j) for j in jpegs])
    outs = (_P * n)()
    errs = (C.c_int32 * n)()
    rc = lib.jda_prepare_batch(n, arr, lens, (PREPARE_DEVICE_PRESCAN if device_prescan else 0) | flags, threads, outs, errs)
    res = [PreparedImage(jpegs[i], _handle=outs[i]) if outs[i] else None for i in range(n)]
    if not strict:
        return res, list(errs)
    if rc != 0:
        for r in res:
            if r is not None:
                r.close()
        raise JdaError(rc, "jda_prepare_batch")
    return res


def upload_batch(ctx: Context, prepared_list):
    """jda_upload_batch: all images in one go (pending block indexes are made on the GPU in two launches)."""
    n = len(prepared_list)
    himgs = (_P * n)(*[(p.handle if p is not None else None) for p in prepared_list])
    outs = (_P * n)()
    st = (C.c_int32 * n)()
    rc = ctx.lib.jda_upload_batch_ex(ctx.handle, n, himgs, outs, st)      # holes (None) stay holes
    if rc != 0:
        raise JdaError(rc, "jda_upload_batch")
    return [DeviceImage(ctx, prepared_list[i], _handle=outs[i]) if outs[i] else None for i in range(n)]


class Batch:
    """Launch plan over resident images (jda_batch_create / jda_batch_decode)."""

    def __init__(self, ctx: Context, images, outputs, pixel_types, options):
        """outputs: list of (device_ptr, pitch_bytes, width_px, rows)."""
        n = len(images)
        self.ctx = ctx
        self.n = n
        himgs = (_P * n)(*[(im.handle if im is not None else None) for im in images])
        outs = (Output * n)(*[Output(*o) for o in outputs])
        pts = (C.c_int32 * n)(*pixel_types)
        opts = (C.c_int32 * n)(*options)
        err = C.c_int32(0)
        self.handle = ctx.lib.jda_batch_create(ctx.handle, n, himgs, outs, pts, opts, C.byref(err))
        if not self.handle:
            raise JdaError(err.value, "jda_batch_create: " + (ctx.lib.jda_last_hip_error(ctx.handle) or b"").decode())
        st = BatchStats()
        ctx.lib.jda_batch_get_stats(self.handle, C.byref(st))
        self.stats = {k: getattr(st, k) for k, _ in BatchStats._fields_}

    def decode(self):
        self.ctx.check(self.ctx.lib.jda_batch_decode(self.ctx.handle, self.handle), "jda_batch_decode")

    def status(self):
        """per image: 0, 2 (JDA_DECODE_ERROR: bad MCU, the MCUs before it are decoded) or 1 (a hole in the image list)"""
        st = (C.c_int32 * self.n)()
        self.ctx.check(self.ctx.lib.jda_batch_get_status(self.handle, st), "jda_batch_get_status")
        return list(st)

    def close(self):
        if self.handle:
            self.ctx.lib.jda_batch_destroy(self.ctx.handle, self.handle)
            self.handle = None


class Pipeline:
    """The streamed pipeline (jda_pipeline_*): files in, pixels resident in HBM out; upload + filter + pre-scan of the next
    batch run under the decode of the current one."""

    def __init__(self, ctx: Context, max_images: int, depth: int = 2, host_threads: int = 0):
        self.ctx = ctx
        err = C.c_int32(0)
        self.handle = ctx.lib.jda_pipeline_create(ctx.handle, max_images, depth, host_threads, C.byref(err))
        if not self.handle:
            raise JdaError(err.value, "jda_pipeline_create")
        self._keep = {}

    @staticmethod
    def pack(jpegs, outputs, pixel_types, options):
        """The C arrays of one submit (a caller that streams the same list again and again builds them once: a C caller has them
        anyway, and per image they cost Python as much as the GPU takes for a 1280x720 file)."""
        n = len(jpegs)
        arr = (C.c_char_p * n)(*jpegs)
        lens = (C.c_int32 * n)(*[len(j) for j in jpegs])
        outs = (Output * n)(*[Output(*o) for o in outputs])
        pts = (C.c_int32 * n)(*pixel_types)
        opts = (C.c_int32 * n)(*options)
        return (jpegs, arr, lens, outs, pts, opts, n)

    @staticmethod
    def pack_pinned(pinned, picks, outputs, pixel_types, options):
        """The same for files that lie in page-locked memory (PinnedFiles): picks = indices into it.  Submitted with
        JDA_SUBMIT_PINNED_INPUT, the copy engine reads them where they are."""
        n = len(picks)
        arr = (C.c_void_p * n)(*[pinned.addrs[k] for k in picks])
        lens = (C.c_int32 * n)(*[pinned.lens[k] for k in picks])
        outs = (Output * n)(*[Output(*o) for o in outputs])
        pts = (C.c_int32 * n)(*pixel_types)
        opts = (C.c_int32 * n)(*options)
        return (pinned, C.cast(arr, C.POINTER(C.c_char_p)), lens, outs, pts, opts, n, arr)

    def submit_packed(self, packed, flags: int = 0) -> int:
        arr, lens, outs, pts, opts, n = packed[1:7]
        t = C.c_int32(-1)
        self.ctx.check(self.ctx.lib.jda_pipeline_submit_ex(self.handle, n, arr, lens, outs, pts, opts, flags, C.byref(t)), "jda_pipeline_submit_ex")
        self._keep[t.value] = packed                                      # the buffers stay alive until the batch is waited for
        return t.value

    def submit(self, jpegs, outputs, pixel_types, options) -> int:
        """outputs: list of (device_ptr, pitch_bytes, width_px, rows).  Returns the batch's ticket."""
        return self.submit_packed(self.pack(jpegs, outputs, pixel_types, options))

    def wait(self, ticket: int):
        """Blocks until the batch is decoded; returns the list of per-image status codes."""
        n = self._keep[ticket][6]
        st = (C.c_int32 * n)()
        self.ctx.check(self.ctx.lib.jda_pipeline_wait(self.handle, ticket, st), "jda_pipeline_wait")
        del self._keep[ticket]
        return list(st)

    def read_index(self, ticket: int, i: int, n_blocks: int):
        idx = np.zeros(n_blocks + 1, np.uint32)
        dc = np.zeros(n_blocks, np.int16)
        flen = C.c_uint32(0)
        rc = self.ctx.lib.jda_pipeline_read_index(self.handle, ticket, i, idx.ctypes.data_as(_P), dc.ctypes.data_as(_P), C.byref(flen))
        if rc != 0:
            raise JdaError(rc, "jda_pipeline_read_index")
        return idx, dc, flen.value

    @property
    def stats(self):
        st = PipelineStats()
        self.ctx.lib.jda_pipeline_get_stats(self.handle, C.byref(st))
        return {k: getattr(st, k) for k, _ in PipelineStats._fields_}

    def close(self):
        if self.handle:
            self.ctx.lib.jda_pipeline_destroy(self.handle)
            self.handle = None


class PinnedFiles:
    """Files copied once into page-locked host memory: what a loader that reads into page-locked memory hands to
    jda_pipeline_submit_ex(.., JDA_SUBMIT_PINNED_INPUT, ..).  Default: ONE arena (jda_host_alloc), each file at a 4 KB boundary.
    separate=True: one allocation of exactly the file's size per file, alternately jda_host_alloc and a malloc'ed buffer made
    known with jda_host_register (separate page-locked objects: no copy may run over the end of one or across two)."""

    def __init__(self, files, separate: bool = False):
        self.lib = load_library()
        self.lens = [len(f) for f in files]
        self.base, self._allocs, self._registered = None, [], []
        if separate:
            self.addrs = []
            for k, f in enumerate(files):
                if k & 1:
                    buf = C.create_string_buffer(len(f))
                    if self.lib.jda_host_register(C.addressof(buf), len(f)) != 0:
                        raise JdaError(5, "jda_host_register(%d)" % len(f))
                    self._registered.append(buf)
                    a = C.addressof(buf)
                else:
                    a = self.lib.jda_host_alloc(len(f))
                    if not a:
                        raise JdaError(5, "jda_host_alloc(%d)" % len(f))
                    self._allocs.append(a)
                C.memmove(a, f, len(f))
                self.addrs.append(a)
            self.bytes = sum(self.lens)
            return
        offs, total = [], 0
        for ln in self.lens:
            offs.append(total)
            total += (ln + 4095) & ~4095
        self.bytes = total
        self.base = self.lib.jda_host_alloc(max(total, 4096))
        if not self.base:
            raise JdaError(5, "jda_host_alloc(%d)" % total)
        self.addrs = [self.base + o for o in offs]
        for a, f in zip(self.addrs, files):
            C.memmove(a, f, len(f))

    def close(self):
        if self.base:
            self.lib.jda_host_free(self.base)
            self.base = None
        for a in self._allocs:
            self.lib.jda_host_free(a)
        for buf in self._registered:
            self.lib.jda_host_unregister(C.addressof(buf))
        self._allocs, self._registered = [], []


SUBMIT_PINNED_INPUT = 1


def surface_checksum_host(canvas: np.ndarray) -> int:
    """The same checksum as Context.checksums, of a host array (rows x row_bytes uint8): the reference value in tests."""
    rows, row_bytes = canvas.shape
    dpr = (row_bytes + 3) // 4
    pad = np.zeros((rows, dpr * 4), np.uint8)
    pad[:, :row_bytes] = canvas
    d = pad.view("<u4").reshape(-1).astype(np.uint64)
    i = np.arange(d.size, dtype=np.uint64)
    m = ((d ^ ((i * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF))) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        return int(np.sum(m * (np.uint64(2) * i + np.uint64(1)), dtype=np.uint64))


def decode_to_host(ctx: Context, jpeg: bytes, pixel_type=RGB8888, options=0, out=None):
    """Decode one image through the GPU path into an MCU-padded host canvas (rows x pitch bytes).
    out: a canvas of that shape from an earlier call, to decode into (a fresh 4096x4096 canvas costs a millisecond of page faults)."""
    info = ImageInfo()
    rc = ctx.lib.jda_parse(jpeg, len(jpeg), C.byref(info))
    if rc != 0:
        raise JdaError(rc, "jda_parse")
    g = output_geometry(info, pixel_type, options)
    shape = (g["canvas_h"], g["canvas_w"] * g["bpp"])
    canvas = out if out is not None and out.shape == shape and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] else np.zeros(shape, dtype=np.uint8)
    rc = ctx.lib.jda_decode_to_host(ctx.handle, jpeg, len(jpeg), pixel_type, options,
                                    canvas.ctypes.data_as(_P), canvas.shape[1], canvas.shape[0])
    return rc, canvas, g


def decode_resident(ctx: Context, prepared: PreparedImage, pixel_type=RGB8888, options=0):
    """One prepared image through jda_upload + jda_batch_create + jda_batch_decode into an MCU-padded host canvas:
    (status of the image, canvas, geometry) -- the path on which an image keeps what its jda_prepare_ex flags asked for."""
    g = prepared.geometry(pixel_type, options)
    pitch = (g["canvas_w"] * g["bpp"] + 15) & ~15
    dimg = DeviceImage(ctx, prepared)
    surf = ctx.malloc(pitch * g["canvas_h"])
    try:
        ctx.memset(surf, 0, pitch * g["canvas_h"])
        batch = Batch(ctx, [dimg], [(surf, pitch, g["canvas_w"], g["canvas_h"])], [pixel_type], [options])
        try:
            batch.decode()
            ctx.sync()
            st = batch.status()[0]
        finally:
            batch.close()
        canvas = ctx.to_host(surf, pitch * g["canvas_h"]).reshape(g["canvas_h"], pitch)[:, : g["canvas_w"] * g["bpp"]].copy()
    finally:
        ctx.free(surf)
        dimg.close()
    return st, canvas, g


def kernel_launch_counts() -> dict:
    """jda_kernel_launch_counts: {kernel symbol: launches} for every kernel this process has launched so far."""
    lib = load_library()
    need = lib.jda_kernel_launch_counts(None, 0)
    buf = C.create_string_buffer(need + 4096)
    lib.jda_kernel_launch_counts(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, _, n = line.rpartition(" ")
        out[name] = int(n)
    return out


def decode_to_host_rect(ctx: Context, jpeg: bytes, pixel_type, options, mcu_rect):
    """jda_decode_to_host_rect: (rc, canvas, geometry, (tiles launched, tiles of the whole image))"""
    info = ImageInfo()
    rc = ctx.lib.jda_parse(jpeg, len(jpeg), C.byref(info))
    if rc != 0:
        raise JdaError(rc, "jda_parse")
    g = output_geometry(info, pixel_type, options)
    canvas = np.zeros((g["canvas_h"], g["canvas_w"] * g["bpp"]), dtype=np.uint8)
    rect = (C.c_int32 * 4)(*mcu_rect) if mcu_rect is not None else None
    tiles = (C.c_int32 * 2)()
    nok = C.c_int32(0)
    rc = ctx.lib.jda_decode_to_host_rect(ctx.handle, jpeg, len(jpeg), pixel_type, options, rect, canvas.ctypes.data_as(_P), canvas.shape[1],
                                         canvas.shape[0], C.byref(nok), tiles)
    return rc, canvas, g, (tiles[0], tiles[1])


def filter_on_device(ctx: Context, raw: bytes, restart_cap: int = 1 << 16):
    """JPEGFilter on the GPU (jda_filter_on_device): (filtered bytes, restart positions incl. the leading 0)."""
    out = np.zeros(max(len(raw), 1), np.uint8)
    n, nr = C.c_int32(0), C.c_int32(0)
    rpos = np.zeros(restart_cap, np.uint32)
    rc = ctx.lib.jda_filter_on_device(ctx.handle, raw, len(raw), out.ctypes.data_as(_P), C.byref(n), rpos.ctypes.data_as(_P), restart_cap, C.byref(nr))
    if rc != 0:
        raise JdaError(rc, "jda_filter_on_device")
    return out[: n.value].tobytes(), rpos[: min(nr.value + 1, restart_cap)].copy(), nr.value
