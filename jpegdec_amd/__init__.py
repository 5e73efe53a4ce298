"""jpegdec_amd -- MI355X-native baseline-JPEG decode path behind bitbank2/JPEGDEC's API.

The product is the C-ABI shared library ``jpegdec_amd/libjpegdec_amd.so`` (host front end in
C++, hand-written gfx950 HIP kernels; see include/jpegdec_amd.h).  This Python package is only a
thin ctypes binding used by the tests and by bench.py -- there is no Python or CPU decode path:
every decode call fails loudly when the HIP library or a GPU is missing.
"""
from .binding import (  # noqa: F401
    GRAY8,
    LUMA_ONLY,
    PREPARE_CONT_ALWAYS,
    PREPARE_CONT_NEVER,
    PREPARE_DEVICE_PRESCAN,
    PREPARE_PARALLEL_PRESCAN,
    PREPARE_SERIAL_PRESCAN,
    RGB565_BE,
    RGB565_LE,
    RGB8888,
    SCALE_EIGHTH,
    SCALE_HALF,
    SCALE_QUARTER,
    Batch,
    Context,
    DeviceImage,
    JdaError,
    Pipeline,
    PinnedFiles,
    SUBMIT_PINNED_INPUT,
    PreparedImage,
    crop_round,
    decode_resident,
    decode_to_host,
    kernel_launch_counts,
    draw_plan,
    draw_plan_ex,
    filter_on_device,
    index_equivalent,
    library_path,
    load_library,
    output_geometry,
    parse,
    prepare_batch,
    surface_checksum_host,
    upload_batch,
)
