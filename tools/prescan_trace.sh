#!/bin/bash
# phase times of the chunk-parallel host pre-scan (jda_frontend.cpp built with -DJDA_PRESCAN_TRACE), CPU only: tools/prescan_trace.sh
g++ -O2 -std=c++17 -fPIC -shared -fwrapv -DJDA_PRESCAN_TRACE -DJDA_LAB -Iinclude -pthread -o /tmp/libfront_trace.so jpegdec_amd/csrc/jda_frontend.cpp || exit 1
python - <<'PY'
import ctypes as C, sys, time
sys.path.insert(0, '.')
from bench import cached_jpeg
lib = C.CDLL("/tmp/libfront_trace.so")
lib.jda_prepare_ex.restype = C.c_void_p; lib.jda_prepare_ex.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
lib.jda_image_free.argtypes = [C.c_void_p]
for (w, h) in ((640, 480), (1280, 720)):
    j = cached_jpeg(w, h, "4:2:0", 1234)
    for flags in (8, 16):                 # JDA_PREPARE_SERIAL_PRESCAN, JDA_PREPARE_PARALLEL_PRESCAN
        for k in range(6):
            e = C.c_int32(0)
            t0 = time.perf_counter(); p = lib.jda_prepare_ex(j, len(j), flags, C.byref(e)); t1 = time.perf_counter(); lib.jda_image_free(p)
            sys.stderr.write("  %dx%d (%d KB) flags %d prepare %.1f us\n" % (w, h, len(j) >> 10, flags, (t1 - t0) * 1e6))
PY
