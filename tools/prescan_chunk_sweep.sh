#!/bin/bash
# the chunk-parallel host pre-scan over chunk sizes (trace build of jda_frontend.cpp, CPU only): tools/prescan_chunk_sweep.sh
g++ -O2 -std=c++17 -fPIC -shared -fwrapv -DJDA_PRESCAN_TRACE -DJDA_LAB -Iinclude -pthread -o /tmp/libfront_trace.so jpegdec_amd/csrc/jda_frontend.cpp || exit 1
for cb in 0 2048 3072 4096 6144 9024; do
JDA_TRACE_CHUNK_BYTES=$cb python - <<'PY'
import ctypes as C, sys, time, os
sys.path.insert(0, '.')
from bench import cached_jpeg
lib = C.CDLL("/tmp/libfront_trace.so")
lib.jda_prepare_ex.restype = C.c_void_p; lib.jda_prepare_ex.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
lib.jda_image_free.argtypes = [C.c_void_p]
cb = int(os.environ["JDA_TRACE_CHUNK_BYTES"])
if cb == 0: del os.environ["JDA_TRACE_CHUNK_BYTES"]
for (w, h) in ((640, 480), (1280, 720), (1920, 1080)):
    j = cached_jpeg(w, h, "4:2:0", 1234)
    for flags in ((8, 16) if cb == 0 else (16,)):
        ts = []
        for k in range(30):
            e = C.c_int32(0)
            t0 = time.perf_counter(); p = lib.jda_prepare_ex(j, len(j), flags, C.byref(e)); t1 = time.perf_counter(); lib.jda_image_free(p)
            ts.append((t1 - t0) * 1e6)
        ts.sort()
        print("RESULT chunk %5d  %dx%d (%d KB) flags %2d: prepare median %.1f us, best %.1f" % (cb, w, h, len(j) >> 10, flags, ts[len(ts) // 2], ts[0]), flush=True)
PY
done 2>/dev/null | grep RESULT
