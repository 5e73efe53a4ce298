/* The reference's performance test (examples/jpeg_perf_test/jpeg_perf_test.ino:8-53: open from memory, decode(0, 0, options) with a
 * draw callback that does nothing, close; timed per decode) as a C program on the C flavour of the API, against libjpegdec_amd.so.
 * Usage: perf_user file.jpg [pixel_type [options [iterations]]]   (pixel_type: RGB565_LITTLE_ENDIAN = 0 .. as include/JPEGDEC.h)
 * Prints one JSON line: the mean and the best time of one open + decode + close in microseconds, after untimed warm-up decodes
 * (the first decode of a thread creates its device context).  bench.py's `c1` leg runs it on test_images/tulips (BASELINE config 1)
 * beside the reference's own builds at one thread.  Exit code != 0 when a decode fails (no GPU: JPEG_decode returns 0). */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "JPEGDEC.h"

static long g_calls;
static int draw(JPEGDRAW *d) { (void)d; g_calls++; return 1; }     /* do nothing; continue the decode */

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 100;
    const int pt = argc > 2 ? atoi(argv[2]) : RGB565_LITTLE_ENDIAN, opt = argc > 3 ? atoi(argv[3]) : 0;
    int iters = argc > 4 ? atoi(argv[4]) : 2000;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 101;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *data = (uint8_t *)malloc((size_t)len);
    if (!data || fread(data, 1, (size_t)len, f) != (size_t)len) return 102;
    fclose(f);
    JPEGIMAGE jpg;
    int w = 0, h = 0;
    double total = 0, best = 1e30;
    const int warm = iters / 10 + 20;
    for (int i = -warm; i < iters; i++) {
        const double t0 = now_us();
        if (!JPEG_openRAM(&jpg, data, (int)len, draw)) { printf("{\"error\": \"open failed: %d\"}\n", JPEG_getLastError(&jpg)); return 103; }
        JPEG_setPixelType(&jpg, pt);
        if (!JPEG_decode(&jpg, 0, 0, opt)) { printf("{\"error\": \"decode failed: %d\"}\n", JPEG_getLastError(&jpg)); return 104; }
        w = JPEG_getWidth(&jpg); h = JPEG_getHeight(&jpg);
        JPEG_close(&jpg);
        const double dt = now_us() - t0;
        if (i >= 0) { total += dt; if (dt < best) best = dt; }
    }
    printf("{\"width\": %d, \"height\": %d, \"pixel_type\": %d, \"options\": %d, \"iterations\": %d, \"us_per_decode\": %.2f, \"best_us\": %.2f, "
           "\"draw_calls_per_decode\": %ld}\n", w, h, pt, opt, iters, total / iters, best, g_calls / (warm + iters));
    free(data);
    return 0;
}
