// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" wrapper around the UNMODIFIED reference decoder, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile; outputs go to
// oracle/_ref/ which is git-ignored).  Two builds are made from this one file:
//   -DNO_SIMD  -> libjpegdec_ref_scalar.so : the scalar integer path = the PARITY oracle
//   (default)  -> libjpegdec_ref_sse2.so   : the SSE2 path = the CPU TIMING baseline
// No reference source is copied into this repository: the reference translation unit is
// pulled in by #include at build time through -I/root/reference/src.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these.
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <pthread.h>

#ifdef SHIM_PRODUCT
// Third build of the same shim: against the PRODUCT's drop-in class (include/JPEGDEC.h, linked
// with libjpegdec_amd.so) so the very same driver code exercises reference and product.
#include "JPEGDEC.h"
#else
#include "JPEGDEC.cpp"   // reference src/JPEGDEC.cpp (which #includes jpeg.inl), via -I
#endif

namespace {

struct Canvas {
    uint8_t *pix;        // destination canvas
    int pitch_bytes;     // canvas pitch in bytes
    int rows;            // canvas rows available
    int cols_bytes;      // canvas usable bytes per row
    int used_only;       // copy iWidthUsed instead of iWidth
    int *log;            // draw-call log: x,y,iWidth,iHeight,iWidthUsed,iBpp per call
    int max_log;
    int n_calls;
    int stop_after;      // return 0 from the callback after this many calls (<=0: never)
    uint16_t *last_ptr;  // for the DMA ping-pong check
    int dma_reuse;       // number of consecutive callbacks that reused pPixels
};

int draw_to_canvas(JPEGDRAW *d)
{
    Canvas *c = (Canvas *)d->pUser;
    if (c->log && c->n_calls < c->max_log) {
        int *e = &c->log[c->n_calls * 6];
        e[0] = d->x; e[1] = d->y; e[2] = d->iWidth; e[3] = d->iHeight;
        e[4] = d->iWidthUsed; e[5] = d->iBpp;
    }
    if (d->pPixels == c->last_ptr) c->dma_reuse++;
    c->last_ptr = d->pPixels;
    c->n_calls++;
    if (c->pix) {
        const int bpp = d->iBpp;                 // 8, 16 or 32 (dither modes unused here)
        const int w = c->used_only ? d->iWidthUsed : d->iWidth;
        const int src_pitch = (d->iWidth * bpp) / 8;
        const uint8_t *src = (const uint8_t *)d->pPixels;
        for (int r = 0; r < d->iHeight; r++) {
            int y = d->y + r;
            if (y < 0 || y >= c->rows) continue;
            int xb = (d->x * bpp) / 8;
            int nb = (w * bpp) / 8;
            if (xb >= c->cols_bytes) continue;
            if (xb + nb > c->cols_bytes) nb = c->cols_bytes - xb;
            memcpy(c->pix + (size_t)y * c->pitch_bytes + xb, src + (size_t)r * src_pitch, nb);
        }
    }
    if (c->stop_after > 0 && c->n_calls >= c->stop_after) return 0;
    return 1;
}

int draw_nop(JPEGDRAW *) { return 1; }

double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

} // namespace

extern "C" {

// 1 = SSE2 build, 0 = scalar (-DNO_SIMD) build
int ref_is_simd(void)
{
#ifdef HAS_SSE
    return 1;
#else
    return 0;
#endif
}

#ifndef SHIM_PRODUCT
int ref_sizeof_state(void) { return (int)sizeof(JPEGIMAGE); }
#endif

// info[0..9] = width,height,subsample,bpp,jpegtype,orientation,hasthumb,thumbw,thumbh,lasterror
int ref_get_info(const uint8_t *data, int len, int *info)
{
    JPEGDEC *j = new JPEGDEC();
    int rc = j->openFLASH(data, len, draw_nop);
    info[0] = j->getWidth(); info[1] = j->getHeight(); info[2] = j->getSubSample();
    info[3] = j->getBpp(); info[4] = j->getJPEGType(); info[5] = j->getOrientation();
    info[6] = j->hasThumb(); info[7] = j->getThumbWidth(); info[8] = j->getThumbHeight();
    info[9] = j->getLastError();
    j->close();
    delete j;
    return rc;
}

// Callback-mode decode assembled into a caller canvas.  Returns decode()'s return value
// (or -1 if open failed).  crop[4] may be NULL (x,y,w,h passed to setCropArea).
// after[0..1] (may be NULL) = getWidth(), getHeight() AFTER decode(): with JPEG_EXIF_THUMBNAIL the object then describes
// the embedded thumbnail (reference test 10, MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp:236-260)
int ref_decode_cb2(const uint8_t *data, int len, int pixel_type, int options, int max_mcus,
                   int xoff, int yoff, const int *crop,
                   uint8_t *canvas, int pitch_bytes, int rows, int used_only,
                   int *log, int max_log, int stop_after,
                   int *n_calls, int *dma_reuse, int *last_error, int *after);
int ref_decode_cb(const uint8_t *data, int len, int pixel_type, int options, int max_mcus,
                  int xoff, int yoff, const int *crop,
                  uint8_t *canvas, int pitch_bytes, int rows, int used_only,
                  int *log, int max_log, int stop_after,
                  int *n_calls, int *dma_reuse, int *last_error)
{
    return ref_decode_cb2(data, len, pixel_type, options, max_mcus, xoff, yoff, crop, canvas, pitch_bytes, rows, used_only,
                          log, max_log, stop_after, n_calls, dma_reuse, last_error, NULL);
}
int ref_decode_cb2(const uint8_t *data, int len, int pixel_type, int options, int max_mcus,
                   int xoff, int yoff, const int *crop,
                   uint8_t *canvas, int pitch_bytes, int rows, int used_only,
                   int *log, int max_log, int stop_after,
                   int *n_calls, int *dma_reuse, int *last_error, int *after)
{
    JPEGDEC *j = new JPEGDEC();
    Canvas c;
    memset(&c, 0, sizeof(c));
    c.pix = canvas; c.pitch_bytes = pitch_bytes; c.rows = rows; c.cols_bytes = pitch_bytes;
    c.used_only = used_only; c.log = log; c.max_log = max_log; c.stop_after = stop_after;
    int rc = j->openFLASH(data, len, draw_to_canvas);
    if (!rc) {
        if (last_error) *last_error = j->getLastError();
        delete j;
        return -1;
    }
    j->setPixelType(pixel_type);
    j->setUserPointer(&c);
    if (max_mcus > 0) j->setMaxOutputSize(max_mcus);
    if (crop) j->setCropArea(crop[0], crop[1], crop[2], crop[3]);
    rc = j->decode(xoff, yoff, options);
    if (n_calls) *n_calls = c.n_calls;
    if (dma_reuse) *dma_reuse = c.dma_reuse;
    if (last_error) *last_error = j->getLastError();
    if (after) { after[0] = j->getWidth(); after[1] = j->getHeight(); }
    j->close();
    delete j;
    return rc;
}

// A SEQUENCE of calls on ONE object (what state does a decode leave behind for the next one?).  ops: op code + 4 arguments each;
// out: the values every op records, in order; returns how many were written.
//   1 setPixelType(a)            -> rc? (void in the class: nothing recorded), then getLastError is NOT touched
//   2 setMaxOutputSize(a)
//   3 setCropArea(a, b, c, d)    -> getCropArea's x, y, w, h
//   4 decode(a, b, options = c)  -> rc, getLastError, draw calls, FNV-1a of the draw log, FNV-1a of the canvas (iWidthUsed pixels of
//                                   every strip; the canvas is cleared first).  ONE decode per open: the reference does not rewind its
//                                   file position, a second decode() fails or decodes from the wrong place
//   5 getters                    -> width, height, bpp, subsample, jpegtype, orientation, hasThumb, thumb w, thumb h, lastError
//   6 close + openFLASH again    -> open's rc
static uint32_t fnv1a(const void *p, size_t n, uint32_t h = 2166136261u)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 16777619u; }
    return h;
}
int ref_run_script(const uint8_t *data, int len, const int *ops, int n_ops, uint8_t *canvas, int pitch_bytes, int rows, int *out, int max_out)
{
    JPEGDEC *j = new JPEGDEC();
    Canvas c;
    static int log[6 * 65536];
    int n = 0;
#define PUT(v) do { if (n < max_out) out[n] = (int)(v); n++; } while (0)
    memset(&c, 0, sizeof(c));
    c.pix = canvas; c.pitch_bytes = pitch_bytes; c.rows = rows; c.cols_bytes = pitch_bytes; c.used_only = 1; c.log = log; c.max_log = 65536;
    int opened = j->openFLASH(data, len, draw_to_canvas);
    PUT(opened);
    if (opened) j->setUserPointer(&c);
    for (int i = 0; i < n_ops && opened; i++) {
        const int *o = ops + 5 * i;
        switch (o[0]) {
        case 1: j->setPixelType(o[1]); break;
        case 2: j->setMaxOutputSize(o[1]); break;
        case 3: { j->setCropArea(o[1], o[2], o[3], o[4]); int x, y, w, h; j->getCropArea(&x, &y, &w, &h); PUT(x); PUT(y); PUT(w); PUT(h); break; }
        case 4: {
            memset(canvas, 0, (size_t)pitch_bytes * rows);
            c.n_calls = 0; c.dma_reuse = 0; c.last_ptr = NULL;
            const int rc = j->decode(o[1], o[2], o[3]);
            PUT(rc); PUT(j->getLastError());
            // (a failed decode: how far the reference got before it noticed is not pinned)
            PUT(rc ? c.n_calls : 0);
            PUT(rc ? fnv1a(log, sizeof(int) * 6 * (size_t)(c.n_calls < 65536 ? c.n_calls : 65536)) : 0);
            PUT(rc ? fnv1a(canvas, (size_t)pitch_bytes * rows) : 0);
            break;
        }
        case 5:
            PUT(j->getWidth()); PUT(j->getHeight()); PUT(j->getBpp()); PUT(j->getSubSample()); PUT(j->getJPEGType()); PUT(j->getOrientation());
            PUT(j->hasThumb()); PUT(j->getThumbWidth()); PUT(j->getThumbHeight()); PUT(j->getLastError());
            break;
        case 6:
            j->close();
            opened = j->openFLASH(data, len, draw_to_canvas);
            PUT(opened);
            if (opened) j->setUserPointer(&c);
            break;
        default: break;
        }
    }
#undef PUT
    if (opened) j->close();
    delete j;
    return n;
}

// Framebuffer-mode decode (jpeg.inl:5114-5124): caller buffer, pitch = image width,
// rows rounded up to an MCU multiple by the caller.
int ref_decode_fb_crop(const uint8_t *data, int len, int pixel_type, int options, const int *crop,
                       void *fb, int *last_error);
int ref_decode_fb(const uint8_t *data, int len, int pixel_type, int options,
                  void *fb, int *last_error)
{
    return ref_decode_fb_crop(data, len, pixel_type, options, NULL, fb, last_error);
}

// the same with setCropArea(crop[0..3]) first: the buffer pitch becomes the cropped width
int ref_decode_fb_crop(const uint8_t *data, int len, int pixel_type, int options, const int *crop,
                       void *fb, int *last_error)
{
    JPEGDEC *j = new JPEGDEC();
    int rc = j->openFLASH(data, len, draw_nop);
    if (!rc) {
        if (last_error) *last_error = j->getLastError();
        delete j;
        return -1;
    }
    j->setPixelType(pixel_type);
    j->setFramebuffer(fb);
    if (crop) j->setCropArea(crop[0], crop[1], crop[2], crop[3]);
    rc = j->decode(0, 0, options);
    if (last_error) *last_error = j->getLastError();
    j->close();
    delete j;
    return rc;
}

// ---- CPU timing baseline: T threads, one JPEGDEC object per thread, images dealt
// round-robin, no-op draw callback (the reference's own perf convention,
// examples/jpeg_perf_test/jpeg_perf_test.ino:8-12).
struct BenchArg {
    const uint8_t *const *datas; const int *lens; int n_images; int tid; int n_threads;
    int pixel_type; int options; int reps; long long pixels; int failures;
};

static void *bench_thread(void *p)
{
    BenchArg *a = (BenchArg *)p;
    JPEGDEC *j = new JPEGDEC();
    for (int r = 0; r < a->reps; r++) {
        for (int i = a->tid; i < a->n_images; i += a->n_threads) {
            if (!j->openFLASH(a->datas[i], a->lens[i], draw_nop)) { a->failures++; continue; }
            j->setPixelType(a->pixel_type);
            if (!j->decode(0, 0, a->options)) a->failures++;
            a->pixels += (long long)j->getWidth() * j->getHeight();
            j->close();
        }
    }
    delete j;
    return NULL;
}

// Returns wall seconds; *pixels = total source pixels decoded; *failures = failed decodes.
double ref_bench(const uint8_t *const *datas, const int *lens, int n_images,
                 int pixel_type, int options, int reps, int n_threads,
                 long long *pixels, int *failures)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    BenchArg *args = (BenchArg *)calloc(n_threads, sizeof(BenchArg));
    double t0 = now_s();
    for (int t = 0; t < n_threads; t++) {
        args[t].datas = datas; args[t].lens = lens; args[t].n_images = n_images;
        args[t].tid = t; args[t].n_threads = n_threads; args[t].pixel_type = pixel_type;
        args[t].options = options; args[t].reps = reps;
        pthread_create(&th[t], NULL, bench_thread, &args[t]);
    }
    long long px = 0; int fl = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        px += args[t].pixels; fl += args[t].failures;
    }
    double t1 = now_s();
    if (pixels) *pixels = px;
    if (failures) *failures = fl;
    free(th); free(args);
    return t1 - t0;
}

} // extern "C"
