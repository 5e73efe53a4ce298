// JPEGDEC.cpp -- the reference's class API (src/JPEGDEC.cpp:38-273) on top of the jpegdec_amd
// C-ABI.  decode() (reference src/JPEGDEC.cpp:244-250 -> DecodeJPEG, jpeg.inl:4946) hands the
// whole image to the GPU path and then either fills the caller's framebuffer
// (jpeg.inl:5114-5124) or replays the exact JPEGDRAW sequence the reference would have issued
// (jpeg.inl:5062-5084, 5300-5336) from the decoded canvas.
//
// Host language: C++, as the reference.  No CPU decode fallback exists: without a usable HIP
// device decode() fails with JPEG_ERROR_NO_DEVICE.
#include "../../include/JPEGDEC.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/jpegdec_amd.h"

struct jpegdec_amd_state {
    std::vector<uint8_t> owned;       // file-sourced data (open(filename) / callbacks)
    const uint8_t *data;
    int size;
    jda_image_info info;
    int error;
    int pixel_type;
    int max_mcus;
    int options;
    int xoff, yoff;
    int crop_x, crop_y, crop_w, crop_h;
    void *user;
    void *framebuffer;
    JPEG_DRAW_CALLBACK *draw;
    JPEG_CLOSE_CALLBACK *close_cb;
    void *close_handle;
    bool opened;
    // strip handed to the draw callback; two halves alternate with JPEG_USES_DMA (jpeg.inl:5073-5076)
    alignas(16) uint16_t strip[MAX_BUFFERED_PIXELS + 8];
};

namespace {

std::mutex g_ctx_mutex;
jda_ctx *g_ctx = NULL;
int g_ctx_err = JDA_SUCCESS;

// one device context per process, created on first decode (device = $JPEGDEC_AMD_DEVICE or 0)
jda_ctx *shared_ctx(int *err)
{
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    if (!g_ctx && g_ctx_err == JDA_SUCCESS) {
        const char *e = getenv("JPEGDEC_AMD_DEVICE");
        int32_t rc = JDA_SUCCESS;
        g_ctx = jda_create(e ? atoi(e) : 0, &rc);
        g_ctx_err = g_ctx ? JDA_SUCCESS : rc;
    }
    if (err) *err = g_ctx_err;
    return g_ctx;
}

void reset(jpegdec_amd_state *s)
{
    s->owned.clear();
    s->data = NULL; s->size = 0;
    memset(&s->info, 0, sizeof(s->info));
    s->error = JPEG_SUCCESS;
    s->pixel_type = RGB565_LITTLE_ENDIAN;     // memset default of the reference (src/JPEGDEC.cpp:66)
    s->max_mcus = 1000;                        // src/JPEGDEC.cpp:75
    s->options = 0; s->xoff = s->yoff = 0;
    s->crop_x = s->crop_y = s->crop_w = s->crop_h = 0;
    s->user = NULL; s->framebuffer = NULL; s->draw = NULL;
    s->close_cb = NULL; s->close_handle = NULL;
    s->opened = false;
}

int finish_open(jpegdec_amd_state *s, JPEG_DRAW_CALLBACK *draw)
{
    s->draw = draw;
    int rc = jda_parse(s->data, s->size, &s->info);   // JPEGInit -> JPEGParseInfo (jpeg.inl:830-833)
    if (rc != JDA_SUCCESS) { s->error = rc; return 0; }
    s->crop_x = s->crop_y = 0;                         // jpeg.inl:1683-1685
    s->crop_w = s->info.width; s->crop_h = s->info.height;
    s->opened = true;
    return 1;
}

} // namespace

JPEGDEC::JPEGDEC() : _jpeg(new jpegdec_amd_state) { reset(_jpeg); }
JPEGDEC::~JPEGDEC() { delete _jpeg; }

int JPEGDEC::openRAM(uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw)
{
    reset(_jpeg);
    _jpeg->data = pData; _jpeg->size = iDataSize;     // pointer kept, not copied (src/JPEGDEC.cpp:73-74)
    return finish_open(_jpeg, pfnDraw);
}

int JPEGDEC::openFLASH(const uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw)
{
    reset(_jpeg);
    _jpeg->data = pData; _jpeg->size = iDataSize;
    return finish_open(_jpeg, pfnDraw);
}

// File sources: the reference pulls 2 KiB at a time through pfnRead while decoding
// (jpeg.inl:1544-1566); the GPU path needs the whole scan, so the file is read once here.
int JPEGDEC::open(void *fHandle, int iDataSize, JPEG_CLOSE_CALLBACK *pfnClose, JPEG_READ_CALLBACK *pfnRead,
                  JPEG_SEEK_CALLBACK *pfnSeek, JPEG_DRAW_CALLBACK *pfnDraw)
{
    (void)pfnSeek;
    reset(_jpeg);
    _jpeg->close_cb = pfnClose; _jpeg->close_handle = fHandle;
    if (!pfnRead || iDataSize <= 0) { _jpeg->error = JPEG_INVALID_PARAMETER; return 0; }
    _jpeg->owned.resize((size_t)iDataSize);
    JPEGFILE f;
    f.iPos = 0; f.iSize = iDataSize; f.pData = NULL; f.fHandle = fHandle;
    int got = 0;
    while (got < iDataSize) {
        int32_t n = (*pfnRead)(&f, _jpeg->owned.data() + got, iDataSize - got);
        if (n <= 0) break;
        got += n;
    }
    _jpeg->data = _jpeg->owned.data(); _jpeg->size = got;
    return finish_open(_jpeg, pfnDraw);
}

int JPEGDEC::open(const char *szFilename, JPEG_OPEN_CALLBACK *pfnOpen, JPEG_CLOSE_CALLBACK *pfnClose,
                  JPEG_READ_CALLBACK *pfnRead, JPEG_SEEK_CALLBACK *pfnSeek, JPEG_DRAW_CALLBACK *pfnDraw)
{
    reset(_jpeg);
    if (!pfnOpen) { _jpeg->error = JPEG_INVALID_PARAMETER; return 0; }
    int32_t size = 0;
    void *h = (*pfnOpen)(szFilename, &size);
    if (!h) return 0;                                  // src/JPEGDEC.cpp:166-168
    return open(h, size, pfnClose, pfnRead, pfnSeek, pfnDraw);
}

int JPEGDEC::open(const char *szFilename, JPEG_DRAW_CALLBACK *pfnDraw)
{
    reset(_jpeg);
    FILE *f = fopen(szFilename, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    _jpeg->owned.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(_jpeg->owned.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    _jpeg->data = _jpeg->owned.data(); _jpeg->size = (int)got;
    return finish_open(_jpeg, pfnDraw);
}

void JPEGDEC::close()
{
    if (_jpeg->close_cb) (*_jpeg->close_cb)(_jpeg->close_handle);   // src/JPEGDEC.cpp:232-236
    _jpeg->close_cb = NULL;
}

void JPEGDEC::setFramebuffer(void *p) { _jpeg->framebuffer = p; }
void JPEGDEC::setUserPointer(void *p) { _jpeg->user = p; }
int JPEGDEC::getOrientation() { return _jpeg->info.orientation; }
int JPEGDEC::getLastError() { return _jpeg->error; }
int JPEGDEC::getWidth() { return _jpeg->info.width; }
int JPEGDEC::getHeight() { return _jpeg->info.height; }
int JPEGDEC::getBpp() { return _jpeg->info.bpp; }
int JPEGDEC::getSubSample() { return _jpeg->info.subsample; }
int JPEGDEC::getJPEGType() { return _jpeg->info.jpeg_type ? JPEG_MODE_PROGRESSIVE : JPEG_MODE_BASELINE; }
int JPEGDEC::hasThumb() { return 0; }          // EXIF thumbnails: host-only metadata, not on this path
int JPEGDEC::getThumbWidth() { return 0; }
int JPEGDEC::getThumbHeight() { return 0; }
int JPEGDEC::getPixelType() { return _jpeg->pixel_type; }

void JPEGDEC::setPixelType(int iType)
{
    if (iType >= 0 && iType < INVALID_PIXEL_TYPE) _jpeg->pixel_type = iType;
    else _jpeg->error = JPEG_INVALID_PARAMETER;        // src/JPEGDEC.cpp:47-53
}

void JPEGDEC::setMaxOutputSize(int iMaxMCUs) { _jpeg->max_mcus = iMaxMCUs < 1 ? 1 : iMaxMCUs; }

// Crop rectangle rounding to MCU boundaries (jpeg.inl:682-727).  Stored and reported; decode()
// of a cropped area is a "next" row and is refused rather than approximated.
void JPEGDEC::setCropArea(int x, int y, int w, int h)
{
    const int mw = _jpeg->info.mcu_w ? _jpeg->info.mcu_w : 8, mh = _jpeg->info.mcu_h ? _jpeg->info.mcu_h : 8;
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (w & (mw - 1)) w = (w & ~(mw - 1)) + mw;
    if (h & (mh - 1)) h = (h & ~(mh - 1)) + mh;
    if (x > _jpeg->info.width - mw) x = _jpeg->info.width - mw;
    if (y > _jpeg->info.height - mh) y = _jpeg->info.height - mh;
    if (x + w > _jpeg->info.width) w = _jpeg->info.width - mw;
    if (y + h > _jpeg->info.height) h = _jpeg->info.height - mh;
    x &= ~(mw - 1);
    y &= ~(mh - 1);
    _jpeg->crop_x = x; _jpeg->crop_y = y; _jpeg->crop_w = w; _jpeg->crop_h = h;
}

void JPEGDEC::getCropArea(int *x, int *y, int *w, int *h)
{
    *x = _jpeg->crop_x; *y = _jpeg->crop_y; *w = _jpeg->crop_w; *h = _jpeg->crop_h;
}

int JPEGDEC::decodeDither(uint8_t *, int) { _jpeg->error = JPEG_UNSUPPORTED_FEATURE; return 0; }   // JPEGDither is off this path
int JPEGDEC::decodeDither(int, int, uint8_t *, int) { _jpeg->error = JPEG_UNSUPPORTED_FEATURE; return 0; }

int JPEGDEC::decode(int x, int y, int iOptions)
{
    jpegdec_amd_state *s = _jpeg;
    s->xoff = x; s->yoff = y; s->options = iOptions;
    if (!s->opened) { s->error = JPEG_INVALID_PARAMETER; return 0; }
    if (s->pixel_type > EIGHT_BIT_GRAYSCALE || (iOptions & JPEG_EXIF_THUMBNAIL)) { s->error = JPEG_UNSUPPORTED_FEATURE; return 0; }
    if (s->crop_x != 0 || s->crop_y != 0 || s->crop_w != s->info.width || s->crop_h != s->info.height) {
        s->error = JPEG_UNSUPPORTED_FEATURE; return 0;
    }
    int pt = s->pixel_type;
    if ((iOptions & JPEG_LUMA_ONLY) && pt < EIGHT_BIT_GRAYSCALE) pt = s->pixel_type = EIGHT_BIT_GRAYSCALE;   // jpeg.inl:4991-4993
    int bpp, ow, oh, cw, ch;
    int rc = jda_output_geometry(&s->info, pt, iOptions, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) { s->error = s->info.mcu_w ? rc : JPEG_UNSUPPORTED_FEATURE; return 0; }
    int cerr = JDA_SUCCESS;
    jda_ctx *ctx = shared_ctx(&cerr);
    if (!ctx) { s->error = cerr; return 0; }

    std::vector<uint8_t> canvas((size_t)cw * ch * bpp);
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        rc = jda_decode_to_host(ctx, s->data, s->size, pt, iOptions, canvas.data(), cw * bpp, ch);
    }
    const bool partial = rc == JDA_DECODE_ERROR;      // the reference still delivers the MCUs before the bad one
    if (rc != JDA_SUCCESS && !partial) { s->error = rc; return 0; }

    if (s->framebuffer) {                              // jpeg.inl:5114-5124: pitch = image width, no callbacks
        const int fb_px = s->info.width;               // iCropCX, also for scaled output
        const int copy_px = cw < fb_px ? cw : fb_px;
        for (int r = 0; r < ch; r++)
            memcpy((uint8_t *)s->framebuffer + (size_t)r * fb_px * bpp, canvas.data() + (size_t)r * cw * bpp, (size_t)copy_px * bpp);
    } else if (s->draw) {
        std::vector<int32_t> rects(6 * 65536);
        int n = jda_draw_plan(&s->info, pt, iOptions, s->max_mcus, (iOptions & JPEG_USES_DMA) ? 1 : 0, rects.data(), 65536);
        const int shift = (iOptions & JPEG_SCALE_HALF) ? 1 : (iOptions & JPEG_SCALE_QUARTER) ? 2 : (iOptions & JPEG_SCALE_EIGHTH) ? 3 : 0;
        const int mh = s->info.mcu_h >> shift;
        // with JPEG_USES_DMA (and no user cap on the MCU count) the strip ping-pongs between the two halves
        bool dma = (iOptions & JPEG_USES_DMA) != 0;
        {   // the halves only alternate when the user cap did not win (jpeg.inl:5071-5076)
            int per_call = MAX_BUFFERED_PIXELS / ((s->info.mcu_w >> shift) * mh);
            if (pt == RGB8888) per_call /= 2;
            if (pt == EIGHT_BIT_GRAYSCALE) per_call *= 2;
            if (per_call > s->info.mcus_x) per_call = s->info.mcus_x;
            if (per_call > s->max_mcus) dma = false;
        }
        int half = 0;
        if (n > 65536) n = 65536;
        for (int i = 0; i < n; i++) {
            const int32_t *r = &rects[(size_t)6 * i];
            uint16_t *buf = s->strip + (dma ? half * (MAX_BUFFERED_PIXELS / 2) : 0);
            const int row_bytes = r[2] * bpp;
            for (int rr = 0; rr < mh; rr++) {
                const int cy_ = r[1] + rr;
                uint8_t *dst = (uint8_t *)buf + (size_t)rr * row_bytes;
                if (cy_ >= ch) { memset(dst, 0, (size_t)row_bytes); continue; }
                int avail = (cw - r[0]) * bpp;
                if (avail > row_bytes) avail = row_bytes;
                if (avail < 0) avail = 0;
                memcpy(dst, canvas.data() + ((size_t)cy_ * cw + r[0]) * bpp, (size_t)avail);
                if (avail < row_bytes) memset(dst + avail, 0, (size_t)(row_bytes - avail));
            }
            JPEGDRAW jd;
            jd.x = s->xoff + r[0]; jd.y = s->yoff + r[1];
            jd.iWidth = r[2]; jd.iHeight = r[3]; jd.iWidthUsed = r[4]; jd.iBpp = r[5];
            jd.pPixels = buf; jd.pUser = s->user;
            const int keep_going = (*s->draw)(&jd);    // jpeg.inl:5325
            half ^= 1;
            if (!keep_going) break;
        }
    }
    if (partial) { s->error = JPEG_DECODE_ERROR; return 0; }   // jpeg.inl:5354-5356
    return 1;
}
