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
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include <unordered_map>
#include <vector>

#include "../../include/jpegdec_amd.h"

#define JDA_MAX_REPLAY_BANDS 8          // bands of the copy back of a large image decoded with draw callbacks

// What an open image IS, as plain data: the source, the parsed header, everything the setters set.  A C++ object holds it inside
// its state; a C JPEGIMAGE holds exactly this (the caller's memory, as the reference's 18 KB struct is the caller's).
struct jpegdec_amd_settings {
    const uint8_t *data;
    int size;
    jda_image_info info;
    int error;
    int pixel_type;
    int max_mcus;
    int options;
    int xoff, yoff;
    int crop_x, crop_y, crop_w, crop_h;
    int device;                       // -1: $JPEGDEC_AMD_DEVICE / 0
    void *user;
    void *framebuffer;
    JPEG_DRAW_CALLBACK *draw;
    JPEG_CLOSE_CALLBACK *close_cb;
    void *close_handle;
    bool opened;
};
struct jpegdec_amd_state : jpegdec_amd_settings {
    std::vector<uint8_t> owned;       // file-sourced data (open(filename) / callbacks)
    // the decoded canvas the draw callbacks / the framebuffer copy are replayed from, and the replay's strip plan: kept from one
    // decode to the next (they grow to the largest image the object has decoded and go with it) -- a fresh canvas per decode was
    // a memset and a page fault per 4 KB of it: 0.2 ms of a 640x480 decode, 5 ms of a 4096x4096 one
    std::vector<uint8_t> canvas;
    uint8_t *pinned_canvas;           // large images: page-locked, so that the copy back runs beside the strip replay (jda_decode_to_host_bands)
    size_t pinned_cap;
    jpegdec_amd_state() : pinned_canvas(NULL), pinned_cap(0) {}
    ~jpegdec_amd_state() { if (pinned_canvas) jda_host_free(pinned_canvas); }
    jpegdec_amd_state(const jpegdec_amd_state &) = delete;
    jpegdec_amd_state &operator=(const jpegdec_amd_state &) = delete;
    std::vector<int32_t> rects;
    // strip handed to the draw callback; two halves alternate with JPEG_USES_DMA (jpeg.inl:5073-5076)
    alignas(16) uint16_t strip[MAX_BUFFERED_PIXELS + 8];
};

namespace {

// One device context per THREAD and device, created on the thread's first decode there and destroyed when the thread ends:
// objects used from different threads ("one JPEGDEC object per thread" is the reference's threading model) decode
// concurrently -- each context has its own HIP stream, staging buffer and block pool, so nothing is shared and nothing is
// locked.  Device of an object: setDevice() / JPEG_setDevice(), else $JPEGDEC_AMD_DEVICE, else 0.
struct ThreadContexts {
    enum { kMax = 16 };
    jda_ctx *ctx[kMax];
    int err[kMax];
    ThreadContexts() { for (int i = 0; i < kMax; i++) { ctx[i] = NULL; err[i] = JDA_SUCCESS; } }
    ~ThreadContexts() { for (int i = 0; i < kMax; i++) if (ctx[i]) jda_destroy(ctx[i]); }
};
thread_local ThreadContexts t_contexts;

int default_device()
{
    static const int dev = []() { const char *e = getenv("JPEGDEC_AMD_DEVICE"); return e ? atoi(e) : 0; }();
    return dev;
}

jda_ctx *thread_ctx(int device, int *err)
{
    if (device < 0) device = default_device();
    if (device < 0 || device >= ThreadContexts::kMax) { if (err) *err = JDA_INVALID_PARAMETER; return NULL; }
    ThreadContexts &T = t_contexts;
    if (!T.ctx[device] && T.err[device] == JDA_SUCCESS) {
        int32_t rc = JDA_SUCCESS;
        T.ctx[device] = jda_create(device, &rc);
        T.err[device] = T.ctx[device] ? JDA_SUCCESS : rc;
    }
    if (err) *err = T.err[device];
    return T.ctx[device];
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

JPEGDEC::JPEGDEC() : _jpeg(new jpegdec_amd_state) { _jpeg->device = -1; reset(_jpeg); }
JPEGDEC::~JPEGDEC() { delete _jpeg; }
// a copy: the open image with everything set on it; data read from a file is cloned, RAM / FLASH data stays the caller's; the
// close callback stays with the original (it closes the caller's file once); the decoded canvas is a cache and starts empty
static jpegdec_amd_state *clone_state(const jpegdec_amd_state *o)
{
    jpegdec_amd_state *s = new jpegdec_amd_state;
    s->owned = o->owned;
    s->data = (!o->owned.empty() && o->data == o->owned.data()) ? s->owned.data() : o->data;
    s->size = o->size; s->info = o->info; s->error = o->error; s->pixel_type = o->pixel_type; s->max_mcus = o->max_mcus;
    s->options = o->options; s->xoff = o->xoff; s->yoff = o->yoff;
    s->crop_x = o->crop_x; s->crop_y = o->crop_y; s->crop_w = o->crop_w; s->crop_h = o->crop_h;
    s->device = o->device; s->user = o->user; s->framebuffer = o->framebuffer; s->draw = o->draw;
    s->close_cb = NULL; s->close_handle = NULL; s->opened = o->opened;
    return s;
}
JPEGDEC::JPEGDEC(const JPEGDEC &o) : _jpeg(clone_state(o._jpeg)) {}
JPEGDEC &JPEGDEC::operator=(const JPEGDEC &o)
{
    if (this != &o) { jpegdec_amd_state *s = clone_state(o._jpeg); delete _jpeg; _jpeg = s; }
    return *this;
}
JPEGDEC::JPEGDEC(JPEGDEC &&o) : _jpeg(new jpegdec_amd_state)      // (allocates: not noexcept -- the source stays a valid, closed object)
{
    _jpeg->device = -1; reset(_jpeg);
    jpegdec_amd_state *t = _jpeg; _jpeg = o._jpeg; o._jpeg = t;
}
JPEGDEC &JPEGDEC::operator=(JPEGDEC &&o) noexcept
{
    if (this != &o) { jpegdec_amd_state *t = _jpeg; _jpeg = o._jpeg; o._jpeg = t; reset(o._jpeg); }
    return *this;
}

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
void JPEGDEC::setDevice(int iDevice) { _jpeg->device = iDevice; }      // (not in the reference: which GPU this object decodes on)
void JPEGDEC::setUserPointer(void *p) { _jpeg->user = p; }
int JPEGDEC::getOrientation() { return _jpeg->info.orientation; }
int JPEGDEC::getLastError() { return _jpeg->error; }
int JPEGDEC::getWidth() { return _jpeg->info.width; }
int JPEGDEC::getHeight() { return _jpeg->info.height; }
int JPEGDEC::getBpp() { return _jpeg->info.bpp; }
int JPEGDEC::getSubSample() { return _jpeg->info.subsample; }
int JPEGDEC::getJPEGType() { return _jpeg->info.jpeg_type ? JPEG_MODE_PROGRESSIVE : JPEG_MODE_BASELINE; }
int JPEGDEC::hasThumb() { return _jpeg->info.has_thumb; }
int JPEGDEC::getThumbWidth() { return _jpeg->info.thumb_w; }
int JPEGDEC::getThumbHeight() { return _jpeg->info.thumb_h; }
int JPEGDEC::getPixelType() { return _jpeg->pixel_type; }

void JPEGDEC::setPixelType(int iType)
{
    if (iType >= 0 && iType < INVALID_PIXEL_TYPE) _jpeg->pixel_type = iType;
    else _jpeg->error = JPEG_INVALID_PARAMETER;        // src/JPEGDEC.cpp:47-53
}

void JPEGDEC::setMaxOutputSize(int iMaxMCUs) { _jpeg->max_mcus = iMaxMCUs < 1 ? 1 : iMaxMCUs; }

// Crop rectangle, rounded to MCU boundaries as the reference does (jpeg.inl:682-727)
void JPEGDEC::setCropArea(int x, int y, int w, int h)
{
    int32_t cx = x, cy = y, cw = w, ch = h;
    jda_crop_round(&_jpeg->info, &cx, &cy, &cw, &ch);
    _jpeg->crop_x = cx; _jpeg->crop_y = cy; _jpeg->crop_w = cw; _jpeg->crop_h = ch;
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
    if (s->pixel_type > EIGHT_BIT_GRAYSCALE) { s->error = JPEG_UNSUPPORTED_FEATURE; return 0; }
    if (iOptions & JPEG_EXIF_THUMBNAIL) {              // jpeg.inl:4967-4976: decode the JPEG embedded in the EXIF block instead
        if (s->info.thumb_offset == 0 || s->info.thumb_w == 0) { s->error = JPEG_INVALID_PARAMETER; return 0; }
        if (s->info.thumb_offset < 0 || s->info.thumb_offset > s->size - 256) { s->error = JPEG_INVALID_FILE; return 0; }   // (JPEGParseInfo wants 256 bytes, :1598)
        // jpeg.inl:4964-4966 runs BEFORE the thumbnail is parsed: it is the MAIN image's mode that ORs JPEG_SCALE_EIGHTH in
        if (s->info.jpeg_type == 1) iOptions |= JPEG_SCALE_EIGHTH;
        jda_image_info ti;
        const int prc = jda_parse(s->data + s->info.thumb_offset, s->size - s->info.thumb_offset, &ti);   // JPEGParseInfo(pJPEG, 1)
        if (prc != JDA_SUCCESS) { s->error = prc; return 0; }
        // a progressive thumbnail inside a baseline file would go through the reference's DC-only decode at whatever scale was
        // asked for (no EIGHTH OR-ed in): not a case this path reproduces
        if (ti.jpeg_type == 1 && s->info.jpeg_type == 0 && !(iOptions & (JPEG_SCALE_HALF | JPEG_SCALE_EIGHTH))) { s->error = JPEG_UNSUPPORTED_FEATURE; return 0; }
        // the reference parses the thumbnail over its own state: the object now describes the thumbnail
        ti.has_thumb = s->info.has_thumb; ti.thumb_w = s->info.thumb_w; ti.thumb_h = s->info.thumb_h; ti.thumb_offset = 0;
        if (!ti.orientation) ti.orientation = s->info.orientation;       // (ucOrientation is only written when an orientation tag is met: the main image's stays)
        s->data += s->info.thumb_offset; s->size -= s->info.thumb_offset;
        s->info = ti;
        s->crop_x = s->crop_y = 0; s->crop_w = ti.width; s->crop_h = ti.height;      // the SOF handler resets the whole crop rectangle (jpeg.inl:1683-1685)
        iOptions &= ~JPEG_EXIF_THUMBNAIL;
    }
    if (s->crop_w <= 0 || s->crop_h <= 0) { s->error = JPEG_INVALID_PARAMETER; return 0; }   // image smaller than one MCU / overhanging request (jpeg.inl:713-719 leaves w <= 0): nothing sane to deliver
    const bool cropped = s->crop_x != 0 || s->crop_y != 0 || s->crop_w != s->info.width || s->crop_h != s->info.height;
    int pt = s->pixel_type;
    if ((iOptions & JPEG_LUMA_ONLY) && pt < EIGHT_BIT_GRAYSCALE) pt = s->pixel_type = EIGHT_BIT_GRAYSCALE;   // jpeg.inl:4991-4993
    int bpp, ow, oh, cw, ch;
    int rc = jda_output_geometry(&s->info, pt, iOptions, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) { s->error = s->info.mcu_w ? rc : JPEG_UNSUPPORTED_FEATURE; return 0; }
    int cerr = JDA_SUCCESS;
    jda_ctx *ctx = thread_ctx(s->device, &cerr);
    if (!ctx) { s->error = cerr; return 0; }

    const int eff0 = jda_effective_options(&s->info, iOptions);
    const int shift0 = (eff0 & JPEG_SCALE_HALF) ? 1 : (eff0 & JPEG_SCALE_QUARTER) ? 2 : (eff0 & JPEG_SCALE_EIGHTH) ? 3 : 0;
    int32_t mcus_decoded = 0;
    // Framebuffer mode, full size, no crop, a width that is a whole number of MCUs: the caller's buffer has the layout of the
    // decoded canvas (pitch = width, MCU-padded rows: what the reference writes, jpeg.inl:5114-5124), so the copy back from the
    // GPU lands in it directly -- no intermediate canvas, no second copy
    if (s->framebuffer && !cropped && shift0 == 0 && cw == s->info.width && s->crop_w == s->info.width) {
        // (a stream with a bad MCU: the MCUs in front of it land in the framebuffer, the rest of it is left alone -- the reference returns there)
        rc = jda_decode_to_host_flags(ctx, s->data, s->size, pt, iOptions, NULL, s->framebuffer, cw * bpp, ch, &mcus_decoded, NULL, JDA_TO_HOST_KEEP_UNDECODED);
        if (rc != JDA_SUCCESS && rc != JDA_DECODE_ERROR) { s->error = rc; return 0; }
        if (rc == JDA_DECODE_ERROR) { s->error = JPEG_DECODE_ERROR; return 0; }   // jpeg.inl:5354-5356
        return 1;
    }
    const size_t canvas_bytes = (size_t)cw * ch * bpp;
    const bool banded = !s->framebuffer && s->draw && !cropped && canvas_bytes >= ((size_t)2 << 20);      // a large image with draw callbacks
    if (banded && s->pinned_cap < canvas_bytes) {
        if (s->pinned_canvas) jda_host_free(s->pinned_canvas);
        s->pinned_canvas = (uint8_t *)jda_host_alloc(canvas_bytes + canvas_bytes / 8);
        s->pinned_cap = s->pinned_canvas ? canvas_bytes + canvas_bytes / 8 : 0;
    }
    const bool use_pinned = banded && s->pinned_canvas != NULL;
    if (!use_pinned && s->canvas.size() < canvas_bytes) s->canvas.resize(canvas_bytes);
    uint8_t *const canvas = use_pinned ? s->pinned_canvas : s->canvas.data();    // (every row the replay reads is copied back by this decode: the MCU rows it keeps)
    // A cropped decode only launches the tiles of the MCUs the reference keeps (jpeg.inl:5111, :5134-5137: MCU rows from the crop's
    // first row on, MCU columns from iCropX up to and including the one AT iCropX + iCropCX -- the '>' there).  The reference
    // still entropy-decodes what it skips (it has to, to find the next MCU); the per-block index makes that unnecessary here.
    int32_t rect[4] = { 0, 0, s->info.mcus_x, s->info.mcus_y };
    if (cropped) {
        const int mw0 = s->info.mcu_w >> shift0, mh0 = s->info.mcu_h >> shift0;
        int x0 = s->info.mcus_x, x1 = 0;
        for (int mx = 0; mx < s->info.mcus_x; mx++)
            if (!(mx * mw0 < s->crop_x || mx * mw0 > s->crop_x + s->crop_w)) { if (mx < x0) x0 = mx; x1 = mx + 1; }
        int y0 = 0;
        while (y0 < s->info.mcus_y && y0 * mh0 < s->crop_y) y0++;
        int y1 = (s->crop_y + s->crop_h + s->info.mcu_h - 1) / s->info.mcu_h;
        if (y1 > s->info.mcus_y) y1 = s->info.mcus_y;
        rect[0] = x0 < x1 ? x0 : 0; rect[1] = y0; rect[2] = x0 < x1 ? x1 : 0; rect[3] = y1 > y0 ? y1 : y0;
    }
    // ---- callback mode: the reference's JPEGDRAW sequence (jpeg.inl:5300-5336), planned before the decode so that the strips of a band
    // of the copy back can go out while the next band is still on the bus (jda_decode_to_host_bands)
    int n = 0, ri = 0, half = 0;
    bool stopped = false, partial = false, dma = false;
    std::vector<int32_t> &rects = s->rects;
    uint16_t *strip0 = s->strip;
    size_t half_words = MAX_BUFFERED_PIXELS / 2;
    std::vector<uint16_t> big;
    const int eff = jda_effective_options(&s->info, iOptions);      // a progressive file is a 1/8 thumbnail (jpeg.inl:4964-4966)
    const int shift = (eff & JPEG_SCALE_HALF) ? 1 : (eff & JPEG_SCALE_QUARTER) ? 2 : (eff & JPEG_SCALE_EIGHTH) ? 3 : 0;
    const int mw = s->info.mcu_w >> shift, mh = s->info.mcu_h >> shift;
    if (!s->framebuffer && s->draw) {
        const int32_t crop[4] = { s->crop_x, s->crop_y, s->crop_w, s->crop_h };
        n = jda_draw_plan_at(&s->info, pt, iOptions, s->max_mcus, (iOptions & JPEG_USES_DMA) ? 1 : 0, cropped ? crop : NULL, s->xoff, NULL, 0);   // how many strips
        if (n > 65536) n = 65536;
        if (n > 0 && s->rects.size() < (size_t)8 * n) s->rects.resize((size_t)8 * n);
        if (n > 0) (void)jda_draw_plan_at(&s->info, pt, iOptions, s->max_mcus, (iOptions & JPEG_USES_DMA) ? 1 : 0, cropped ? crop : NULL, s->xoff, rects.data(), n);
        // with JPEG_USES_DMA (and no user cap on the MCU count) the strip ping-pongs between the two halves
        dma = (iOptions & JPEG_USES_DMA) != 0;
        {   // the halves only alternate when the user cap did not win (jpeg.inl:5071-5076)
            int per_call = MAX_BUFFERED_PIXELS / ((s->info.mcu_w >> shift) * mh);
            if (pt == RGB8888) per_call /= 2;
            if (pt == EIGHT_BIT_GRAYSCALE) per_call *= 2;
            if (per_call > s->info.mcus_x) per_call = s->info.mcus_x;
            if (per_call > s->max_mcus) dma = false;
        }
        if (n > 65536) n = 65536;
        // The reference's strip buffer is usPixels[2048] inside its state; a plan can ask for more (a decode x offset on an image whose
        // width is not a whole number of MCUs widens the row's last strip, jpeg.inl:5328-5335 -- the reference then writes over the
        // tables behind usPixels).  Here such a strip gets a buffer of its own size: the callback sees the strip the plan describes.
        size_t need = 0;
        for (int i = 0; i < n; i++) {
            const int32_t *r = &rects[(size_t)8 * i];
            const size_t bytes = (size_t)(r[2] > 0 ? r[2] : 0) * bpp * (size_t)mh;
            if (bytes > need) need = bytes;
        }
        if (need > (dma ? sizeof(s->strip) / 2 : sizeof(s->strip)) - 16) {
            half_words = (need + 31) / 2 & ~(size_t)7;
            big.assign(half_words * 2 + 16, 0);
            strip0 = big.data();
        }
    }
    // A large image with draw callbacks leaves the GPU STRIP-MAJOR when its plan is the regular one (every strip of a row strip_mcus
    // MCUs wide, the last one what is left: any whole-image decode): the kernels write every strip's pixels contiguously
    // (jda_decode_to_host_strips), the callback gets a pointer into the page-locked canvas and no strip is copied together here
    int strip_mcus = 0, n_sx = 0;
    size_t strip_bytes = 0;
    if (use_pinned && n > 0 && mw > 0) {
        strip_mcus = rects[2] / mw;
        bool regular = strip_mcus > 0 && rects[2] == strip_mcus * mw;
        for (int i = 0; i < n && regular; i++) {
            const int32_t *r = &rects[(size_t)8 * i];
            const int left = s->info.mcus_x * mw - r[6];
            regular = r[6] % (strip_mcus * mw) == 0 && r[7] % mh == 0 && r[2] == (left < strip_mcus * mw ? left : strip_mcus * mw);
        }
        if (regular) {
            n_sx = (s->info.mcus_x + strip_mcus - 1) / strip_mcus;
            strip_bytes = (size_t)strip_mcus * mw * mh * bpp;
            const size_t total = strip_bytes * n_sx * s->info.mcus_y;
            if (s->pinned_cap < total) {
                jda_host_free(s->pinned_canvas);
                s->pinned_canvas = (uint8_t *)jda_host_alloc(total + total / 8);
                s->pinned_cap = s->pinned_canvas ? total + total / 8 : 0;
            }
            if (!s->pinned_canvas) { s->error = JPEG_ERROR_MEMORY; return 0; }
        } else strip_mcus = 0;
    }
    uint8_t *const strips = strip_mcus ? s->pinned_canvas : NULL;
    auto replay = [&](int row_limit) {
        for (; ri < n && !stopped; ri++) {
            const int i = ri;
            const int32_t *r = &rects[(size_t)8 * i];
            if (r[7] + mh > row_limit) return;                   // its rows have not landed yet: the next band brings them
            if (strips) {                                        // strip-major: the strip is where the GPU wrote it
                if (partial) {
                    int x_last = (r[6] + r[2]) / mw - 1;
                    if (x_last > s->info.mcus_x - 1) x_last = s->info.mcus_x - 1;
                    if ((r[7] / mh) * s->info.mcus_x + x_last >= mcus_decoded) { stopped = true; break; }
                }
                JPEGDRAW jd;
                jd.x = s->xoff + r[0]; jd.y = s->yoff + r[1];
                jd.iWidth = r[2]; jd.iHeight = r[3]; jd.iWidthUsed = r[4]; jd.iBpp = r[5];
                jd.pPixels = (uint16_t *)(strips + ((size_t)(r[7] / mh) * n_sx + (size_t)(r[6] / (strip_mcus * mw))) * strip_bytes);
                jd.pUser = s->user;
                if (!(*s->draw)(&jd)) { stopped = true; break; }     // jpeg.inl:5325
                continue;
            }
            if (partial) {
                // the reference returns at the first bad MCU (jpeg.inl:5150-5297 "if (iErr) ... return 0" paths): only the strips it
                // had completed before that MCU reach the callback
                int x_last = (r[6] + r[2]) / mw - 1;
                if (x_last > s->info.mcus_x - 1) x_last = s->info.mcus_x - 1;
                if ((r[7] / mh) * s->info.mcus_x + x_last >= mcus_decoded) { stopped = true; break; }
            }
            uint16_t *buf = strip0 + (dma ? half * half_words : 0);
            const int row_bytes = r[2] > 0 ? r[2] * bpp : 0;
            // A strip is a few hundred bytes from each of mh canvas rows, a canvas pitch apart: not a pattern the hardware prefetcher
            // follows.  Ask for the rows of the strip after the next one while this one is copied.
            if (i + 2 < n && (size_t)cw * ch * bpp > ((size_t)2 << 20)) {      // (a canvas that fits the caches needs no help)
                const int32_t *q = &rects[(size_t)8 * (i + 2)];
                const int qb = q[2] > 0 ? q[2] * bpp : 0;
                for (int rr = 0; rr < mh && q[7] + rr < ch; rr++) {
                    const uint8_t *src = canvas + ((size_t)(q[7] + rr) * cw + q[6]) * bpp;
                    for (int o = 0; o < qb; o += 64) __builtin_prefetch(src + o, 0, 0);
                }
            }
            for (int rr = 0; rr < mh; rr++) {
                const int cy_ = r[7] + rr;                       // strip position in the decoded canvas
                uint8_t *dst = (uint8_t *)buf + (size_t)rr * row_bytes;
                if (cy_ >= ch) { memset(dst, 0, (size_t)row_bytes); continue; }
                int avail = (cw - r[6]) * bpp;
                if (avail > row_bytes) avail = row_bytes;
                if (avail < 0) avail = 0;
                memcpy(dst, canvas + ((size_t)cy_ * cw + r[6]) * bpp, (size_t)avail);
                if (avail < row_bytes) memset(dst + avail, 0, (size_t)(row_bytes - avail));
            }
            JPEGDRAW jd;
            jd.x = s->xoff + r[0]; jd.y = s->yoff + r[1];
            jd.iWidth = r[2]; jd.iHeight = r[3]; jd.iWidthUsed = r[4]; jd.iBpp = r[5];
            jd.pPixels = buf; jd.pUser = s->user;
            const int keep_going = (*s->draw)(&jd);    // jpeg.inl:5325
            half ^= 1;
            if (!keep_going) { stopped = true; break; }
        }
    };
    const int32_t mcus_total = s->info.mcus_x * s->info.mcus_y;
    if (use_pinned) {
        // a large image with draw callbacks: the copy back in bands, each band's strips replayed as it lands
        struct Trampoline { decltype(replay) *fn; int32_t *decoded; int32_t total; bool *partial; };
        Trampoline tr = { &replay, &mcus_decoded, mcus_total, &partial };
        if (strips)
            rc = jda_decode_to_host_strips(ctx, s->data, s->size, pt, iOptions, strip_mcus, strips, s->pinned_cap, &mcus_decoded, JDA_MAX_REPLAY_BANDS,
                                           [](void *u, int32_t, int32_t row1) { Trampoline *t = (Trampoline *)u; *t->partial = *t->decoded < t->total; (*t->fn)(row1); }, &tr);
        else
        rc = jda_decode_to_host_bands(ctx, s->data, s->size, pt, iOptions, NULL, canvas, cw * bpp, ch, &mcus_decoded, NULL, 0, JDA_MAX_REPLAY_BANDS,
                                      [](void *u, int32_t, int32_t row1) { Trampoline *t = (Trampoline *)u; *t->partial = *t->decoded < t->total; (*t->fn)(row1); }, &tr);
    } else
    rc = jda_decode_to_host_rect(ctx, s->data, s->size, pt, iOptions, cropped ? rect : NULL, canvas, cw * bpp, ch, &mcus_decoded, NULL);
    // the reference walks the MCU rows down to the crop's bottom only (jpeg.inl:5014-5037): a bad MCU below it is never met
    if (rc == JDA_DECODE_ERROR && cropped && mcus_decoded >= rect[3] * s->info.mcus_x && rect[3] < s->info.mcus_y) rc = JDA_SUCCESS;
    partial = rc == JDA_DECODE_ERROR;                 // the reference still delivers the MCUs before the bad one
    if (rc != JDA_SUCCESS && !partial) { s->error = rc; return 0; }

    // MCU rows the reference walks (jpeg.inl:5014-5037); a crop that reaches below the last MCU row makes it
    // decode whatever follows the scan and fail -- here the real rows are delivered and the same error returned
    int rows_mcu = (s->crop_y + s->crop_h + s->info.mcu_h - 1) / s->info.mcu_h;
    const bool overrun = rows_mcu > s->info.mcus_y;
    if (overrun) rows_mcu = s->info.mcus_y;

    if (s->framebuffer) {
        // jpeg.inl:5114-5124 + :5134-5137: no callbacks; the kept MCUs of a row are laid side by side at a pitch
        // of iCropCX pixels (unscaled, also for scaled output).  The reference keeps the MCU at
        // x*mcuCX == iCropX+iCropCX too ('>' at :5135) and the last MCU of a ragged width, so a row of MCUs can be
        // wider than the pitch.  What happens to the part past the pitch depends on the JPEGPutMCU* variant:
        //  - the scaled paths and JPEGPutMCU8BitGray for 8x8 MCUs (:2799-2840) do not clip: the overhang lands at
        //    the start of the next buffer row, after that row's own pixels were written ("wrap", replayed in the
        //    same order here; only the overhang of the very last row is dropped instead of overrunning the buffer);
        //  - the full-size paths of 4:4:4, gray -> RGB565 and 4:2:0 clip at the pitch (:3017-3030, :3079-3095, :3520-3524, :4318-4330).
        // Known divergence: a 4:4:4 image whose width is not a multiple of 8, full size, RGB565/RGB8888: the
        // reference's clipped last MCU advances pCb/pCr but not pY per row (:3521-3557), so its last partial MCU
        // column mixes luma of the wrong pixels; this path delivers the correct pixels there.
        //  - neither do JPEGPutMCU21 / JPEGPutMCU12 (4:2:2, 4:4:0) at any size or pixel type (:4546-4868, :2842-2943: fixed 8- / 16-pixel loops).
        const bool wrap = shift != 0 || (pt == EIGHT_BIT_GRAYSCALE && s->info.mcu_w == 8) || s->info.subsample == 0x21 || s->info.subsample == 0x12;
        const int pitch_px = s->crop_w;
        int x0 = 0, x1 = -1;
        for (int mx = 0; mx < s->info.mcus_x; mx++) {
            if (mx * mw < s->crop_x || mx * mw > s->crop_x + s->crop_w) continue;
            if (x1 < 0) x0 = mx;
            x1 = mx;
        }
        const int strip_px = x1 < 0 ? 0 : (x1 - x0 + 1) * mw;
        const int main_px = strip_px < pitch_px ? strip_px : pitch_px;
        int over_px = wrap ? strip_px - main_px : 0;
        if (over_px > pitch_px) over_px = pitch_px;
        long last_row = -1;
        for (int my = 0; my < rows_mcu; my++) if (my * mh >= s->crop_y) last_row = (long)my * mh - s->crop_y + mh - 1;
        uint8_t *fb0 = (uint8_t *)s->framebuffer;
        for (int my = 0; my < rows_mcu; my++) {
            if (my * mh < s->crop_y) continue;                   // bSkipRow, :5111
            const long ty = (long)my * mh - s->crop_y;
            // 1-byte pixels: the row start is computed in 16-bit units, usPixels += ty*iPitch/2 (:5118-5119),
            // so an odd ty*iPitch starts one byte early
            uint8_t *fb = fb0 - ((bpp == 1) ? ((ty * pitch_px) & 1) : 0);
            for (int pass = 0; pass < 2; pass++)
                for (int rr = 0; rr < mh; rr++) {
                    const uint8_t *src = canvas + ((size_t)(my * mh + rr) * cw + (size_t)x0 * mw) * bpp;
                    if (pass == 0) memcpy(fb + (size_t)(ty + rr) * pitch_px * bpp, src, (size_t)main_px * bpp);
                    else if (over_px > 0 && ty + rr + 1 <= last_row)
                        memcpy(fb + (size_t)(ty + rr + 1) * pitch_px * bpp, src + (size_t)main_px * bpp, (size_t)over_px * bpp);
                }
        }
    } else if (s->draw) {
        replay(1 << 30);                                      // (what the bands of the copy back have not delivered yet: everything, when it was not banded)
    }
    if (partial || overrun) { s->error = JPEG_DECODE_ERROR; return 0; }   // jpeg.inl:5354-5356
    return 1;
}


// ---- C flavour (reference src/jpeg.inl:564-739).  The caller's JPEGIMAGE holds the open image as plain data (jpegdec_amd_settings:
// source pointer, parsed header, what the setters set) -- the caller's memory, like the reference's struct: any number of handles,
// no initialisation before JPEG_open*, no JPEG_close for RAM / FLASH sources (src/JPEGDEC.cpp:232-236), nothing shared between
// handles, no table and no lock inside the library.  A file-sourced handle also owns the file's bytes (read whole at open: the GPU
// path needs the whole scan) until JPEG_close, as the reference's owns its open file.  The work is done by one JPEGDEC object per
// THREAD that a call loads the handle's settings into and stores them back from; the decoded canvas and the page-locked copy-back
// buffer are that object's and serve every handle the thread decodes.
#define JPEGIMAGE_MAGIC0 0x4a444133u   /* "JDA3" */
#define JPEGIMAGE_MAGIC1 0x9b1e5a7du
struct jpegdec_amd_c_api { static jpegdec_amd_state *state(JPEGDEC &j) { return j._jpeg; } };
namespace {
static_assert(sizeof(jpegdec_amd_settings) <= sizeof(((JPEGIMAGE *)0)->state), "JPEGIMAGE holds the settings");
thread_local JPEGDEC t_cworker;
// a live handle: opened by this library (64 bits that stack garbage does not hold), wherever it lies now -- a struct copy of an open
// JPEGIMAGE is an open JPEGIMAGE, as with the reference's plain struct (src/JPEGDEC.h:199-239)
bool c_live(const JPEGIMAGE *p) { return p && p->magic[0] == JPEGIMAGE_MAGIC0 && p->magic[1] == JPEGIMAGE_MAGIC1; }
// The bytes of the files JPEG_openFile read, by address, each under a serial number no other buffer ever had.  Whether a buffer
// is freed is decided HERE, never from the words inside a caller's struct: any number of struct copies name the same buffer, and
// closing (or re-opening) one of them must leave the others knowing that the bytes are gone -- a handle whose (address, serial)
// pair is not in the table is a closed handle.  (Never destroyed: handles on other threads may outlive main()'s statics.)
struct c_file_table {
    std::mutex mu;
    std::unordered_map<const void *, uint64_t> live;
    uint64_t next = 0x6a64615f66696c65ull;
};
c_file_table &c_files() { static c_file_table *t = new c_file_table; return *t; }
uint64_t c_file_adopt(void *bytes)
{
    c_file_table &t = c_files();
    std::lock_guard<std::mutex> g(t.mu);
    const uint64_t serial = ++t.next;
    t.live[bytes] = serial;
    return serial;
}
bool c_file_alive(const JPEGIMAGE *p)
{
    if (!c_live(p) || !p->file_data) return false;
    c_file_table &t = c_files();
    std::lock_guard<std::mutex> g(t.mu);
    auto it = t.live.find(p->file_data);
    return it != t.live.end() && it->second == p->file_check;
}
// gives the bytes back if this handle still names a live buffer; true when it did
bool c_file_release(JPEGIMAGE *p)
{
    if (!c_live(p) || !p->file_data) return false;
    c_file_table &t = c_files();
    {
        std::lock_guard<std::mutex> g(t.mu);
        auto it = t.live.find(p->file_data);
        if (it == t.live.end() || it->second != p->file_check) return false;
        t.live.erase(it);
    }
    free(p->file_data);
    return true;
}
// a file-sourced handle whose bytes another copy of it has given back is a closed handle
bool c_stale_file(const JPEGIMAGE *p) { return c_live(p) && p->file_data && !c_file_alive(p); }
jpegdec_amd_state *c_load(JPEGIMAGE *p)
{
    if (!c_live(p) || c_stale_file(p)) return NULL;
    jpegdec_amd_state *s = jpegdec_amd_c_api::state(t_cworker);
    memcpy(static_cast<jpegdec_amd_settings *>(s), p->state, sizeof(jpegdec_amd_settings));
    return s;
}
void c_store(JPEGIMAGE *p) { memcpy(p->state, static_cast<const jpegdec_amd_settings *>(jpegdec_amd_c_api::state(t_cworker)), sizeof(jpegdec_amd_settings)); }
// JPEG_open*: the reference memsets its state (jpeg.inl:569).  The handle that READ a file gives its bytes back when it is opened
// again without a close (the reference leaks its FILE there); a copy of it that is opened again leaves them to the original.
void c_begin_open(JPEGIMAGE *p)
{
    if (c_live(p) && p->file_owner == p) c_file_release(p);
    memset(p, 0, sizeof(*p));
    p->magic[0] = JPEGIMAGE_MAGIC0; p->magic[1] = JPEGIMAGE_MAGIC1;
    jpegdec_amd_c_api::state(t_cworker)->device = -1;
}
}
extern "C" {
int JPEG_openRAM(JPEGIMAGE *pJPEG, uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw)
{
    if (!pJPEG) return 0;
    c_begin_open(pJPEG);
    const int rc = t_cworker.openRAM(pData, iDataSize, pfnDraw);
    c_store(pJPEG);
    return rc;
}
int JPEG_openFile(JPEGIMAGE *pJPEG, const char *szFilename, JPEG_DRAW_CALLBACK *pfnDraw)
{
    if (!pJPEG) return 0;
    c_begin_open(pJPEG);
    const int rc = t_cworker.open(szFilename, pfnDraw);
    jpegdec_amd_state *s = jpegdec_amd_c_api::state(t_cworker);
    if (!s->owned.empty()) {                                    // the file's bytes go with the handle (the worker serves other handles next)
        if (rc) pJPEG->file_data = malloc(s->owned.size());     // (a file that was read and did not parse: nothing to keep -- the reference has nothing to free after a failed open either)
        if (pJPEG->file_data) {
            memcpy(pJPEG->file_data, s->owned.data(), s->owned.size()); s->data = (const uint8_t *)pJPEG->file_data;
            pJPEG->file_owner = pJPEG; pJPEG->file_check = c_file_adopt(pJPEG->file_data);
        } else { s->data = NULL; s->size = 0; s->opened = false; if (rc) s->error = JPEG_ERROR_MEMORY; }
        std::vector<uint8_t>().swap(s->owned);
    }
    c_store(pJPEG);
    return pJPEG->file_data ? rc : 0;
}
#define JDA_C_CALL(p, expr) do { if (jpegdec_amd_state *s_ = c_load(p)) { (void)s_; expr; c_store(p); } } while (0)
void JPEG_setFramebuffer(JPEGIMAGE *pJPEG, void *pFramebuffer) { JDA_C_CALL(pJPEG, t_cworker.setFramebuffer(pFramebuffer)); }
void JPEG_setDevice(JPEGIMAGE *pJPEG, int iDevice) { JDA_C_CALL(pJPEG, t_cworker.setDevice(iDevice)); }
void JPEG_setCropArea(JPEGIMAGE *pJPEG, int x, int y, int w, int h) { JDA_C_CALL(pJPEG, t_cworker.setCropArea(x, y, w, h)); }
void JPEG_getCropArea(JPEGIMAGE *pJPEG, int *x, int *y, int *w, int *h) { JDA_C_CALL(pJPEG, t_cworker.getCropArea(x, y, w, h)); }
int JPEG_getWidth(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getWidth() : 0; }
int JPEG_getHeight(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getHeight() : 0; }
int JPEG_decode(JPEGIMAGE *pJPEG, int x, int y, int iOptions)
{
    int rc = 0;
    JDA_C_CALL(pJPEG, rc = t_cworker.decode(x, y, iOptions));
    return rc;
}
int JPEG_decodeDither(JPEGIMAGE *pJPEG, uint8_t *pDither, int iOptions)
{
    int rc = 0;
    JDA_C_CALL(pJPEG, rc = t_cworker.decodeDither(pDither, iOptions));
    return rc;
}
void JPEG_close(JPEGIMAGE *pJPEG)
{
    if (!pJPEG) return;
    if (c_live(pJPEG)) {
        JDA_C_CALL(pJPEG, t_cworker.close());               // (not for a copy whose file another copy has closed: nothing is open there)
        c_file_release(pJPEG);                              // frees only bytes that are still on the library's list
    }
    pJPEG->file_data = NULL; pJPEG->file_owner = NULL; pJPEG->file_check = 0; pJPEG->magic[0] = pJPEG->magic[1] = 0;
}
int JPEG_getLastError(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getLastError() : JPEG_INVALID_PARAMETER; }
int JPEG_getOrientation(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getOrientation() : 0; }
int JPEG_getBpp(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getBpp() : 0; }
int JPEG_getSubSample(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getSubSample() : 0; }
int JPEG_hasThumb(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.hasThumb() : 0; }
int JPEG_getThumbWidth(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getThumbWidth() : 0; }
int JPEG_getThumbHeight(JPEGIMAGE *pJPEG) { return c_load(pJPEG) ? t_cworker.getThumbHeight() : 0; }
void JPEG_setPixelType(JPEGIMAGE *pJPEG, int iType) { JDA_C_CALL(pJPEG, t_cworker.setPixelType(iType)); }
void JPEG_setMaxOutputSize(JPEGIMAGE *pJPEG, int iMaxMCUs) { JDA_C_CALL(pJPEG, t_cworker.setMaxOutputSize(iMaxMCUs)); }
#undef JDA_C_CALL
}
