// tests/class_cpu/stub_runtime.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A CPU stand-in for the four device entry points the drop-in class (jpegdec_amd/csrc/JPEGDEC.cpp) calls, so that the class's HOST
// logic -- option handling, crop rounding, the JPEGDRAW replay, framebuffer wrap / clip, EXIF thumbnails, error codes, what one call
// leaves behind for the next -- can be run here, without a GPU, against the walks recorded from the unmodified reference
// (tests/golden/api_walks.json, script_walks.json), and under AddressSanitizer.  The pixels come from the oracle's CPU restatement
// (oracle/jpegdec_oracle.c), the number of MCUs in front of a bad one from the product's own serial pre-scan (jda_frontend.cpp, host
// code): exactly what the device runtime reports.  Nothing here is measured or shipped; the GPU tests run the same walks through
// the real runtime.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/jpegdec_amd.h"
extern "C" {
#include "../../oracle/jpegdec_oracle.h"
}

struct jda_ctx { int device; };

extern "C" {

void *jda_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void jda_host_free(void *p) { free(p); }
jda_ctx *jda_create(int32_t device, int32_t *err) { if (err) *err = JDA_SUCCESS; jda_ctx *c = new jda_ctx; c->device = device; return c; }
void jda_destroy(jda_ctx *ctx) { delete ctx; }

int jda_decode_to_host_flags(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags);
int jda_decode_to_host_rect(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                            void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles)
{
    return jda_decode_to_host_flags(ctx, jpeg, len, pixel_type, options, mcu_rect, host_pixels, pitch_bytes, rows, mcus_decoded, tiles, 0);
}
int jda_decode_to_host_flags(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags)
{
    if (mcus_decoded) *mcus_decoded = 0;
    if (tiles) tiles[0] = tiles[1] = 0;
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    int32_t err = JDA_SUCCESS;
    jda_image *img = jda_prepare(jpeg, len, &err);
    if (!img) return err;
    const jda_image_info I = *jda_image_get_info(img);
    uint32_t nok = 0;
    (void)jda_image_block_index(img, &nok);
    jda_image_free(img);
    if (mcus_decoded) *mcus_decoded = (int32_t)nok;
    int bpp, ow, oh, cw, ch;
    const int rc = jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) return rc;
    std::vector<uint8_t> canvas((size_t)cw * bpp * ch, 0);
    int oerr = 0;
    (void)orc_decode(jpeg, len, pixel_type, options, canvas.data(), cw * bpp, ch, &oerr);
    const bool complete = nok == (uint32_t)(I.mcus_x * I.mcus_y);
    // MCUs from the bad one on are zeros in the product (the oracle may have left a partly written one)
    const int mw = cw / I.mcus_x, mh = ch / I.mcus_y;
    if (!complete)
        for (uint32_t m = nok; m < (uint32_t)(I.mcus_x * I.mcus_y); m++) {
            const int mx = (int)(m % (uint32_t)I.mcus_x), my = (int)(m / (uint32_t)I.mcus_x);
            for (int r = 0; r < mh; r++) memset(canvas.data() + ((size_t)(my * mh + r) * cw + (size_t)mx * mw) * bpp, 0, (size_t)mw * bpp);
        }
    int r0 = 0, r1 = rows < ch ? rows : ch, x0 = 0, x1 = cw;
    if (mcu_rect) {                                                  // only the rectangle's rows are written, zeros left and right of it
        r0 = mcu_rect[1] * mh; r1 = mcu_rect[3] * mh < r1 ? mcu_rect[3] * mh : r1;
        x0 = mcu_rect[0] * mw; x1 = mcu_rect[2] * mw < cw ? mcu_rect[2] * mw : cw;
        if (x1 < x0) x1 = x0;
    }
    const size_t row_bytes = (size_t)cw * bpp < (size_t)pitch_bytes ? (size_t)cw * bpp : (size_t)pitch_bytes;
    if (!complete && (flags & JDA_TO_HOST_KEEP_UNDECODED) && !mcu_rect) {      // only the MCUs in front of the bad one reach the caller's buffer
        const int full = (int)(nok / (uint32_t)I.mcus_x), part = (int)(nok % (uint32_t)I.mcus_x);
        for (int r = 0; r < r1 && r < (full + 1) * mh; r++) {
            const size_t nbytes = r < full * mh ? row_bytes : std::min(row_bytes, (size_t)part * mw * bpp);
            memcpy((uint8_t *)host_pixels + (size_t)r * pitch_bytes, canvas.data() + (size_t)r * cw * bpp, nbytes);
        }
        return JDA_DECODE_ERROR;
    }
    for (int r = r0; r < r1; r++) {
        uint8_t *dst = (uint8_t *)host_pixels + (size_t)r * pitch_bytes;
        memset(dst, 0, row_bytes);
        const size_t a = (size_t)x0 * bpp, b = (size_t)x1 * bpp < row_bytes ? (size_t)x1 * bpp : row_bytes;
        if (b > a) memcpy(dst + a, canvas.data() + (size_t)r * cw * bpp + a, b - a);
    }
    return complete ? JDA_SUCCESS : JDA_DECODE_ERROR;
}

// (the copy back in bands: the stand-in fills the canvas, then reports the bands one after the other -- the class's resumable replay
// runs exactly as it does behind the GPU's copies)
int jda_decode_to_host_bands(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags,
                             int32_t n_bands, jda_band_callback *band_ready, void *user)
{
    const int rc = jda_decode_to_host_flags(ctx, jpeg, len, pixel_type, options, mcu_rect, host_pixels, pitch_bytes, rows, mcus_decoded, tiles, flags);
    if ((rc == JDA_SUCCESS || rc == JDA_DECODE_ERROR) && band_ready && n_bands > 1 && !mcu_rect && !(flags & JDA_TO_HOST_KEEP_UNDECODED)) {
        jda_image_info I;
        int bpp, ow, oh, cw, ch;
        if (jda_parse(jpeg, len, &I) == JDA_SUCCESS && jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch) == JDA_SUCCESS && I.mcus_y > 0) {
            const int mh = ch / I.mcus_y, r1 = rows < ch ? rows : ch;
            int nb = n_bands > 8 ? 8 : n_bands;
            const int mrows = (r1 + mh - 1) / mh;
            if (nb > mrows) nb = mrows;
            const int per = ((mrows + nb - 1) / nb) * mh;
            for (int k = 0; k < nb && k * per < r1; k++) (*band_ready)(user, k * per, std::min(r1, (k + 1) * per));
        }
    }
    return rc;
}

// strip-major: the row-major stand-in canvas, rearranged as the kernels would have written it
int jda_decode_to_host_strips(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, int32_t strip_mcus,
                              void *host_pixels, size_t host_bytes, int32_t *mcus_decoded, int32_t n_bands, jda_band_callback *band_ready, void *user)
{
    jda_image_info I;
    int bpp, ow, oh, cw, ch;
    int rc = jda_parse(jpeg, len, &I);
    if (rc != JDA_SUCCESS) return rc;
    rc = jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) return rc;
    if (strip_mcus <= 0 || I.mcus_x <= 0 || I.mcus_y <= 0) return JDA_INVALID_PARAMETER;
    const int mw = cw / I.mcus_x, mh = ch / I.mcus_y, n_sx = (I.mcus_x + strip_mcus - 1) / strip_mcus;
    const size_t strip_bytes = (size_t)strip_mcus * mw * mh * bpp;
    if (host_bytes < strip_bytes * n_sx * I.mcus_y) return JDA_INVALID_PARAMETER;
    std::vector<uint8_t> canvas((size_t)cw * ch * bpp, 0);
    rc = jda_decode_to_host_flags(ctx, jpeg, len, pixel_type, options, NULL, canvas.data(), cw * bpp, ch, mcus_decoded, NULL, 0);
    if (rc != JDA_SUCCESS && rc != JDA_DECODE_ERROR) return rc;
    for (int y = 0; y < I.mcus_y; y++)
        for (int sx = 0; sx < n_sx; sx++) {
            const int m0 = sx * strip_mcus, mc = std::min(strip_mcus, I.mcus_x - m0);
            uint8_t *dst = (uint8_t *)host_pixels + ((size_t)y * n_sx + sx) * strip_bytes;
            for (int r = 0; r < mh; r++)
                memcpy(dst + (size_t)r * mc * mw * bpp, canvas.data() + ((size_t)(y * mh + r) * cw + (size_t)m0 * mw) * bpp, (size_t)mc * mw * bpp);
        }
    if (band_ready && n_bands > 1) {
        int nb = n_bands > 8 ? 8 : n_bands;
        if (nb > I.mcus_y) nb = I.mcus_y;
        const int per = (I.mcus_y + nb - 1) / nb;
        for (int k = 0; k < nb && k * per < I.mcus_y; k++) (*band_ready)(user, k * per * mh, std::min(I.mcus_y, (k + 1) * per) * mh);
    }
    return rc;
}

int jda_decode_to_host_ex(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, void *host_pixels,
                          int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded)
{
    return jda_decode_to_host_rect(ctx, jpeg, len, pixel_type, options, NULL, host_pixels, pitch_bytes, rows, mcus_decoded, NULL);
}

} // extern "C"
