// jda_plan.h -- host-side construction of the device descriptors and the strip work list.
// Shared by the HIP runtime (jda_runtime.cpp) and the unit-test wave emulator (tests/hostsim).
#ifndef JDA_PLAN_H
#define JDA_PLAN_H

#include <string.h>

#include <vector>

#include "jda_internal.h"
#include "jda_device_core.h"     // the LDS layout decides how many wavefronts (= tiles) share a workgroup

extern "C" void jda_image_component_ids(const jda_image *img, uint8_t *dc_id, uint8_t *ac_id, uint8_t *q_id);
extern "C" uint32_t jda_image_fast_mul(const jda_image *img);
extern "C" uint32_t jda_image_general_p1(const jda_image *img);
extern "C" int jda_host_prescan_threads(void);      // the threads a host pre-scan runs on (the caller + the helpers: jda_frontend.cpp, RstPool)


// descriptor byte pad_[0]: bits 1:0 profiling switches (JDA_DEBUG_SKIP), bit 2 = JDA_DESC_GENERAL_P1, bit 3 = the scan holds DC symbols only (first scan
// of a progressive file), bits 7:4 = Al, the point transform of those DC differences (jpeg.inl:1884)
inline uint8_t jda_desc_stream_bits(const jda_image_info &I)
{
    return I.jpeg_type == 1 ? (uint8_t)(JDA_DESC_DC_ONLY | ((I.approx & 15) << 4)) : (uint8_t)0;
}

inline int jda_mode_of(const jda_image_info &I)
{
    if (I.subsample == 0x22) return JDA_MODE_420;
    if (I.subsample == 0x21) return JDA_MODE_422;
    if (I.subsample == 0x12) return JDA_MODE_440;
    if (I.subsample == 0x11) return JDA_MODE_444;
    return JDA_MODE_GRAY;
}

// Fill everything of the descriptor except the device pointers.  Returns JDA_* status.
inline int jda_fill_desc(jda_dev_desc &D, const jda_image *img, int pixel_type, int options,
                         const jda_output &out)
{
    const jda_image_info &I = *jda_image_get_info(img);
    memset(&D, 0, sizeof(D));
    if (pixel_type < 0 || pixel_type > JDA_EIGHT_BIT_GRAYSCALE) return JDA_INVALID_PARAMETER;   // src/JPEGDEC.cpp:47-53
    if ((options & JDA_LUMA_ONLY) && pixel_type < JDA_EIGHT_BIT_GRAYSCALE) pixel_type = JDA_EIGHT_BIT_GRAYSCALE; // jpeg.inl:4991-4993
    int bpp, ow, oh, cw, ch;
    int rc = jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) return rc;
    options = jda_effective_options(&I, options);                   // progressive: 1/8 thumbnail from the DC scan
    D.mode = (uint8_t)jda_mode_of(I);
    D.ncomp = (uint8_t)I.ncomp;
    // gray JPEG with an RGB8888 request is drawn as RGB565 by the reference
    D.pixel_type = (uint8_t)((D.mode == JDA_MODE_GRAY && pixel_type == JDA_RGB8888) ? JDA_RGB565_BIG_ENDIAN : pixel_type);
    D.scale_shift = (uint8_t)((options & JDA_SCALE_HALF) ? 1 : (options & JDA_SCALE_QUARTER) ? 2 : (options & JDA_SCALE_EIGHTH) ? 3 : 0);
    D.gray_from_color = (uint8_t)(D.mode != JDA_MODE_GRAY && pixel_type == JDA_EIGHT_BIT_GRAYSCALE);
    jda_image_component_ids(img, D.dc_id, D.ac_id, D.q_id);
    D.fast_mul = (uint8_t)jda_image_fast_mul(img);
    D.pad_[0] = (uint8_t)(jda_desc_stream_bits(I) | (jda_image_general_p1(img) ? JDA_DESC_GENERAL_P1 : 0u));
    D.mcus_x = (uint32_t)I.mcus_x;
    D.mcus_y = (uint32_t)I.mcus_y;
    uint32_t nok = 0, slen = 0;
    jda_image_block_index(img, &nok);
    jda_image_scan(img, &slen);
    D.n_mcus_ok = nok;
    D.scan_len = slen;
    D.out = (uint8_t *)out.pixels;
    D.out_pitch = (uint32_t)out.pitch_bytes;
    D.out_w = (uint32_t)(out.width_px < cw ? out.width_px : cw);
    D.out_rows = (uint32_t)(out.rows < ch ? out.rows : ch);
    if (out.pitch_bytes < (int)D.out_w * bpp) return JDA_INVALID_PARAMETER;
    return JDA_SUCCESS;
}

// Tiles of one image: each MCU row is cut into runs of <= 64 blocks (10 MCUs of 4:2:0, 20 of 4:4:4, 16 of 4:2:2 / 4:4:0,
// 64 of gray); one wavefront decodes one tile, and a workgroup is as many wavefronts as fit in a CU's LDS
// next to one copy of the tables (16 for 4:2:0, 15 otherwise).  The list is padded with empty tiles per
// image so that a workgroup never spans two images (it stages one table set).
inline uint32_t jda_tiles_per_wg(int mode, int big = 0)
{
    switch (mode) {
    case JDA_MODE_420: return (uint32_t)jda_lds_layout<JDA_MODE_420>::WAVES - (uint32_t)big;
    case JDA_MODE_444: return (uint32_t)jda_lds_layout<JDA_MODE_444>::WAVES - (uint32_t)big;
    case JDA_MODE_422: return (uint32_t)jda_lds_layout<JDA_MODE_422>::WAVES - (uint32_t)big;
    case JDA_MODE_440: return (uint32_t)jda_lds_layout<JDA_MODE_440>::WAVES - (uint32_t)big;
    default: return (uint32_t)jda_lds_layout<JDA_MODE_GRAY>::WAVES - (uint32_t)big;
    }
}
// bytes of scan window a wavefront of the kernel has (BIG = 0 / 1)
inline uint32_t jda_window_bytes(int mode, int big)
{
    switch (mode) {
    case JDA_MODE_420: return big ? (uint32_t)jda_lds_layout<JDA_MODE_420, 1>::WIN_BYTES : (uint32_t)jda_lds_layout<JDA_MODE_420, 0>::WIN_BYTES;
    case JDA_MODE_444: return big ? (uint32_t)jda_lds_layout<JDA_MODE_444, 1>::WIN_BYTES : (uint32_t)jda_lds_layout<JDA_MODE_444, 0>::WIN_BYTES;
    case JDA_MODE_422: return big ? (uint32_t)jda_lds_layout<JDA_MODE_422, 1>::WIN_BYTES : (uint32_t)jda_lds_layout<JDA_MODE_422, 0>::WIN_BYTES;
    case JDA_MODE_440: return big ? (uint32_t)jda_lds_layout<JDA_MODE_440, 1>::WIN_BYTES : (uint32_t)jda_lds_layout<JDA_MODE_440, 0>::WIN_BYTES;
    default: return big ? (uint32_t)jda_lds_layout<JDA_MODE_GRAY, 1>::WIN_BYTES : (uint32_t)jda_lds_layout<JDA_MODE_GRAY, 0>::WIN_BYTES;
    }
}
inline uint32_t jda_mcus_per_tile(int mode)
{
    return mode == JDA_MODE_420 ? 10u : mode == JDA_MODE_444 ? 20u : (mode == JDA_MODE_422 || mode == JDA_MODE_440) ? 16u : 64u;
}

// rect = {mx0, my0, mx1, my1} in MCUs (half open) restricts the list to the tiles of that rectangle (crop-aware decode:
// the per-block index lets a tile start at any MCU); NULL = the whole image
// edge_mcus != 0: no tile crosses a multiple of edge_mcus MCUs (a strip-major surface: jda_dev_desc::strip_mcus)
// same_tables: the image shares its tables with the one appended before it (one table generation: jda_strip::ord)
inline void jda_append_strips(std::vector<jda_strip> &v, uint32_t image, uint32_t mcus_x, uint32_t mcus_y, int mode, int big = 0, const int32_t *rect = nullptr, uint32_t edge_mcus = 0,
                              bool same_tables = false)
{
    const uint32_t per = jda_mcus_per_tile(mode);
    const uint32_t ord = v.empty() ? 0u : v.back().ord + (same_tables ? 0u : 1u);      // images are appended one after the other
    uint32_t x0 = 0, y0 = 0, x1 = mcus_x, y1 = mcus_y;
    if (rect) {
        x0 = rect[0] < 0 ? 0u : (uint32_t)rect[0]; y0 = rect[1] < 0 ? 0u : (uint32_t)rect[1];
        x1 = rect[2] < 0 ? 0u : ((uint32_t)rect[2] < mcus_x ? (uint32_t)rect[2] : mcus_x);
        y1 = rect[3] < 0 ? 0u : ((uint32_t)rect[3] < mcus_y ? (uint32_t)rect[3] : mcus_y);
    }
    bool first = true;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0, step = per; x < x1; x += step) {
            jda_strip s;
            memset(&s, 0, sizeof(s));
            s.image = image; s.mcu_y = (uint16_t)y; s.mcu_x0 = (uint16_t)x;
            step = x1 - x < per ? x1 - x : per;
            if (edge_mcus && (x / edge_mcus + 1u) * edge_mcus - x < step) step = (x / edge_mcus + 1u) * edge_mcus - x;
            s.count = (uint8_t)step;
            s.first = first ? 1 : 0; s.ord = ord;
            first = false;
            v.push_back(s);
        }
    if (first) {                                                   // an empty rectangle: one padding tile keeps the image's place (ord) in the list
        jda_strip s;
        memset(&s, 0, sizeof(s));
        s.image = image; s.first = 1; s.ord = ord;
        v.push_back(s);
    }
    while (v.size() % jda_tiles_per_wg(mode, big)) {
        jda_strip s;
        memset(&s, 0, sizeof(s));
        s.image = image; s.ord = ord;
        v.push_back(s);
    }
}

#endif
