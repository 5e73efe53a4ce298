// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE: a lane-by-lane CPU emulation of one wavefront.
//
// Compiles the very same per-lane decode logic the HIP kernels use (jpegdec_amd/csrc/
// jda_device_core.h) with g++ and runs every strip the way the kernel does: all 64 lanes through
// phase A, then all 64 lanes through phase B, over a byte array standing in for the wave's LDS.
// It exists so the kernel logic can be checked against the oracle on machines without a GPU
// (`pytest -m "not gpu"`).  It is not part of libjpegdec_amd.so and nothing in the product calls it.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../jpegdec_amd/csrc/jda_device_core.h"
#include "../../jpegdec_amd/csrc/jda_plan.h"

static int g_reverse_tiles = 0;   // tests run the tiles in reverse order too: results must not depend on which wave finishes first
extern "C" void hostsim_set_reverse(int on) { g_reverse_tiles = on; }
static uint32_t g_window_bytes = JDA_WIN_BYTES;   // tests shrink it to exercise the HBM fall-back of the bit reader
extern "C" void hostsim_set_window(uint32_t bytes) { g_window_bytes = bytes > JDA_WIN_BYTES ? JDA_WIN_BYTES : (bytes & ~15u); }

// one wavefront = 64 lanes stepping through the kernel's phases; a phase runs for every lane before
// the next one starts (= the wave-local fence between them)
template <int MODE, bool FAST>
static void run_tiles(const jda_dev_desc &D, const std::vector<jda_strip> &tiles)
{
    typedef jda_lds_layout<MODE> L;
    std::vector<uint64_t> tab_store((JDA_LT_BYTES + 7) / 8), lds_store((L::WAVE_BYTES + 7) / 8);
    uint8_t *tab = (uint8_t *)tab_store.data(), *wl = (uint8_t *)lds_store.data();
    for (uint32_t tid = 0; tid < 256; tid++) jda_p0_tables(D, tid, 256, tab);
    for (size_t ii = 0; ii < tiles.size(); ii++) {
        const size_t i = g_reverse_tiles ? tiles.size() - 1 - ii : ii;
        const jda_strip &S = tiles[i];
        memset(wl, 0xA5, L::WAVE_BYTES);         // poison: LDS is not zero-initialised on the GPU either
        const jda_tile_ctx C = jda_tile_setup<MODE>(D, S);
        jda_p1_inputs in[JDA_TILE_THREADS];
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) in[t] = jda_p1_prefetch<MODE>(D, C, t);
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p0_stage<MODE>(D, C, t, wl, g_window_bytes);
        uint32_t flags[JDA_TILE_THREADS];
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) flags[t] = jda_p1_entropy<MODE>(D, C, in[t], tab, wl, wl + L::WIN_OFF, g_window_bytes);
        if (D.scale_shift < 2) {
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p1_lists<MODE>(t, flags[t], flags, wl);
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p2_columns<MODE, FAST>(D, t, tab, wl);
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p3_rows<MODE>(D, t, tab, wl);
        }
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p4_output<MODE>(D, S, C, t, wl);
    }
}

extern "C" int hostsim_decode(const uint8_t *jpeg, int len, int pixel_type, int options,
                              uint8_t *out, int pitch_bytes, int width_px, int rows)
{
    int32_t err = 0;
    jda_image *img = jda_prepare(jpeg, len, &err);
    if (!img) return err ? err : -1;
    jda_dev_desc D;
    jda_output o;
    o.pixels = out; o.pitch_bytes = pitch_bytes; o.width_px = width_px; o.rows = rows;
    int rc = jda_fill_desc(D, img, pixel_type, options, o);
    if (rc != JDA_SUCCESS) { jda_image_free(img); return rc; }
    uint32_t n;
    D.scan = jda_image_scan(img, &n);
    D.blk_index = jda_image_block_index(img, &n);
    D.blk_dc = jda_image_block_dc(img);
    D.tables = jda_image_tables(img, &n);
    std::vector<jda_strip> strips;
    jda_append_strips(strips, 0, D.mcus_x, D.mcus_y, D.mode);
    switch (D.mode * 2 + (D.fast_mul ? 1 : 0)) {
    case JDA_MODE_GRAY * 2: run_tiles<JDA_MODE_GRAY, false>(D, strips); break;
    case JDA_MODE_GRAY * 2 + 1: run_tiles<JDA_MODE_GRAY, true>(D, strips); break;
    case JDA_MODE_444 * 2: run_tiles<JDA_MODE_444, false>(D, strips); break;
    case JDA_MODE_444 * 2 + 1: run_tiles<JDA_MODE_444, true>(D, strips); break;
    case JDA_MODE_420 * 2: run_tiles<JDA_MODE_420, false>(D, strips); break;
    default: run_tiles<JDA_MODE_420, true>(D, strips); break;
    }
    const jda_image_info *I = jda_image_get_info(img);
    rc = (D.n_mcus_ok == (uint32_t)(I->mcus_x * I->mcus_y)) ? JDA_SUCCESS : JDA_DECODE_ERROR;
    jda_image_free(img);
    return rc;
}

extern "C" int hostsim_fast_mul(const uint8_t *jpeg, int len)
{
    int32_t err = 0;
    jda_image *img = jda_prepare(jpeg, len, &err);
    if (!img) return -1;
    int f = (int)jda_image_fast_mul(img);
    jda_image_free(img);
    return f;
}
