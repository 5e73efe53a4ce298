// jda_runtime_internal.h -- what the pieces of the device runtime (jda_runtime.cpp, jda_pipeline.cpp) share.
#ifndef JDA_RUNTIME_INTERNAL_H
#define JDA_RUNTIME_INTERNAL_H

#include <hip/hip_runtime.h>

#include "jda_internal.h"
#include "jda_plan.h"

// launch lists of a batch: one per (mode, fast_mul, kernel variant, window size, P1 in chunks);
// index = (((mode * 2 + fast) * 4 + variant) * 2 + big) * 2 + cont
#define JDA_N_LISTS (32 * JDA_N_MODES)
#define JDA_LIST_MODE(m) ((m) >> 5)
#define JDA_LIST_FAST(m) (((m) >> 4) & 1)
#define JDA_LIST_VARIANT(m) (((m) >> 2) & 3)
#define JDA_LIST_BIG(m) (((m) >> 1) & 1)
#define JDA_LIST_CONT(m) ((m) & 1)

#define JDA_POOL_SLOTS 192
#define JDA_POOL_IDLE_MAX ((size_t)2 << 30)      // idle bytes kept at most
#define JDA_MAX_BANDS 8
struct jda_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev_start, ev_stop;
    hipEvent_t ev_band[JDA_MAX_BANDS];   // jda_decode_to_host_bands: one per band of the copy back (made on first use)
    uint8_t *pinned;          // page-locked staging for uploads (grow-only, reused)
    size_t pinned_cap;
    int last_segscan_rounds;  // speculative rounds the last marker-less device pre-scan needed (diagnostics)
    char last_error[256];
    // Device blocks the runtime allocated for itself (resident images, launch plans, the one-call path's surface), kept when
    // released and handed out again: hipFree costs ~0.23 ms and synchronises the device, which was most of a small image's
    // time through jda_decode_to_host (the JPEGDEC class).  Everything of a context runs on its one stream, so a block
    // released while work on it is still queued is safe to reuse: the next user's work queues behind it.
    struct { void *p; size_t bytes; bool busy; } pool[JDA_POOL_SLOTS];
    size_t pool_idle;
};

struct jda_dev_image {
    uint8_t *base;            // one allocation: tables | index | dc | scan
    size_t bytes;
    size_t off_tables, off_index, off_dc, off_scan;
    jda_image_info info;
    uint32_t scan_len, n_mcus_ok;
    uint8_t dc_id[3], ac_id[3], q_id[3];
    uint8_t fast_mul;
    uint8_t general_p1;          // JDA_DESC_GENERAL_P1
    uint8_t prescan_on_device;   // the block index was made on the device (jda_segscan_*)
    size_t off_cont_first, off_cont;   // continuation entries (0 / 0: none uploaded)
    uint32_t n_cont;
    uint32_t tiles_total, tiles_over_small;   // host index known: tiles, and those whose scan slice exceeds the 16-wave kernel's window (0 / 0: unknown)
    uint8_t *tables_host;     // a host copy of the tables (JDA_TABLE_BYTES): a launch plan lets consecutive images with equal tables share the first one's copy
};

struct jda_batch {
    int32_t n_images;
    jda_dev_desc *d_descs;
    jda_strip *d_strips[JDA_N_LISTS];
    uint32_t n_strips[JDA_N_LISTS];
    jda_batch_stats stats;
    uint32_t flat_max_items;    // JDA_LIST_THUMB_FLAT: the launch's width
    int32_t *status;            // per image: JDA_SUCCESS / JDA_DECODE_ERROR (bad MCU) / JDA_INVALID_PARAMETER (hole)
};


int jda_set_err(jda_ctx *ctx, hipError_t e, const char *what);
hipError_t jda_pool_alloc(jda_ctx *ctx, void **out, size_t bytes);
void jda_pool_free(jda_ctx *ctx, void *p);
extern "C" jda_batch *jda_batch_create_strips(jda_ctx *ctx, int32_t n, jda_dev_image *const *images, const jda_output *outputs, const int32_t *pixel_types,
                                              const int32_t *options, const int32_t *mcu_rects, const int32_t *strip_mcus, int32_t *err);
int jda_plain_variant(const jda_dev_desc &D);
extern "C" int jda_host_range_of(const void *p, size_t len, uintptr_t *base, size_t *bytes);   // the page-locked range (jda_host_alloc / jda_host_register) that holds [p, p + len)
// which launch list an image's tiles go to: ((mode * 2 + fast) * 4 + variant) * 2 + big.  The combination fast = 0, variant = 3 (no
// plain-case kernel exists without the 24-bit multiplies) names the DC thumbnail kernel: 1/8 scale -- also every progressive
// file's DC scan at its default scale --, whose pixels are the blocks' DC values (jpeg.inl:5146-5154): no scan, no index, no IDCT
#define JDA_LIST_THUMB(mode) (((((mode) * 2 + 0) * 4 + 3) * 2 + 0) * 2 + 0)
// .. and the same with big = 1 the 1/4-scale kernel (jda_quarter_tiles): a block is its DC value, <= 4 AC symbols and a 2x2 IDCT
// (jpeg.inl:2305-2326).  (A strip-major surface at 1/4 stays with the decode kernel, whose colour stage knows the layout.)
#define JDA_LIST_QUARTER(mode) (((((mode) * 2 + 0) * 4 + 3) * 2 + 1) * 2 + 0)
// .. and (fast = 0, variant = 3, cont = 1) the flat thumbnail kernels: a WHOLE gray or 4:2:0 image at 1/8 on a row-major surface is a
// pointwise map of its DC array -- the list holds one record per image (jda_dc_thumbnail_flat)
#define JDA_LIST_THUMB_FLAT(mode) (((((mode) * 2 + 0) * 4 + 3) * 2 + 0) * 2 + 1)
#define JDA_LIST_IS_THUMB_FLAT(m) ((m) == JDA_LIST_THUMB_FLAT(JDA_MODE_GRAY) || (m) == JDA_LIST_THUMB_FLAT(JDA_MODE_420))
// whole: every MCU of the image is launched (no crop rectangle)
inline int jda_list_index(const jda_dev_desc &D, int variant, int big, int cont = 0, bool whole = false)
{
    if (D.strip_mcus == 0) {                                      // (a strip-major surface stays with the decode kernel, whose colour stage knows the layout)
        if (D.scale_shift == 3 && whole && (D.mode == JDA_MODE_GRAY || D.mode == JDA_MODE_420)) return JDA_LIST_THUMB_FLAT(D.mode);
        if (D.scale_shift == 3) return JDA_LIST_THUMB(D.mode);
        if (D.scale_shift == 2) return JDA_LIST_QUARTER(D.mode);
    }
    return (((D.mode * 2 + (D.fast_mul ? 1 : 0)) * 4 + variant) * 2 + big) * 2 + cont;
}
int jda_use_cont(const jda_dev_desc &D, int variant, uint32_t n_cont);
int jda_big_window(const jda_dev_desc &D, int variant, uint32_t tiles_total = 0, uint32_t tiles_over_small = 0);
// Fill the descriptor of one image of a launch plan (everything but the pointers into the image's HBM block, which the
// caller sets) and validate the output surface.  Returns JDA_SUCCESS or the error jda_batch_create reports.
int jda_fill_launch_desc(jda_dev_desc &D, const jda_image_info &I, const uint8_t dc_id[3], const uint8_t ac_id[3], const uint8_t q_id[3],
                         int fast_mul, int general_p1, uint32_t n_mcus_ok, uint32_t scan_len, const jda_output &O, int pixel_type, int options,
                         int *bpp_out);

extern "C" hipError_t jda_launch_segscan_fused(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t round, hipStream_t stream);
extern "C" hipError_t jda_launch_checksum(const void *base, uint32_t pitch, uint32_t row_bytes, uint32_t rows, unsigned long long *out, hipStream_t stream);
extern "C" hipError_t jda_launch_segscan_tail(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t first_round, uint32_t max_round, hipStream_t stream);
extern "C" hipError_t jda_launch_filter(const jda_filter_params *params, uint32_t n_images, uint32_t max_raw_len, hipStream_t stream);
extern "C" hipError_t jda_launch_fill_strips(const jda_strips_params *params, uint32_t n_images, uint32_t max_tiles, hipStream_t stream);
extern "C" hipError_t jda_launch_walk_tables(const jda_segscan_params *params, uint32_t n_images, hipStream_t stream);   // before the first walk
extern "C" hipError_t jda_launch_segscan_sums(const jda_segscan_params *params, uint32_t n_images, hipStream_t stream);
extern "C" hipError_t jda_launch_prescan_passes(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t list_rounds, uint32_t max_round,
                                                int any_record, hipStream_t stream);
// the same, or -- states_first -- the order for a batch too small to fill the GPU: exit states first (SPEC walks), then ONE recording round
extern "C" hipError_t jda_launch_prescan_passes_ex(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t list_rounds, uint32_t max_round,
                                                   int any_record, int states_first, hipStream_t stream);
// aux: JDA_LIST_THUMB_FLAT: the most items an image of the batch's flat lists has (jda_flat_items); 0 otherwise
extern "C" hipError_t jda_launch_decode(int mode, int fast_mul, int variant, int big, int cont, const jda_dev_desc *descs, const jda_strip *strips,
                                        uint32_t n_strips, uint32_t aux, hipStream_t stream);
inline uint32_t jda_flat_items(const jda_dev_desc &D) { return (D.mode == JDA_MODE_GRAY ? (D.mcus_x + 3u) >> 2 : D.mcus_x) * D.mcus_y; }      // quads of blocks (gray) / MCUs

#endif
