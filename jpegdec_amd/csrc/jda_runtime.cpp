// jda_runtime.cpp -- device runtime behind the C-ABI: context (device, stream, events), resident
// images, batch launch plans.  The reference has no such layer (it is a single-threaded
// streaming decoder); this is the seam that replaces the body of the MCU loops of
// DecodeJPEG (reference src/jpeg.inl:5109-5353) with one kernel launch per batch.
//
// There is NO CPU fallback here: every entry point reports JDA_ERROR_NO_DEVICE / JDA_ERROR_HIP
// when the HIP device or a HIP call is not available.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <thread>
#include <vector>

#include <map>
#include <mutex>

#include "jda_runtime_internal.h"

extern "C" uint32_t jda_image_fast_mul(const jda_image *img);
extern "C" uint32_t jda_image_general_p1(const jda_image *img);
extern "C" const uint32_t *jda_image_restart_positions(const jda_image *img, uint32_t *n);
extern "C" void jda_image_run_host_prescan(jda_image *img);
extern "C" int jda_image_index_on_device(const jda_image *img);
extern "C" uint32_t jda_image_record_cap(const jda_image *img);
extern "C" void jda_image_adopt_prescan(jda_image *img, uint32_t n_mcus_ok, uint32_t max_ac_bits, int32_t max_abs_dc, uint32_t trunc_events);
// The plain-case kernel variants (jda_desc_uniform in jda_kernels.hip): full size, every multiply in 24 bits, no stream flags,
// and one of the (layout, output format) pairs a kernel was built for.  0 = the general kernel.
int jda_plain_variant(const jda_dev_desc &D)
{
    if (D.scale_shift != 0 || D.pad_[0] != 0 || !D.fast_mul || D.strip_mcus != 0) return 0;
    const bool colour = D.mode != JDA_MODE_GRAY;
    if (D.pixel_type == JDA_RGB8888 && !D.gray_from_color) return (D.mode == JDA_MODE_444 || D.mode == JDA_MODE_420 || D.mode == JDA_MODE_422) ? 1 : 0;
    if (D.pixel_type == JDA_RGB565_LITTLE_ENDIAN && colour && !D.gray_from_color) return (D.mode == JDA_MODE_444 || D.mode == JDA_MODE_420) ? 2 : 0;
    if (D.pixel_type == JDA_EIGHT_BIT_GRAYSCALE) return (D.mode == JDA_MODE_GRAY || (D.mode == JDA_MODE_420 && D.gray_from_color)) ? 3 : 0;
    return 0;
}

// High-bitrate images go to the kernel variant with one wavefront less per CU and a larger scan window (jda_lds_layout<MODE, 1>):
// a tile whose slice of the scan does not fit the window takes the general bit reader, which goes to HBM at every refill.
// Decided per image from its average bytes of scan per full tile (+ 50 % for the spread between tiles), or from the tiles' slices
// themselves where the host made the index; every decode kernel takes either layout (a launch argument).
int jda_big_window(const jda_dev_desc &D, int variant, uint32_t tiles_total, uint32_t tiles_over_small)
{
    static const int forced = []() { const char *e = JDA_LAB_ENV("JDA_BIG_WINDOW"); return e ? atoi(e) : -1; }();
    if (D.scale_shift == 3) return 0;                             // JDA_LIST_THUMB (or, strip-major, the decode kernel without a window)
    if (D.scale_shift == 2 && D.strip_mcus == 0) return 1;        // JDA_LIST_QUARTER: its lists are padded as the large-window kernels' (no window is staged)
    if (forced >= 0) return forced ? 1 : 0;
    // the index was made on the host: the tiles' slices are known exactly -- the larger window (and the wavefront it costs) only
    // when more than one tile in a hundred does not fit the smaller one
    if (tiles_total) return (uint64_t)tiles_over_small * 100u > tiles_total ? 1 : 0;
    const uint64_t n_mcus = (uint64_t)D.mcus_x * D.mcus_y;
    if (!n_mcus) return 0;
    const uint64_t avg = (uint64_t)D.scan_len * jda_mcus_per_tile(D.mode) / n_mcus;
    return avg + avg / 2 + 48 > jda_window_bytes(D.mode, 0) ? 1 : 0;
}

// P1 in chunks: for an image that HAS continuation entries (the serial pre-scan wrote them: the image lies in the window in which
// the mode was measured to pay, or its caller asked for them: JDA_PREPARE_CONT_*) and whose decode has such a kernel -- the RGB8888
// plain case of 4:2:0 and 4:4:4
int jda_use_cont(const jda_dev_desc &D, int variant, uint32_t n_cont)
{
    if (!n_cont || variant != 1 || (D.mode != JDA_MODE_420 && D.mode != JDA_MODE_444)) return 0;
    return 1;
}

int jda_set_err(jda_ctx *ctx, hipError_t e, const char *what)
{
    if (ctx) snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: %s", what, hipGetErrorString(e));
    return JDA_ERROR_HIP;
}

#define JDA_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return jda_set_err((ctx), e_, #call); } while (0)

static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

hipError_t jda_pool_alloc(jda_ctx *ctx, void **out, size_t bytes)
{
    if (!bytes) bytes = 16;
    int best = -1, empty = -1;
    for (int i = 0; i < JDA_POOL_SLOTS; i++) {
        if (!ctx->pool[i].p) { if (empty < 0) empty = i; continue; }
        if (!ctx->pool[i].busy && ctx->pool[i].bytes >= bytes && ctx->pool[i].bytes <= 2 * bytes + 65536 &&
            (best < 0 || ctx->pool[i].bytes < ctx->pool[best].bytes)) best = i;
    }
    if (best >= 0) { ctx->pool[best].busy = true; ctx->pool_idle -= ctx->pool[best].bytes; *out = ctx->pool[best].p; return hipSuccess; }
    const hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess && empty >= 0) { ctx->pool[empty].p = *out; ctx->pool[empty].bytes = bytes; ctx->pool[empty].busy = true; }
    return e;          // (no slot left: the block is not tracked and pool_free hands it to hipFree)
}
void jda_pool_free(jda_ctx *ctx, void *p)
{
    if (!p) return;
    for (int i = 0; i < JDA_POOL_SLOTS; i++)
        if (ctx->pool[i].p == p) {
            ctx->pool[i].busy = false;
            ctx->pool_idle += ctx->pool[i].bytes;
            while (ctx->pool_idle > JDA_POOL_IDLE_MAX) {           // too much idle memory: give the largest idle blocks back
                int big = -1;
                for (int k = 0; k < JDA_POOL_SLOTS; k++)
                    if (ctx->pool[k].p && !ctx->pool[k].busy && (big < 0 || ctx->pool[k].bytes > ctx->pool[big].bytes)) big = k;
                if (big < 0) break;
                (void)hipFree(ctx->pool[big].p);
                ctx->pool_idle -= ctx->pool[big].bytes;
                ctx->pool[big].p = NULL; ctx->pool[big].bytes = 0;
            }
            return;
        }
    (void)hipFree(p);
}

extern "C" {

int jda_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

jda_ctx *jda_create(int32_t device, int32_t *err)
{
    int32_t dummy;
    if (!err) err = &dummy;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) { *err = JDA_ERROR_NO_DEVICE; return NULL; }
    jda_ctx *ctx = new (std::nothrow) jda_ctx;
    if (!ctx) { *err = JDA_ERROR_MEMORY; return NULL; }
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        delete ctx;
        *err = JDA_ERROR_NO_DEVICE;
        return NULL;
    }
    *err = JDA_SUCCESS;
    return ctx;
}

void jda_destroy(jda_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev_start);
    (void)hipEventDestroy(ctx->ev_stop);
    for (int i = 0; i < JDA_MAX_BANDS; i++) if (ctx->ev_band[i]) (void)hipEventDestroy(ctx->ev_band[i]);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (int i = 0; i < JDA_POOL_SLOTS; i++) if (ctx->pool[i].p) (void)hipFree(ctx->pool[i].p);
    delete ctx;
}

const char *jda_last_hip_error(const jda_ctx *ctx) { return ctx ? ctx->last_error : "no context"; }
void *jda_stream(jda_ctx *ctx) { return ctx ? (void *)ctx->stream : NULL; }

// The page-locked host ranges the library has made or been told of (jda_host_alloc / jda_host_register), by base address:
// jda_pipeline_submit_ex(.., JDA_SUBMIT_PINNED_INPUT) lets the copy engine read a file where it lies only when the file lies
// inside ONE of them, and joins two files into one copy command only inside the same one (a copy that runs over the end of a
// page-locked object, or across the gap between two, is not the caller's memory to read).
namespace {
std::mutex g_host_ranges_mu;
std::map<uintptr_t, size_t> g_host_ranges;
void host_range_add(void *p, size_t bytes) { std::lock_guard<std::mutex> g(g_host_ranges_mu); g_host_ranges[(uintptr_t)p] = bytes; }
void host_range_del(void *p) { std::lock_guard<std::mutex> g(g_host_ranges_mu); g_host_ranges.erase((uintptr_t)p); }
}
// the range [*base, *base + *bytes) that holds [p, p + len); 0 when no single known range does
int jda_host_range_of(const void *p, size_t len, uintptr_t *base, size_t *bytes)
{
    std::lock_guard<std::mutex> g(g_host_ranges_mu);
    auto it = g_host_ranges.upper_bound((uintptr_t)p);
    if (it == g_host_ranges.begin()) return 0;
    --it;
    if ((uintptr_t)p + len > it->first + it->second) return 0;
    *base = it->first; *bytes = it->second;
    return 1;
}
void *jda_host_alloc(size_t bytes)
{
    void *p = NULL;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) return NULL;      // (every GPU of the node may read it: jda_node)
    host_range_add(p, bytes ? bytes : 16);
    return p;
}
void jda_host_free(void *p) { if (p) { host_range_del(p); (void)hipHostFree(p); } }
int jda_host_register(void *p, size_t bytes)
{
    if (!p || !bytes || hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) return JDA_ERROR_HIP;
    host_range_add(p, bytes);
    return JDA_SUCCESS;
}
int jda_host_unregister(void *p)
{
    if (!p) return JDA_ERROR_HIP;
    host_range_del(p);
    return hipHostUnregister(p) == hipSuccess ? JDA_SUCCESS : JDA_ERROR_HIP;
}

void *jda_malloc(jda_ctx *ctx, size_t bytes)
{
    if (!ctx) return NULL;
    void *p = NULL;
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) { jda_set_err(ctx, e, "hipMalloc"); return NULL; }
    return p;
}

void jda_free(jda_ctx *ctx, void *dptr)
{
    if (ctx && dptr) { (void)hipSetDevice(ctx->device); (void)hipFree(dptr); }
}

int jda_memset(jda_ctx *ctx, void *dptr, int value, size_t bytes)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    JDA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDA_SUCCESS;
}

int jda_copy_to_host(jda_ctx *ctx, void *host, const void *dptr, size_t bytes)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JDA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDA_SUCCESS;
}

int jda_copy_to_device(jda_ctx *ctx, void *dptr, const void *host, size_t bytes)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    JDA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDA_SUCCESS;
}

// Upload n prepared images.  Images whose block index is still pending (JDA_PREPARE_DEVICE_PRESCAN) get it made on the
// GPU, all of them by the same launches (one grid row per image; the passes of DESIGN.md 5.3) -- the walk is latency-bound per
// lane, so it is the number of segments in flight that makes it fast.  out[i] receives the device image (NULL on failure).
static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int jda_upload_batch(jda_ctx *ctx, int32_t n, jda_image *const *imgs, jda_dev_image **out)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    const bool trace = JDA_LAB_ENV("JDA_UPLOAD_TRACE") != NULL;          // stage timings on stderr (diagnostics)
    const double t_begin = now_ms();
    double t_mark = t_begin;
#define JDA_UP_MARK(what) do { if (trace) { (void)hipStreamSynchronize(ctx->stream); const double t_ = now_ms(); fprintf(stderr, "jda_upload_batch: %-28s %8.3f ms\n", what, t_ - t_mark); t_mark = t_; } } while (0)
    if (n <= 0 || !imgs || !out) return JDA_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    struct Item {
        jda_dev_image *d; uint8_t *stage; std::vector<uint8_t> heap; bool on_device; uint32_t n_int;
        size_t alloc, n_blocks; uint32_t tbytes;
        // the index is made on the device (8f N1 / N2): segments of the scan, see jda_seg_walk
        bool seg_mode; uint32_t n_segs; size_t off_ea, off_sum, off_start, off_wl, off_rp, off_wt, off_ev, off_sstats;
        bool record; uint32_t rec_cap, cand_cap; size_t off_recs, off_cands, zero_end;      // RECORD mode of the pre-scan (no restart intervals)
        std::vector<uint32_t> rp;            // restart positions + the sentinel, until their copy has been made
        uint32_t sst[68];
    };
    std::vector<Item> items((size_t)n);
    int rc = JDA_SUCCESS;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n; i++) out[i] = NULL;
    auto fail_all = [&](int code) {
        for (int i = 0; i < n; i++) { if (items[i].d) { if (items[i].d->base) jda_pool_free(ctx, items[i].d->base); delete items[i].d; items[i].d = NULL; } out[i] = NULL; }
        return code;
    };
    std::vector<jda_segscan_params> seg_params;
    std::vector<int> seg_owner;
    uint32_t max_segs = 0;
    // staging: slices of one page-locked buffer (H2D at link speed, truly asynchronous); when it is full the
    // copies in flight are drained and it is reused from the start
    const size_t kPinnedMax = (size_t)512 << 20;
    size_t pin_off = 0;
    // host copies into the page-locked buffer are the bulk of an upload's host time: they are collected and done by a
    // few threads, and each image's H2D copies are enqueued when its bytes are in place
    struct StageJob { uint8_t *dst; const uint8_t *src; size_t bytes; int image; };
    std::vector<StageJob> stage_jobs;
    hipError_t stage_err = hipSuccess;
    auto flush_stage_jobs = [&]() {
        if (stage_jobs.empty()) return;
        const size_t nj = stage_jobs.size();
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::min<size_t>(std::min<size_t>(nj, 8), hw ? hw : 1);
        std::atomic<size_t> next(0);
        auto work = [&]() { for (size_t k; (k = next.fetch_add(1)) < nj;) memcpy(stage_jobs[k].dst, stage_jobs[k].src, stage_jobs[k].bytes); };
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        for (size_t k = 0; k < nj && stage_err == hipSuccess; k++) {
            Item &it = items[stage_jobs[k].image];
            stage_err = hipMemcpyAsync(it.d->base + it.d->off_tables, it.stage, it.tbytes, hipMemcpyHostToDevice, ctx->stream);
            if (stage_err == hipSuccess) stage_err = hipMemcpyAsync(it.d->base + it.d->off_scan, stage_jobs[k].dst, stage_jobs[k].bytes, hipMemcpyHostToDevice, ctx->stream);
        }
        stage_jobs.clear();
    };
    auto pin_slice = [&](size_t bytes) -> uint8_t * {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > kPinnedMax) return NULL;
        if (bytes > ctx->pinned_cap) {
            flush_stage_jobs();
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            ctx->pinned = NULL; ctx->pinned_cap = 0;
            size_t want = bytes * 8 < kPinnedMax ? bytes * 8 : kPinnedMax;      // room for a few images in flight
            if (want < bytes) want = bytes;
            if (hipHostMalloc((void **)&ctx->pinned, want, hipHostMallocDefault) != hipSuccess) { ctx->pinned = NULL; return NULL; }
            ctx->pinned_cap = want;
            pin_off = 0;
        }
        if (pin_off + bytes > ctx->pinned_cap) { flush_stage_jobs(); (void)hipStreamSynchronize(ctx->stream); pin_off = 0; }
        uint8_t *p = ctx->pinned + pin_off;
        pin_off += bytes;
        return p;
    };
    for (int i = 0; i < n && rc == JDA_SUCCESS; i++) {
        Item &it = items[i];
        jda_image *img = imgs[i];
        if (!img) { rc = JDA_INVALID_PARAMETER; break; }
        const jda_image_info &I = *jda_image_get_info(img);
        uint32_t scan_len = 0, nok = 0;
        const uint8_t *scan = jda_image_scan(img, &scan_len);
        const uint8_t *tables = jda_image_tables(img, &it.tbytes);
        it.n_blocks = (size_t)I.mcus_x * I.mcus_y * I.blocks_per_mcu;
        it.on_device = jda_image_index_on_device(img) != 0;      // index to be made on the device
        if (it.on_device) {                                      // (the record area's bound: jda_pipeline_submit_ex has the same)
            uint32_t sl = 0;
            (void)jda_image_scan(img, &sl);
            if ((uint64_t)(sl / JDA_SEG_BYTES + 1u) * jda_image_record_cap(img) * 4u > 8ull * sl + (16ull << 20)) { jda_image_run_host_prescan(img); it.on_device = false; }
        }
        const uint32_t *rpos = jda_image_restart_positions(img, &it.n_int);
        jda_dev_image *d = new (std::nothrow) jda_dev_image;
        if (!d) { rc = JDA_ERROR_MEMORY; break; }
        memset(d, 0, sizeof(*d));
        it.d = d;
        d->info = I;
        d->scan_len = scan_len;
        jda_image_component_ids(img, d->dc_id, d->ac_id, d->q_id);
        d->general_p1 = (uint8_t)jda_image_general_p1(img);
        d->tables_host = (uint8_t *)malloc(JDA_TABLE_BYTES);
        if (d->tables_host) { uint32_t tb_ = 0; memcpy(d->tables_host, jda_image_tables(img, &tb_), JDA_TABLE_BYTES); }
        d->off_tables = 0;
        d->off_index = align16(it.tbytes);
        d->off_dc = d->off_index + align16((it.n_blocks + 1) * sizeof(uint32_t));
        d->off_scan = d->off_dc + align16(it.n_blocks * sizeof(int16_t));
        d->bytes = d->off_scan + align16((size_t)scan_len + JDA_SCAN_PAD);
        d->off_cont_first = d->off_cont = 0; d->n_cont = 0;
        if (!it.on_device) {                                      // the serial pre-scan's continuation entries travel with its index
            const uint32_t *cf = NULL;
            uint32_t nc = 0;
            (void)jda_image_block_cont(img, &cf, &nc);
            if (nc) {
                d->n_cont = nc;
                d->off_cont_first = d->bytes; d->off_cont = d->off_cont_first + align16((it.n_blocks + 1) * sizeof(uint32_t));
                d->bytes = d->off_cont + align16((size_t)nc * sizeof(uint32_t));
            }
        }
        it.alloc = d->bytes;
        it.seg_mode = it.on_device;
        it.n_segs = 0;
        if (it.seg_mode) {
            // the scan is read in whole 256-byte segments (+ a few bytes): zero padded behind its last byte; then the entry
            // states, the per-segment sums and start values, the two work lists, the restart positions, the result words
            it.n_segs = scan_len / JDA_SEG_BYTES + 1u;
            d->bytes = d->off_scan + align16(std::max((size_t)scan_len + JDA_SCAN_PAD, (size_t)it.n_segs * JDA_SEG_BYTES + 16));
            it.off_ea = d->bytes;
            it.off_sum = it.off_ea + align16(((size_t)it.n_segs + 1) * 4);
            it.off_start = it.off_sum + align16((size_t)it.n_segs * 4 * JDA_SEG_SUM_WORDS);
            it.off_wl = it.off_start + align16((size_t)it.n_segs * 20);
            it.off_rp = it.off_wl + align16((size_t)it.n_segs * 8);
            it.off_wt = it.off_rp + (I.restart_interval ? align16(((size_t)it.n_int + 1) * 4) : 0);
            it.off_ev = it.off_wt + JDA_WT_BYTES;                 // RECORD mode: who ended which restart interval (zeroed)
            it.off_sstats = it.off_ev + (I.restart_interval ? align16((size_t)it.n_int * 8) : 0);
            it.alloc = it.off_sstats + 512;
            it.zero_end = it.alloc;                          // (what is memset: everything up to here; records and candidates need none)
            it.rec_cap = jda_image_record_cap(img);
            it.record = true;                                 // (a stream the device walks has record slots: front_common)
            if (it.record) {
                it.off_recs = (it.alloc + 255) & ~(size_t)255;
                it.off_cands = it.off_recs + align16((size_t)it.n_segs * it.rec_cap * 4);
                it.cand_cap = std::max<uint32_t>(1024u, it.n_segs * 16u);
                it.alloc = it.off_cands + (size_t)it.cand_cap * 16;
            }
        }
        e = jda_pool_alloc(ctx, (void **)&d->base, it.alloc);
        if (e != hipSuccess) { jda_set_err(ctx, e, "hipMalloc(image)"); rc = JDA_ERROR_MEMORY; break; }
        if (it.seg_mode) {
            // only the tables and the scan travel; everything behind the scan's last byte starts as zeros (padding, round-0
            // entry states, counters).  The index is written by the WRITE pass.
            const size_t up = align16(it.tbytes) + align16((size_t)scan_len);
            it.stage = pin_slice(up);
            if (!it.stage) { it.heap.assign(up, 0); it.stage = it.heap.data(); }
            memcpy(it.stage, tables, it.tbytes);
            // the scan's copy into the page-locked slice is left to the staging threads (below): job = (dst, src, bytes, image)
            if (it.heap.empty()) { stage_jobs.push_back({ it.stage + align16(it.tbytes), scan, (size_t)scan_len, i }); }
            else {
                memcpy(it.stage + align16(it.tbytes), scan, scan_len);
                e = hipMemcpyAsync(d->base + d->off_tables, it.stage, it.tbytes, hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(d->base + d->off_scan, it.stage + align16(it.tbytes), scan_len, hipMemcpyHostToDevice, ctx->stream);
            }
            if (e == hipSuccess) e = hipMemsetAsync(d->base + d->off_scan + scan_len, 0, it.zero_end - (d->off_scan + scan_len), ctx->stream);
            if (e != hipSuccess) { rc = jda_set_err(ctx, e, "hipMemcpy(image)"); break; }
            jda_segscan_params SP;
            memset(&SP, 0, sizeof(SP));
            SP.scan = d->base + d->off_scan; SP.tables = d->base + d->off_tables;
            SP.entry_cur = (uint32_t *)(d->base + it.off_ea); SP.entry_nxt = SP.entry_cur;
            SP.seg_sum = (uint32_t *)(d->base + it.off_sum); SP.seg_start = (const uint32_t *)(d->base + it.off_start);
            SP.worklist = (uint32_t *)(d->base + it.off_wl); SP.worklist_cap = it.n_segs;
            if (I.restart_interval) {                        // the walk ends intervals where the (host) filter found the markers
                it.rp.assign(rpos, rpos + it.n_int); it.rp.push_back(JDA_RST_SENTINEL);
                e = hipMemcpyAsync(d->base + it.off_rp, it.rp.data(), it.rp.size() * 4, hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) { rc = jda_set_err(ctx, e, "hipMemcpy(restart positions)"); break; }
                SP.restart_pos = (const uint32_t *)(d->base + it.off_rp); SP.n_intervals = it.n_int;
                SP.interval_blocks = (uint32_t)I.restart_interval * (uint32_t)I.blocks_per_mcu;
                SP.round_last = ((uint32_t)(I.mcus_x * I.mcus_y) % (uint32_t)I.restart_interval) == 0 ? 1u : 0u;
            }
            SP.blk_index = (uint32_t *)(d->base + d->off_index); SP.blk_dc = (int16_t *)(d->base + d->off_dc);
            SP.stats = (uint32_t *)(d->base + it.off_sstats);
            SP.walk_tables = d->base + it.off_wt;
            if (it.record) {
                SP.records = (uint32_t *)(d->base + it.off_recs); SP.rec_cap = it.rec_cap;
                SP.cands = (uint32_t *)(d->base + it.off_cands); SP.cand_cap = it.cand_cap;
                if (I.restart_interval) SP.rst_events = (uint32_t *)(d->base + it.off_ev);
            }
            SP.scan_len = scan_len; SP.n_segs = it.n_segs; SP.n_blocks_total = (uint32_t)it.n_blocks;
            SP.nblocks = (uint8_t)I.blocks_per_mcu; SP.nluma = (uint8_t)(I.blocks_per_mcu - (I.ncomp == 3 ? 2 : 0));
            for (int c = 0; c < 3; c++) { SP.dc_id[c] = d->dc_id[c]; SP.ac_id[c] = d->ac_id[c]; }
            seg_params.push_back(SP); seg_owner.push_back(i);
            if (it.n_segs > max_segs) max_segs = it.n_segs;
            continue;
        }
        // stage through one host buffer so it is a single H2D copy
        it.stage = pin_slice(it.alloc);
        if (!it.stage) { it.heap.assign(it.alloc, 0); it.stage = it.heap.data(); }
        memcpy(it.stage + d->off_tables, tables, it.tbytes);
        memcpy(it.stage + d->off_scan, scan, (size_t)scan_len + JDA_SCAN_PAD);
        {
            const uint32_t *index = jda_image_block_index(img, &nok);
            memcpy(it.stage + d->off_index, index, (it.n_blocks + 1) * sizeof(uint32_t));
            memcpy(it.stage + d->off_dc, jda_image_block_dc(img), it.n_blocks * sizeof(int16_t));
            if (d->n_cont) {
                const uint32_t *cf = NULL;
                const uint32_t *ce = jda_image_block_cont(img, &cf, NULL);
                memcpy(it.stage + d->off_cont_first, cf, (it.n_blocks + 1) * sizeof(uint32_t));
                memcpy(it.stage + d->off_cont, ce, (size_t)d->n_cont * sizeof(uint32_t));
            }
            {   // the scan slice of every tile, as jda_tile_setup_from computes it: how many exceed the 16-wave kernel's window
                const int mode = jda_mode_of(I);
                const uint32_t per = jda_mcus_per_tile(mode), win = jda_window_bytes(mode, 0), nb = (uint32_t)I.blocks_per_mcu;
                uint32_t total = 0, over = 0;
                for (uint32_t y = 0; y < (uint32_t)I.mcus_y; y++)
                    for (uint32_t x = 0; x < (uint32_t)I.mcus_x; x += per) {
                        const uint32_t cnt = (uint32_t)I.mcus_x - x < per ? (uint32_t)I.mcus_x - x : per;
                        const size_t b0 = ((size_t)y * I.mcus_x + x) * nb, b1 = b0 + (size_t)cnt * nb;
                        const uint32_t lo = (index[b0] >> JDA_INDEX_OFF_BITS) & ~15u, hi = ((index[b1] >> JDA_INDEX_OFF_BITS) + 8u + 12u + 15u) & ~15u;
                        total++;
                        if (hi - lo > win) over++;
                    }
                d->tiles_total = total; d->tiles_over_small = over;
            }
        }
        e = hipMemcpyAsync(d->base, it.stage, it.alloc, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { rc = jda_set_err(ctx, e, "hipMemcpy(image)"); break; }
    }
    flush_stage_jobs();
    if (rc == JDA_SUCCESS && stage_err != hipSuccess) rc = jda_set_err(ctx, stage_err, "hipMemcpy(image)");
    if (rc != JDA_SUCCESS) { (void)hipStreamSynchronize(ctx->stream); return fail_all(rc); }
    JDA_UP_MARK("alloc + stage + H2D");

    // ---- the index on the device (with or without restart intervals): round 0 and round 1 over every segment, two rounds over the
    // work lists, every further round in one launch, sums, WRITE -- the passes of jda_pipeline, on this context's stream
    jda_segscan_params *d_seg = NULL;
    if (!seg_params.empty() && e == hipSuccess) {
        const uint32_t ns = (uint32_t)seg_params.size();
        e = jda_pool_alloc(ctx, (void **)&d_seg, seg_params.size() * sizeof(jda_segscan_params));
        if (e == hipSuccess) e = hipMemcpyAsync(d_seg, seg_params.data(), seg_params.size() * sizeof(jda_segscan_params), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = jda_launch_walk_tables(d_seg, ns, ctx->stream);
        // round 0, the recording round, two work-list rounds, every further round in one launch, the sums (first block ordinal, DC
        // predictors, window lag per segment), then finalize + candidates (records -> index entries and DC values)
        // A batch that cannot fill the GPU -- one image at a time is the case -- is as slow as its rounds are long, and a round is one
        // wavefront's chain of walk steps: the entry states are settled with the SPEC walk (half a RECORD walk's instructions) and every
        // segment is recorded once, from its settled state (one 4096x4096 image: 0.07 + 3 x 0.15 ms of rounds -> 4 x 0.07 + 0.15).  A batch
        // that does fill it pays per walk, not per round: there the RECORD round right behind round 0 is the cheaper order.
        uint64_t total_segs = 0;
        for (const jda_segscan_params &sp : seg_params) total_segs += sp.n_segs;
        const int states_first = total_segs <= 131072u ? 1 : 0;                 // (two lanes for every SIMD lane of the GPU)
        if (e == hipSuccess) e = jda_launch_prescan_passes_ex(d_seg, ns, max_segs, 4, 56, 1, states_first, ctx->stream);
        for (uint32_t p = 0; p < ns && e == hipSuccess; p++) {
            Item &it = items[seg_owner[p]];
            e = hipMemcpyAsync(it.sst, it.d->base + it.off_sstats, sizeof(it.sst), hipMemcpyDeviceToHost, ctx->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    JDA_UP_MARK("write pass + D2H");
    if (d_seg) jda_pool_free(ctx, d_seg);
    if (e != hipSuccess) return fail_all(jda_set_err(ctx, e, "jda_upload_batch"));

    // a marker that is not where the MCU count puts it, a corrupt or truncated stream, states that did not settle: the serial host
    // pre-scan reproduces what the reference does with such a stream
    bool reupload = false;
    int rounds_max = 0;
    for (size_t p = 0; p < seg_owner.size() && e == hipSuccess; p++) {
        const int i = seg_owner[p];
        Item &it = items[i];
        const jda_image_info &I = *jda_image_get_info(imgs[i]);
        { int r = 2; while (r <= 57 && it.sst[8 + r]) r++; if (r > rounds_max) rounds_max = r; }
        if (it.sst[7] == 1 && it.sst[6] == 1 && it.sst[0] == 0 && it.sst[1] == 1 && it.sst[5] == 0 && it.sst[8 + 57] == 0) {   // (.. and the states-first order's recording round found every exit state as the SPEC rounds had left it)   // states settled, enough blocks, no bad code before the end, the closing index entry written once, every marker where the MCU count puts it
            jda_image_adopt_prescan(imgs[i], (uint32_t)(I.mcus_x * I.mcus_y), it.sst[2], (int32_t)it.sst[3], it.sst[4]);
            it.d->prescan_on_device = 1;
        } else {                                                         // corrupt or truncated stream: the serial pre-scan knows what the reference does
            jda_image_run_host_prescan(imgs[i]);
            uint32_t nok = 0;
            const uint32_t *index = jda_image_block_index(imgs[i], &nok);
            e = hipMemcpyAsync(it.d->base + it.d->off_index, index, (it.n_blocks + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(it.d->base + it.d->off_dc, jda_image_block_dc(imgs[i]), it.n_blocks * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream);
            reupload = true;
        }
    }
    if (!seg_owner.empty()) ctx->last_segscan_rounds = rounds_max;           // rounds that had something to walk
    if (reupload && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail_all(jda_set_err(ctx, e, "jda_upload_batch(re-upload)"));
    for (int i = 0; i < n; i++) {
        uint32_t nok = 0;
        (void)jda_image_block_index(imgs[i], &nok);
        items[i].d->n_mcus_ok = nok;
        items[i].d->fast_mul = (uint8_t)jda_image_fast_mul(imgs[i]);
        out[i] = items[i].d;
    }
    return JDA_SUCCESS;
}

// The same, tolerant of holes: imgs[i] == NULL (a file jda_prepare_batch rejected) gets out[i] = NULL and
// status[i] = JDA_INVALID_PARAMETER, everybody else is uploaded -- a bad image does not cost the batch its place in the arrays.
int jda_upload_batch_ex(jda_ctx *ctx, int32_t n, jda_image *const *imgs, jda_dev_image **out, int32_t *status)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (n <= 0 || !imgs || !out) return JDA_INVALID_PARAMETER;
    std::vector<jda_image *> v;
    std::vector<int> ix;
    for (int i = 0; i < n; i++) {
        out[i] = NULL;
        if (status) status[i] = imgs[i] ? JDA_SUCCESS : JDA_INVALID_PARAMETER;
        if (imgs[i]) { v.push_back(imgs[i]); ix.push_back(i); }
    }
    if (v.empty()) return JDA_SUCCESS;
    std::vector<jda_dev_image *> o(v.size(), NULL);
    const int rc = jda_upload_batch(ctx, (int32_t)v.size(), v.data(), o.data());
    for (size_t k = 0; k < v.size(); k++) { out[ix[k]] = o[k]; if (status && rc != JDA_SUCCESS) status[ix[k]] = rc; }
    return rc;
}

jda_dev_image *jda_upload(jda_ctx *ctx, jda_image *img, int32_t *err)
{
    int32_t dummy;
    if (!err) err = &dummy;
    if (!ctx || !img) { *err = ctx ? JDA_INVALID_PARAMETER : JDA_ERROR_NO_DEVICE; return NULL; }
    jda_dev_image *d = NULL;
    *err = jda_upload_batch(ctx, 1, &img, &d);
    return d;
}

int jda_dev_image_prescan_on_device(const jda_dev_image *dimg) { return dimg ? dimg->prescan_on_device : 0; }
int jda_last_prescan_rounds(const jda_ctx *ctx) { return ctx ? ctx->last_segscan_rounds : 0; }
uint32_t jda_dev_image_mcus_ok(const jda_dev_image *dimg) { return dimg ? dimg->n_mcus_ok : 0; }

int jda_filter_on_device(jda_ctx *ctx, const uint8_t *raw, int32_t len, uint8_t *out, int32_t *out_len,
                         uint32_t *restart_pos, int32_t restart_cap, int32_t *n_restarts)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (!raw || len < 0 || !out || !out_len) return JDA_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    const size_t raw_cap = align16((size_t)len) + 16, rcap = restart_cap > 0 ? (size_t)restart_cap : 1;
    const size_t off_out = raw_cap, off_rpos = off_out + align16((size_t)len + 16), off_res = off_rpos + align16(rcap * 4), off_work = off_res + 16;
    const size_t off_par = off_work + align16(JDA_FILTER_WORK_BYTES(len));
    uint8_t *d = NULL;
    hipError_t e = hipMalloc((void **)&d, off_par + sizeof(jda_filter_params));
    if (e != hipSuccess) return jda_set_err(ctx, e, "hipMalloc(filter)");
    jda_filter_params P;
    P.raw = d; P.out = d + off_out; P.restart_pos = (uint32_t *)(d + off_rpos); P.result = (uint32_t *)(d + off_res);
    P.raw_len = (uint32_t)len; P.restart_cap = (uint32_t)rcap; P.work = (uint32_t *)(d + off_work); P.raw_skip = 0; P.pad_ = 0;
    e = hipMemsetAsync(d, 0, off_par, ctx->stream);
    if (e == hipSuccess && len) e = hipMemcpyAsync(d, raw, (size_t)len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + off_par, &P, sizeof(P), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = jda_launch_filter((const jda_filter_params *)(d + off_par), 1, (uint32_t)len, ctx->stream);
    uint32_t res[2] = { 0, 0 };
    if (e == hipSuccess) e = hipMemcpyAsync(res, d + off_res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && res[0]) e = hipMemcpy(out, d + off_out, res[0], hipMemcpyDeviceToHost);
    if (e == hipSuccess && restart_pos && restart_cap > 0)
        e = hipMemcpy(restart_pos, d + off_rpos, (size_t)std::min<uint32_t>(res[1] + 1, (uint32_t)restart_cap) * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return jda_set_err(ctx, e, "jda_filter_on_device");
    *out_len = (int32_t)res[0];
    if (n_restarts) *n_restarts = (int32_t)res[1];
    return JDA_SUCCESS;
}
int jda_dev_image_read_index(jda_ctx *ctx, const jda_dev_image *dimg, uint32_t *index, int16_t *dc)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (!dimg) return JDA_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    const size_t nb = (size_t)dimg->info.mcus_x * dimg->info.mcus_y * dimg->info.blocks_per_mcu;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && index) e = hipMemcpy(index, dimg->base + dimg->off_index, (nb + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && dc) e = hipMemcpy(dc, dimg->base + dimg->off_dc, nb * sizeof(int16_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? JDA_SUCCESS : jda_set_err(ctx, e, "jda_dev_image_read_index");
}

void jda_dev_image_free(jda_ctx *ctx, jda_dev_image *dimg)
{
    if (!dimg) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (dimg->base) { if (ctx) jda_pool_free(ctx, dimg->base); else (void)hipFree(dimg->base); }
    free(dimg->tables_host);
    delete dimg;
}

size_t jda_dev_image_bytes(const jda_dev_image *dimg) { return dimg ? dimg->bytes : 0; }

} // extern "C"

int jda_fill_launch_desc(jda_dev_desc &D, const jda_image_info &I, const uint8_t dc_id[3], const uint8_t ac_id[3], const uint8_t q_id[3],
                         int fast_mul, int general_p1, uint32_t n_mcus_ok, uint32_t scan_len, const jda_output &O, int pt_req, int options,
                         int *bpp_out)
{
    const int opt = jda_effective_options(&I, options);      // progressive: 1/8 thumbnail from the DC scan
    memset(&D, 0, sizeof(D));
    if (pt_req < 0 || pt_req > JDA_EIGHT_BIT_GRAYSCALE) return JDA_INVALID_PARAMETER;
    int pt = pt_req;
    if ((opt & JDA_LUMA_ONLY) && pt < JDA_EIGHT_BIT_GRAYSCALE) pt = JDA_EIGHT_BIT_GRAYSCALE;   // jpeg.inl:4991-4993
    int bpp, ow, oh, cw, ch;
    { const int grc = jda_output_geometry(&I, pt, opt, &bpp, &ow, &oh, &cw, &ch); if (grc != JDA_SUCCESS) return grc; }
    D.mode = (uint8_t)jda_mode_of(I);
    D.ncomp = (uint8_t)I.ncomp;
    D.pixel_type = (uint8_t)((D.mode == JDA_MODE_GRAY && pt == JDA_RGB8888) ? JDA_RGB565_BIG_ENDIAN : pt);
    D.scale_shift = (uint8_t)((opt & JDA_SCALE_HALF) ? 1 : (opt & JDA_SCALE_QUARTER) ? 2 : (opt & JDA_SCALE_EIGHTH) ? 3 : 0);
    D.gray_from_color = (uint8_t)(D.mode != JDA_MODE_GRAY && pt == JDA_EIGHT_BIT_GRAYSCALE);
    memcpy(D.dc_id, dc_id, 3); memcpy(D.ac_id, ac_id, 3); memcpy(D.q_id, q_id, 3);
    D.fast_mul = (uint8_t)(fast_mul ? 1 : 0);
    { static const char *dbg = JDA_LAB_ENV("JDA_DEBUG_SKIP"); D.pad_[0] = (uint8_t)((dbg ? (atoi(dbg) & 3) : 0) | jda_desc_stream_bits(I) | (general_p1 ? JDA_DESC_GENERAL_P1 : 0u)); }   // profiling aid: 1 = no P4, 2 = no IDCT
    D.mcus_x = (uint32_t)I.mcus_x; D.mcus_y = (uint32_t)I.mcus_y;
    D.n_mcus_ok = n_mcus_ok; D.scan_len = scan_len;
    D.out = (uint8_t *)O.pixels;
    D.out_pitch = (uint32_t)O.pitch_bytes;
    D.out_w = (uint32_t)(O.width_px < cw ? O.width_px : cw);
    D.out_rows = (uint32_t)(O.rows < ch ? O.rows : ch);
    // the kernels address a surface with 32-bit byte offsets: MCU-padded rows x pitch must stay below 4 GiB
    if (!O.pixels || ((uintptr_t)O.pixels & 15) || (O.pitch_bytes & 15) || O.width_px < 0 || O.rows < 0 || O.pitch_bytes < (int)D.out_w * bpp ||
        (uint64_t)(ch + 16) * (uint64_t)O.pitch_bytes >= (1ull << 32) || O.pitch_bytes >= (1 << 23)) return JDA_INVALID_PARAMETER;
    if (bpp_out) *bpp_out = bpp;
    return JDA_SUCCESS;
}

extern "C" {

jda_batch *jda_batch_create(jda_ctx *ctx, int32_t n, jda_dev_image *const *images,
                            const jda_output *outputs, const int32_t *pixel_types,
                            const int32_t *options, int32_t *err)
{
    return jda_batch_create_rect(ctx, n, images, outputs, pixel_types, options, NULL, err);
}

jda_batch *jda_batch_create_rect(jda_ctx *ctx, int32_t n, jda_dev_image *const *images,
                                 const jda_output *outputs, const int32_t *pixel_types,
                                 const int32_t *options, const int32_t *mcu_rects, int32_t *err)
{
    return jda_batch_create_strips(ctx, n, images, outputs, pixel_types, options, mcu_rects, NULL, err);
}

// strip_mcus[i] != 0: image i's surface is strip-major (jda_dev_desc::strip_mcus; outputs[i] then describes the surface's extent only:
// pitch_bytes x rows >= the strips' bytes)
jda_batch *jda_batch_create_strips(jda_ctx *ctx, int32_t n, jda_dev_image *const *images, const jda_output *outputs, const int32_t *pixel_types,
                                   const int32_t *options, const int32_t *mcu_rects, const int32_t *strip_mcus, int32_t *err)
{
    int32_t dummy;
    if (!err) err = &dummy;
    if (!ctx) { *err = JDA_ERROR_NO_DEVICE; return NULL; }
    if (n <= 0 || !images || !outputs) { *err = JDA_INVALID_PARAMETER; return NULL; }
    std::vector<jda_dev_desc> descs((size_t)n);
    std::vector<jda_strip> strips[JDA_N_LISTS];
    std::vector<int32_t> image_status;
    jda_batch_stats st;
    memset(&st, 0, sizeof(st));
    uint32_t flat_max_items = 0;
    // consecutive images of a list with equal tables and table ids use ONE copy of them (the first one's: it is part of this plan) and are
    // one table generation to the decode kernel -- its workgroups pass from one to the next without restaging (jda_strip::ord)
    int list_owner[JDA_N_LISTS];
    for (int m = 0; m < JDA_N_LISTS; m++) list_owner[m] = -1;
    for (int i = 0; i < n; i++) {
        const jda_dev_image *im = images[i];
        jda_dev_desc &D = descs[(size_t)i];
        if (!im) { memset(&D, 0, sizeof(D)); image_status.push_back(JDA_INVALID_PARAMETER); continue; }   // a hole (rejected file): nothing is launched for it
        const jda_image_info &I = im->info;
        image_status.push_back(im->n_mcus_ok < (uint32_t)(I.mcus_x * I.mcus_y) ? JDA_DECODE_ERROR : JDA_SUCCESS);   // jpeg.inl:5354-5356
        int bpp = 0;
        const int rc = jda_fill_launch_desc(D, I, im->dc_id, im->ac_id, im->q_id, im->fast_mul, im->general_p1, im->n_mcus_ok, im->scan_len,
                                            outputs[i], pixel_types ? pixel_types[i] : JDA_RGB8888, options ? options[i] : 0, &bpp);
        if (rc != JDA_SUCCESS) { *err = rc; return NULL; }
        D.tables = im->base + im->off_tables;
        D.blk_index = (const uint32_t *)(im->base + im->off_index);
        D.blk_dc = (const int16_t *)(im->base + im->off_dc);
        D.scan = im->base + im->off_scan;
        D.strip_mcus = (strip_mcus && strip_mcus[i] > 0) ? (uint32_t)strip_mcus[i] : 0u;
        // kernel variant 1: the plain case -- full size, RGB8888, every block decoded -- runs a kernel in which these
        // descriptor fields are compile-time constants (jda_desc_uniform<1>)
        const int variant = jda_plain_variant(D);
        const int big = D.scale_shift == 3 ? 0 : jda_big_window(D, variant, im->tiles_total, im->tiles_over_small);
        const int cont = jda_use_cont(D, variant, im->n_cont);
        if (cont) { D.blk_cont_first = (const uint32_t *)(im->base + im->off_cont_first); D.blk_cont = (const uint32_t *)(im->base + im->off_cont); }
        {
            const int li = jda_list_index(D, variant, big, cont, mcu_rects == NULL);
            std::vector<jda_strip> &lst = strips[li];
            bool same_tables = false;
            if (list_owner[li] >= 0) {
                const jda_dev_image *om = images[list_owner[li]];
                same_tables = om->tables_host && im->tables_host && !memcmp(om->dc_id, im->dc_id, 3) && !memcmp(om->ac_id, im->ac_id, 3) && !memcmp(om->q_id, im->q_id, 3) &&
                              om->info.ncomp == im->info.ncomp && !memcmp(om->tables_host, im->tables_host, JDA_TABLE_BYTES);
            }
            if (same_tables) D.tables = descs[(size_t)list_owner[li]].tables; else list_owner[li] = i;
            const uint32_t per = jda_mcus_per_tile(D.mode);
            if (JDA_LIST_IS_THUMB_FLAT(li)) {                     // a whole gray image at 1/8: one record (jda_dc_thumbnail_flat)
                jda_strip r;
                memset(&r, 0, sizeof(r));
                r.image = (uint32_t)i; r.count = 1; r.first = 1; r.ord = lst.empty() ? 0u : lst.back().ord + (same_tables ? 0u : 1u);
                lst.push_back(r);
                flat_max_items = std::max(flat_max_items, jda_flat_items(D));
                st.tiles += (int64_t)D.mcus_y * ((D.mcus_x + per - 1) / per);
            } else {
                const size_t before = lst.size();
                jda_append_strips(lst, (uint32_t)i, D.mcus_x, D.mcus_y, D.mode, big, mcu_rects ? mcu_rects + 4 * i : NULL, D.strip_mcus, same_tables);
                for (size_t k = before; k < lst.size(); k++) if (lst[k].count) st.tiles++;
            }
            st.tiles_whole_images += (int64_t)D.mcus_y * ((D.mcus_x + per - 1) / per);
        }
        st.source_pixels += (int64_t)I.width * I.height;
        st.output_bytes += (int64_t)D.out_w * D.out_rows * bpp;
        st.scan_bytes += im->scan_len;
        st.index_bytes += (int64_t)I.mcus_x * I.mcus_y * I.blocks_per_mcu * 6;
        st.table_bytes += JDA_TABLE_BYTES;
    }
    jda_batch *b = new (std::nothrow) jda_batch;
    if (!b) { *err = JDA_ERROR_MEMORY; return NULL; }
    memset(b, 0, sizeof(*b));
    b->n_images = n;
    b->flat_max_items = flat_max_items;
    b->status = new (std::nothrow) int32_t[(size_t)n];
    if (b->status) memcpy(b->status, image_status.data(), (size_t)n * sizeof(int32_t));
    (void)hipSetDevice(ctx->device);
    hipError_t e = jda_pool_alloc(ctx, (void **)&b->d_descs, descs.size() * sizeof(jda_dev_desc));
    if (e == hipSuccess) e = hipMemcpyAsync(b->d_descs, descs.data(), descs.size() * sizeof(jda_dev_desc), hipMemcpyHostToDevice, ctx->stream);
    for (int m = 0; m < JDA_N_LISTS && e == hipSuccess; m++) {
        b->n_strips[m] = (uint32_t)strips[m].size();
        if (!b->n_strips[m]) continue;
        e = jda_pool_alloc(ctx, (void **)&b->d_strips[m], strips[m].size() * sizeof(jda_strip));
        if (e == hipSuccess) e = hipMemcpyAsync(b->d_strips[m], strips[m].data(), strips[m].size() * sizeof(jda_strip), hipMemcpyHostToDevice, ctx->stream);
        st.n_launches++;
        st.n_workgroups += JDA_LIST_IS_THUMB_FLAT(m) ? (int32_t)strips[m].size() : (int32_t)(strips[m].size() / jda_tiles_per_wg(JDA_LIST_MODE(m), JDA_LIST_BIG(m)));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        jda_set_err(ctx, e, "jda_batch_create");
        jda_batch_destroy(ctx, b);
        *err = JDA_ERROR_HIP;
        return NULL;
    }
    b->stats = st;
    *err = JDA_SUCCESS;
    return b;
}

void jda_batch_destroy(jda_ctx *ctx, jda_batch *b)
{
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (b->d_descs) { if (ctx) jda_pool_free(ctx, b->d_descs); else (void)hipFree(b->d_descs); }
    for (int m = 0; m < JDA_N_LISTS; m++) if (b->d_strips[m]) { if (ctx) jda_pool_free(ctx, b->d_strips[m]); else (void)hipFree(b->d_strips[m]); }
    delete[] b->status;
    delete b;
}

int jda_batch_decode(jda_ctx *ctx, jda_batch *b)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (!b) return JDA_INVALID_PARAMETER;
    for (int m = 0; m < JDA_N_LISTS; m++) {
        if (!b->n_strips[m]) continue;
        JDA_HIP(ctx, jda_launch_decode(JDA_LIST_MODE(m), JDA_LIST_FAST(m), JDA_LIST_VARIANT(m), JDA_LIST_BIG(m), JDA_LIST_CONT(m), b->d_descs, b->d_strips[m], b->n_strips[m],
                                       JDA_LIST_IS_THUMB_FLAT(m) ? b->flat_max_items : 0u, ctx->stream));
    }
    return JDA_SUCCESS;
}

// status[i] of every image of the plan: JDA_SUCCESS, JDA_DECODE_ERROR (the stream has a bad MCU: the MCUs before it are decoded,
// as the reference leaves them, jpeg.inl:5354-5356) or JDA_INVALID_PARAMETER (a hole in the image array)
int jda_batch_get_status(const jda_batch *b, int32_t *status)
{
    if (!b || !status || !b->status) return JDA_INVALID_PARAMETER;
    memcpy(status, b->status, (size_t)b->n_images * sizeof(int32_t));
    return JDA_SUCCESS;
}

int jda_batch_get_stats(const jda_batch *b, jda_batch_stats *stats)
{
    if (!b || !stats) return JDA_INVALID_PARAMETER;
    *stats = b->stats;
    return JDA_SUCCESS;
}

// checksums[i] of n surfaces (device pointers): see jda_surface_checksum in jda_kernels.hip.  Synchronous.
int jda_checksum_surfaces(jda_ctx *ctx, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *checksums)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (n <= 0 || !surfaces || !row_bytes || !checksums) return JDA_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    unsigned long long *d = NULL;
    hipError_t e = jda_pool_alloc(ctx, (void **)&d, (size_t)n * 8);
    if (e != hipSuccess) return jda_set_err(ctx, e, "hipMalloc(checksums)");
    e = hipMemsetAsync(d, 0, (size_t)n * 8, ctx->stream);
    for (int i = 0; i < n && e == hipSuccess; i++) {
        if (!surfaces[i].pixels || surfaces[i].pitch_bytes < row_bytes[i] || row_bytes[i] < 0 || surfaces[i].rows < 0) { jda_pool_free(ctx, d); return JDA_INVALID_PARAMETER; }
        e = jda_launch_checksum(surfaces[i].pixels, (uint32_t)surfaces[i].pitch_bytes, (uint32_t)row_bytes[i], (uint32_t)surfaces[i].rows, d + i, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(checksums, d, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    jda_pool_free(ctx, d);
    return e == hipSuccess ? JDA_SUCCESS : jda_set_err(ctx, e, "jda_checksum_surfaces");
}

// "0000:8e:00.0" of the context's GPU (for NUMA placement of the host threads that feed it); buf >= 16 bytes
int jda_device_pci_bus_id_of(int32_t device, char *buf, int32_t len)
{
    if (!buf || len < 16) return JDA_INVALID_PARAMETER;
    return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? JDA_SUCCESS : JDA_ERROR_HIP;
}
int jda_device_pci_bus_id(jda_ctx *ctx, char *buf, int32_t len)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (!buf || len < 16) return JDA_INVALID_PARAMETER;
    return hipDeviceGetPCIBusId(buf, len, ctx->device) == hipSuccess ? JDA_SUCCESS : JDA_ERROR_HIP;
}

int jda_sync(jda_ctx *ctx)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDA_SUCCESS;
}

int jda_timer_start(jda_ctx *ctx)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return JDA_SUCCESS;
}

int jda_timer_stop(jda_ctx *ctx)
{
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    JDA_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    return JDA_SUCCESS;
}

double jda_timer_elapsed_ms(jda_ctx *ctx)
{
    if (!ctx) return -1.0;
    if (hipEventSynchronize(ctx->ev_stop) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop) != hipSuccess) return -1.0;
    return (double)ms;
}

int jda_decode_to_host(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type,
                       int32_t options, void *host_pixels, int32_t pitch_bytes, int32_t rows)
{
    return jda_decode_to_host_ex(ctx, jpeg, len, pixel_type, options, host_pixels, pitch_bytes, rows, NULL);
}

int jda_decode_to_host_ex(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type,
                          int32_t options, void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded)
{
    return jda_decode_to_host_rect(ctx, jpeg, len, pixel_type, options, NULL, host_pixels, pitch_bytes, rows, mcus_decoded, NULL);
}

int jda_decode_to_host_rect(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                            void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles)
{
    return jda_decode_to_host_flags(ctx, jpeg, len, pixel_type, options, mcu_rect, host_pixels, pitch_bytes, rows, mcus_decoded, tiles, 0);
}

int jda_decode_to_host_flags(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags)
{
    return jda_decode_to_host_bands(ctx, jpeg, len, pixel_type, options, mcu_rect, host_pixels, pitch_bytes, rows, mcus_decoded, tiles, flags, 1, NULL, NULL);
}

// Who makes the index of ONE image (jda_decode_to_host*, the class).
static int32_t jda_onecall_prepare_flags(int32_t len)
{
    // One image at a time, the device pre-scan is seven latency-bound launches of ~0.1-0.15 ms each whatever the size (a 640x480 scan
    // is four wavefronts' worth of segments).  The host pre-scan costs 8 us per KB of file on one thread and 2.3 on six around one L3
    // (jda_frontend.cpp, host_prescan_chunks / _intervals): measured end to end on the GPU box (six threads), 1920x1080 (214 KB) 0.65 ms
    // with the host's index against 0.80 with the device's, 2560x1440 (377 KB) 1.08 against 0.90, 4096x4096 4.5 against 1.44 -- the
    // device from 256 KB on; from 128 KB where the host has fewer threads (one: 6.8 us per KB, 1920x1080 1.95 ms).  In batches the
    // device always does it.
    static const int32_t dev_from = []() {
        const char *e = JDA_LAB_ENV("JDA_ONECALL_DEVICE_PRESCAN_BYTES");   // (for measuring the crossover)
        return e ? atoi(e) : ((jda_host_prescan_threads() >= 6 ? 256 : 128) << 10);
    }();
    return len >= dev_from ? JDA_PREPARE_DEVICE_PRESCAN : 0;
}

// The whole image as the reference's JPEGDRAW strips (jpeg.inl:5300-5336): strip_mcus MCUs wide, one MCU row high, in raster order, every
// strip's pixels contiguous -- what JPEGDEC::decode hands to the draw callback without touching a pixel (a strip of a row-major canvas is
// a few hundred bytes from each of 16 rows a pitch apart: 8,192 strips of a 4096x4096 image were 1.5 ms of small strided copies).
int jda_decode_to_host_strips(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, int32_t strip_mcus,
                              void *host_pixels, size_t host_bytes, int32_t *mcus_decoded, int32_t n_bands, jda_band_callback *band_ready, void *user)
{
    if (mcus_decoded) *mcus_decoded = 0;
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    if (!jpeg || !host_pixels || strip_mcus <= 0) return JDA_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    int32_t err = JDA_SUCCESS;
    jda_image *img = jda_prepare_ex(jpeg, len, jda_onecall_prepare_flags(len), &err);
    if (!img) return err;
    const jda_image_info I = *jda_image_get_info(img);
    int bpp, ow, oh, cw, ch;
    int rc = jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) { jda_image_free(img); return rc; }
    const int mw = cw / I.mcus_x, mh = ch / I.mcus_y;
    const int n_sx = (I.mcus_x + strip_mcus - 1) / strip_mcus;
    const size_t strip_bytes = (size_t)strip_mcus * mw * mh * bpp, row_bytes = (size_t)n_sx * strip_bytes, total = row_bytes * I.mcus_y;
    if (host_bytes < total) { jda_image_free(img); return JDA_INVALID_PARAMETER; }
    jda_dev_image *dimg = jda_upload(ctx, img, &err);
    uint32_t nok = 0;
    jda_image_block_index(img, &nok);
    const bool complete = nok == (uint32_t)(I.mcus_x * I.mcus_y);
    if (mcus_decoded) *mcus_decoded = (int32_t)nok;
    jda_image_free(img);
    if (!dimg) return err;
    void *dout = NULL;
    const int dpitch = (int)align16((size_t)cw * bpp);
    const size_t surf = std::max(total, (size_t)dpitch * ch) + 256;
    if (jda_pool_alloc(ctx, &dout, surf) != hipSuccess) { jda_dev_image_free(ctx, dimg); return JDA_ERROR_MEMORY; }
    jda_output O;
    O.pixels = dout; O.pitch_bytes = dpitch; O.width_px = cw; O.rows = ch;
    jda_batch *b = jda_batch_create_strips(ctx, 1, &dimg, &O, &pixel_type, &options, NULL, &strip_mcus, &err);
    rc = err;
    if (b) {
        if (!complete) (void)hipMemsetAsync(dout, 0, total, ctx->stream);       // (callback mode: what a bad stream does not reach reads zero)
        rc = jda_batch_decode(ctx, b);
        if (rc == JDA_SUCCESS) {
            int nb = (band_ready && n_bands > 1) ? (n_bands > JDA_MAX_BANDS ? JDA_MAX_BANDS : n_bands) : 1;
            if (nb > I.mcus_y) nb = I.mcus_y;
            const int per = (I.mcus_y + nb - 1) / nb;                            // MCU rows a band
            hipError_t e = hipSuccess;
            int made = 0;
            for (int k = 0; k < nb && e == hipSuccess; k++) {
                const int y0 = k * per, y1 = std::min(I.mcus_y, y0 + per);
                if (y0 >= y1) break;
                if (!ctx->ev_band[k]) e = hipEventCreateWithFlags(&ctx->ev_band[k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipMemcpyAsync((uint8_t *)host_pixels + (size_t)y0 * row_bytes, (uint8_t *)dout + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipEventRecord(ctx->ev_band[k], ctx->stream);
                if (e == hipSuccess) made++;
            }
            for (int k = 0; k < made && e == hipSuccess; k++) {
                e = hipEventSynchronize(ctx->ev_band[k]);
                if (e == hipSuccess && band_ready) (*band_ready)(user, k * per * mh, std::min(I.mcus_y, (k + 1) * per) * mh);
            }
            { const hipError_t es = hipStreamSynchronize(ctx->stream); if (e == hipSuccess) e = es; }
            if (e != hipSuccess) rc = jda_set_err(ctx, e, "copy back");
        }
        jda_batch_destroy(ctx, b);
    }
    jda_pool_free(ctx, dout);
    jda_dev_image_free(ctx, dimg);
    if (rc == JDA_SUCCESS && !complete) rc = JDA_DECODE_ERROR;   // jpeg.inl:5354-5356
    return rc;
}

int jda_decode_to_host_bands(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags,
                             int32_t n_bands, jda_band_callback *band_ready, void *user)
{
    if (mcus_decoded) *mcus_decoded = 0;
    if (tiles) tiles[0] = tiles[1] = 0;
    if (!ctx) return JDA_ERROR_NO_DEVICE;
    int32_t err = JDA_SUCCESS;
    static const bool trace = JDA_LAB_ENV("JDA_ONECALL_TRACE") != NULL;       // stage timings on stderr (diagnostics)
    double t_mark = trace ? now_ms() : 0.0;
#define JDA_OC_MARK(what) do { if (trace) { const double t_ = now_ms(); fprintf(stderr, "jda_decode_to_host: %-24s %7.3f ms\n", what, t_ - t_mark); t_mark = t_; } } while (0)
    const int32_t prep_flags = jda_onecall_prepare_flags(len);
    jda_image *img = jda_prepare_ex(jpeg, len, prep_flags, &err);
    if (!img) return err;
    const jda_image_info I = *jda_image_get_info(img);       // (by value: the image is freed as soon as it is uploaded)
    int bpp, ow, oh, cw, ch;
    int rc = jda_output_geometry(&I, pixel_type, options, &bpp, &ow, &oh, &cw, &ch);
    if (rc != JDA_SUCCESS) { jda_image_free(img); return rc; }
    const int dpitch = (int)align16((size_t)cw * bpp);
    const int drows = rows < ch ? rows : ch;
    JDA_OC_MARK("prepare (host)");
    jda_dev_image *dimg = jda_upload(ctx, img, &err);
    JDA_OC_MARK("upload + device pre-scan");
    uint32_t nok = 0;
    jda_image_block_index(img, &nok);                   // (after the upload: a deferred pre-scan has run by now)
    const bool complete = nok == (uint32_t)(I.mcus_x * I.mcus_y);
    if (mcus_decoded) *mcus_decoded = (int32_t)nok;
    jda_image_free(img);
    if (!dimg) return err;
    void *dout = NULL;
    if (jda_pool_alloc(ctx, &dout, (size_t)dpitch * ch) != hipSuccess) dout = NULL;
    if (!dout) { jda_dev_image_free(ctx, dimg); return JDA_ERROR_MEMORY; }
    jda_output O;
    O.pixels = dout; O.pitch_bytes = dpitch; O.width_px = cw; O.rows = drows;
    jda_batch *b = jda_batch_create_rect(ctx, 1, &dimg, &O, &pixel_type, &options, mcu_rect, &err);
    rc = err;
    JDA_OC_MARK("surface + launch plan");
    if (b) {
        if (tiles) { tiles[0] = (int32_t)b->stats.tiles; tiles[1] = (int32_t)b->stats.tiles_whole_images; }
        // only the MCU rows of the rectangle are decoded, zeroed where nothing is written, and copied back
        int r0 = 0, r1 = drows;
        if (mcu_rect) {
            const int mh_out = ch / (I.mcus_y ? I.mcus_y : 1);
            r0 = std::max(0, std::min(drows, mcu_rect[1] * mh_out)); r1 = std::max(r0, std::min(drows, mcu_rect[3] * mh_out));
        }
        if ((!complete || mcu_rect) && r1 > r0) (void)hipMemsetAsync((uint8_t *)dout + (size_t)r0 * dpitch, 0, (size_t)dpitch * (r1 - r0), ctx->stream);
        rc = jda_batch_decode(ctx, b);
        if (rc == JDA_SUCCESS && !complete && (flags & JDA_TO_HOST_KEEP_UNDECODED) && !mcu_rect) {
            // the reference returns at the bad MCU and leaves the rest of the caller's buffer alone (jpeg.inl:5354-5356): copy back the
            // whole MCU rows in front of it and, of its own row, the MCUs in front of it -- nothing else of the host buffer is touched
            const int mh_out = ch / (I.mcus_y ? I.mcus_y : 1), mw_out = cw / (I.mcus_x ? I.mcus_x : 1);
            const int full = (int)(nok / (uint32_t)I.mcus_x), part = (int)(nok % (uint32_t)I.mcus_x);
            const int rf = std::min(drows, full * mh_out), rp = std::min(drows, (full + 1) * mh_out);
            const size_t row_bytes = (size_t)cw * bpp < (size_t)pitch_bytes ? (size_t)cw * bpp : (size_t)pitch_bytes;
            hipError_t e = hipSuccess;
            if (rf > 0) e = hipMemcpy2DAsync(host_pixels, (size_t)pitch_bytes, dout, (size_t)dpitch, row_bytes, (size_t)rf, hipMemcpyDeviceToHost, ctx->stream);
            const size_t part_bytes = std::min(row_bytes, (size_t)part * mw_out * bpp);
            if (e == hipSuccess && part_bytes && rp > rf)
                e = hipMemcpy2DAsync((uint8_t *)host_pixels + (size_t)rf * pitch_bytes, (size_t)pitch_bytes, (uint8_t *)dout + (size_t)rf * dpitch, (size_t)dpitch, part_bytes, (size_t)(rp - rf), hipMemcpyDeviceToHost, ctx->stream);
            { const hipError_t es = hipStreamSynchronize(ctx->stream); if (e == hipSuccess) e = es; }
            if (e != hipSuccess) rc = jda_set_err(ctx, e, "copy back");
        } else if (rc == JDA_SUCCESS && r1 > r0 && band_ready && n_bands > 1) {
            // the copy back in bands of whole MCU rows, the caller told as each one lands: what it does with band k (the class replays its
            // draw callbacks) runs while band k + 1 is still on the bus
            const int mh_out = ch / (I.mcus_y ? I.mcus_y : 1);
            int nb = n_bands > JDA_MAX_BANDS ? JDA_MAX_BANDS : n_bands;
            const int mrows = (r1 - r0 + mh_out - 1) / mh_out;
            if (nb > mrows) nb = mrows;
            const int per = ((mrows + nb - 1) / nb) * mh_out;
            const size_t row_bytes = (size_t)cw * bpp < (size_t)pitch_bytes ? (size_t)cw * bpp : (size_t)pitch_bytes;
            hipError_t e = hipSuccess;
            int made = 0;
            for (int k = 0; k < nb && e == hipSuccess; k++) {
                const int b0 = r0 + k * per, b1 = std::min(r1, b0 + per);
                if (b0 >= b1) break;
                if (!ctx->ev_band[k]) e = hipEventCreateWithFlags(&ctx->ev_band[k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipMemcpy2DAsync((uint8_t *)host_pixels + (size_t)b0 * pitch_bytes, (size_t)pitch_bytes, (uint8_t *)dout + (size_t)b0 * dpitch, (size_t)dpitch, row_bytes, (size_t)(b1 - b0), hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipEventRecord(ctx->ev_band[k], ctx->stream);
                if (e == hipSuccess) made++;
            }
            for (int k = 0; k < made && e == hipSuccess; k++) {
                e = hipEventSynchronize(ctx->ev_band[k]);
                if (e == hipSuccess) (*band_ready)(user, r0 + k * per, std::min(r1, r0 + (k + 1) * per));
            }
            // (also after an error: copies queued before it may still be writing the caller's buffer, and the surface goes back to the pool)
            { const hipError_t es = hipStreamSynchronize(ctx->stream); if (e == hipSuccess) e = es; }
            if (e != hipSuccess) rc = jda_set_err(ctx, e, "copy back");
        } else if (rc == JDA_SUCCESS && r1 > r0) {
            const size_t row_bytes = (size_t)cw * bpp < (size_t)pitch_bytes ? (size_t)cw * bpp : (size_t)pitch_bytes;
            hipError_t e = hipMemcpy2DAsync((uint8_t *)host_pixels + (size_t)r0 * pitch_bytes, (size_t)pitch_bytes, (uint8_t *)dout + (size_t)r0 * dpitch, (size_t)dpitch, row_bytes, (size_t)(r1 - r0), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = jda_set_err(ctx, e, "copy back");
        }
        JDA_OC_MARK("decode + copy back");
        jda_batch_destroy(ctx, b);
    }
    jda_pool_free(ctx, dout);
    jda_dev_image_free(ctx, dimg);
    JDA_OC_MARK("release");
#undef JDA_OC_MARK
    if (rc == JDA_SUCCESS && !complete) rc = JDA_DECODE_ERROR;   // jpeg.inl:5354-5356
    return rc;
}

} // extern "C"
