// jda_pipeline.cpp -- the whole path as one streamed pipeline: files in, pixels resident in HBM out.
//
//   host   parse + Huffman LUTs + prescaled quantisers per file (a few microseconds each, on a small thread pool)
//   H2D    the UNFILTERED entropy-coded bytes of every file + one control blob (tables, kernel parameters, launch plan)
//   GPU    marker / stuffing filter (JPEGFilter, jpeg.inl:1431-1540)            jda_filter_count / _carry / _write
//          per-block index = the serial pre-scan's, entry for entry              jda_segscan_fused / _tail / _sums / _write
//          the MCU loops of DecodeJPEG (jpeg.inl:5109-5353)                      jda_decode_tiles_persistent
// The host never touches a compressed byte.  Upload + filter + pre-scan of batch n+1 run on their own stream under the
// decode of batch n; every batch lives in one device arena per pipeline slot (no allocation per image, no host
// synchronisation inside a batch).  The decode is launched OPTIMISTICALLY -- every MCU valid, 24-bit multiplies where the
// quantisers allow them for every legal stream -- and the pre-scan's own verdict (bad code, marker out of place, states not
// settled, a magnitude beyond the legal categories) is read when the batch is waited for: an image that fails it, or that
// the device walk cannot take (progressive, one restart interval, odd table ids), is redone through the serial host
// pre-scan (jda_prepare), which reproduces what the reference does with such streams.  A bad image never poisons its
// batch: every image has its own status.
//
// There is NO CPU decode fallback here either: the slow path is the same kernels behind the host pre-scan.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#include "jda_runtime_internal.h"

extern "C" int jda_front_prepare(const uint8_t *jpeg, int32_t len, uint8_t *tables, jda_front *out);
extern "C" uint32_t jda_front_fast_mul(const uint8_t *tables, const jda_front *f, uint32_t max_ac_bits, int32_t max_abs_dc);

#if defined(__x86_64__) || defined(_M_X64)
#include <emmintrin.h>
#endif

namespace {

// A file's entropy-coded bytes into the page-locked mirror: written once, read next by the copy engine, never by this core again --
// streaming stores (no read for ownership, nothing of the files evicted from the caches for it); the short ends by memcpy.
inline void copy_to_mirror(uint8_t *dst, const uint8_t *src, size_t n)
{
#if defined(__x86_64__) || defined(_M_X64)
    static const bool plain = JDA_LAB_ENV("JDA_PIPE_PLAIN_COPY") != NULL;      // (measuring)
    if (n >= 4096 && !plain) {
        const size_t head = (size_t)(-(intptr_t)dst) & 15u;
        memcpy(dst, src, head);
        dst += head; src += head; n -= head;
        const size_t body = n & ~(size_t)63;
        for (size_t i = 0; i < body; i += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
            _mm_stream_si128((__m128i *)(dst + i), a); _mm_stream_si128((__m128i *)(dst + i + 16), b);
            _mm_stream_si128((__m128i *)(dst + i + 32), c); _mm_stream_si128((__m128i *)(dst + i + 48), d);
        }
        _mm_sfence();
        memcpy(dst + body, src + body, n - body);
        return;
    }
#endif
    memcpy(dst, src, n);
}

inline size_t a16(size_t v) { return (v + 15) & ~(size_t)15; }
inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

// a few persistent worker threads: run(n, fn) calls fn(0..n-1) across them and the caller.  Items are handed out under the
// mutex, a few at a time (an item is a few to tens of microseconds of work), tagged with the run's generation so that a
// worker that is late leaving one run cannot take an item of the next one twice.
class Workers {
public:
    explicit Workers(int n_threads) : stop_(false), gen_(0), fn_(nullptr), n_(0), next_(0), done_(0)
    {
        for (int t = 1; t < n_threads; t++) th_.emplace_back([this]() { loop(); });
    }
    ~Workers()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int threads() const { return (int)th_.size() + 1; }
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
        uint64_t g;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; next_ = 0; done_ = 0; g = ++gen_;
        }
        cv_.notify_all();
        work(g);
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [this]() { return done_ >= n_; });
        fn_ = nullptr; n_ = 0;
    }

private:
    void work(uint64_t g)
    {
        for (;;) {
            int i, k;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (gen_ != g || next_ >= n_) return;
                // (a run of hundreds of small files: a few items per visit to the mutex -- a quarter of an even share, so that the
                // threads still finish together)
                k = n_ / (4 * ((int)th_.size() + 1));
                k = k < 1 ? 1 : (k > n_ - next_ ? n_ - next_ : k);
                i = next_; next_ += k; f = fn_;
            }
            for (int j = 0; j < k; j++) (*f)(i + j);
            std::lock_guard<std::mutex> lk(m_);
            if (gen_ == g && (done_ += k) >= n_) cv_done_.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return gen_ != seen; });
                seen = g = gen_;
                if (stop_) return;
            }
            work(g);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    bool stop_;
    uint64_t gen_;
    const std::function<void(int)> *fn_;
    int n_, next_, done_;
};

struct Img {
    jda_front f;
    int32_t err;                 // front-end verdict
    bool device;                 // filter + pre-scan + decode were enqueued for it
    bool fast;                   // launched with 24-bit multiplies
    uint32_t n_segs_ub, n_blocks;
    size_t off_raw, off_scan, scan_bytes, off_index, off_dc, off_work, off_zero, off_stats;   // arena offsets
    size_t work_bytes, zero_bytes, off_rpos, off_fwork, off_wt;     // off_rpos / off_fwork: restart positions / the filter's work words inside the work region
    bool direct;                 // its bytes go to the GPU from where the caller has them (JDA_SUBMIT_PINNED_INPUT): no copy into the mirror
    uint32_t raw_skip;           // .. keeping their alignment: the stream's first byte is raw_skip bytes behind the 16-byte aligned off_raw
    bool record;                 // the pre-scan runs in RECORD mode (no WRITE walk)
    size_t off_recs, off_cands;  // block records / truncation candidates inside the work region
    uint32_t cand_cap;
    size_t ctl_tables;           // offset of its tables inside the control blob (images with equal tables and table ids share one copy)
    uint64_t tab_hash;           // of the tables + the components' table ids (what the walk's tables are made from)
    int tab_owner;               // the first image of the batch with the same tables (itself: it owns the copy that travels)
    uint32_t list, n_tiles;      // launch list it is in, tiles (padded)
    size_t strip_off;            // its first strip inside the list
    uint32_t ord;
};

} // namespace

#define JDA_PIPE_MAX_DEPTH 8
#ifndef JDA_PIPE_SPEC_ROUNDS
#define JDA_PIPE_SPEC_ROUNDS 4       // rounds launched one by one (a round with an empty work list returns at once); the rest in one launch (jda_segscan_tail)
#endif
#define JDA_PIPE_MAX_ROUNDS 56       // stats[8 + r] = length of round r's list, r <= 57 < 60
#define JDA_PIPE_STATS_BYTES 288      // per image: filter result (2 words, 16 bytes) | 64 result words of the pre-scan + 16 bytes

struct jda_pipeline {
    jda_ctx *ctx;
    jda_ctx *slow_ctx;                   // the redo path's own context (made when the first image needs it)
    int depth, max_images;
    hipStream_t s_up, s_copy;            // pre-scan | memset + H2D + filter (its own stream: a copy must not queue behind the previous batch's pre-scan)
    hipStream_t s_upx[2];                // more pre-scan streams, taken in turn with s_up: a batch's late rounds (a handful of wavefronts, the
    int n_upx;                           // latency of a walk each) must not hold back the next batch's round 0 ($JDA_PIPE_UP_STREAMS: 1 .. 3 in all)
    Workers *workers;
    struct Slot {
        int ticket;
        bool in_flight;
        uint8_t *dev; size_t dev_cap;
        uint8_t *pin; size_t pin_cap;          // control blob (H2D) followed by the statistics read back (D2H)
        hipEvent_t ev_copy, ev_up, ev_dec;
        std::vector<Img> imgs;
        std::vector<const uint8_t *> jpegs; std::vector<int32_t> lens, pts, opts; std::vector<jda_output> outs;
        size_t ctl_bytes, off_stats_dev, stats_bytes, pin_stats;
        size_t list_off[JDA_N_LISTS]; uint32_t list_n[JDA_N_LISTS];
        size_t off_descs;
        uint32_t flat_max_items;             // JDA_LIST_THUMB_FLAT: the launch's width
        jda_pipeline_stats st;
    } slots[JDA_PIPE_MAX_DEPTH];
    int next_ticket;
    jda_pipeline_stats total;
};

static void slot_free(jda_pipeline::Slot &s)
{
    if (s.dev) (void)hipFree(s.dev);
    if (s.pin) (void)hipHostFree(s.pin);
    s.dev = NULL; s.pin = NULL; s.dev_cap = s.pin_cap = 0;
}

// tiles of one image, padded to whole workgroups (the same list jda_append_strips makes)
static uint32_t count_tiles(uint32_t mcus_x, uint32_t mcus_y, int mode, int big)
{
    const uint32_t per = jda_mcus_per_tile(mode), wg = jda_tiles_per_wg(mode, big);
    const uint32_t n = mcus_y * ((mcus_x + per - 1) / per);
    return (n + wg - 1) / wg * wg;
}
extern "C" {

jda_pipeline *jda_pipeline_create(jda_ctx *ctx, int32_t max_images, int32_t depth, int32_t host_threads, int32_t *err)
{
    int32_t dummy;
    if (!err) err = &dummy;
    if (!ctx) { *err = JDA_ERROR_NO_DEVICE; return NULL; }
    if (max_images <= 0 || depth < 1 || depth > JDA_PIPE_MAX_DEPTH) { *err = JDA_INVALID_PARAMETER; return NULL; }
    jda_pipeline *p = new (std::nothrow) jda_pipeline;
    if (!p) { *err = JDA_ERROR_MEMORY; return NULL; }
    p->ctx = ctx; p->slow_ctx = NULL; p->depth = depth; p->max_images = max_images; p->next_ticket = 0; p->workers = NULL; p->s_up = NULL; p->s_upx[0] = p->s_upx[1] = NULL; p->n_upx = 0; p->s_copy = NULL;
    memset(&p->total, 0, sizeof(p->total));
    (void)hipSetDevice(ctx->device);
    bool ok = hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking) == hipSuccess;
    {
        const char *e = JDA_LAB_ENV("JDA_PIPE_UP_STREAMS");
        int want = e ? atoi(e) : 2;
        if (want > depth) want = depth;
        for (int i = 1; i < want && i < 3 && ok; i++) { ok = hipStreamCreateWithFlags(&p->s_upx[i - 1], hipStreamNonBlocking) == hipSuccess; if (ok) p->n_upx = i; }
    }
    for (int i = 0; i < JDA_PIPE_MAX_DEPTH; i++) {
        jda_pipeline::Slot &s = p->slots[i];
        s.ticket = -1; s.in_flight = false; s.dev = NULL; s.pin = NULL; s.dev_cap = s.pin_cap = 0; s.ev_copy = s.ev_up = s.ev_dec = NULL;
        if (i < depth && ok) ok = hipEventCreateWithFlags(&s.ev_copy, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.ev_dec, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { jda_pipeline_destroy(p); *err = JDA_ERROR_HIP; return NULL; }
    unsigned nt = host_threads > 0 ? (unsigned)host_threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    p->workers = new (std::nothrow) Workers((int)nt);
    if (!p->workers) { jda_pipeline_destroy(p); *err = JDA_ERROR_MEMORY; return NULL; }
    *err = JDA_SUCCESS;
    return p;
}

void jda_pipeline_destroy(jda_pipeline *p)
{
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    if (p->s_copy) (void)hipStreamSynchronize(p->s_copy);
    if (p->s_up) (void)hipStreamSynchronize(p->s_up);
    for (int i = 0; i < 2; i++) if (p->s_upx[i]) (void)hipStreamSynchronize(p->s_upx[i]);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (int i = 0; i < JDA_PIPE_MAX_DEPTH; i++) {
        jda_pipeline::Slot &s = p->slots[i];
        slot_free(s);
        if (s.ev_copy) (void)hipEventDestroy(s.ev_copy);
        if (s.ev_up) (void)hipEventDestroy(s.ev_up);
        if (s.ev_dec) (void)hipEventDestroy(s.ev_dec);
    }
    if (p->s_up) (void)hipStreamDestroy(p->s_up);
    for (int i = 0; i < 2; i++) if (p->s_upx[i]) (void)hipStreamDestroy(p->s_upx[i]);
    if (p->s_copy) (void)hipStreamDestroy(p->s_copy);
    delete p->workers;
    if (p->slow_ctx) jda_destroy(p->slow_ctx);
    delete p;
}

// (measuring: JDA_PIPE_TIME=1 prints where jda_pipeline_submit's time goes, summed over the submits, when the process ends)
struct jda_submit_clock {
    bool on; double acc[8]; long n; std::chrono::steady_clock::time_point t;
    jda_submit_clock() : on(JDA_LAB_ENV("JDA_PIPE_TIME") != NULL), n(0) { for (double &a : acc) a = 0; }
    ~jda_submit_clock() { if (on && n) fprintf(stderr, "jda_pipeline_submit x %ld: setup %.1f us, parse + tables %.1f, layout %.1f, grow buffers %.1f, parameters %.1f, strips + copy into the page-locked mirror %.1f, enqueue %.1f (per submit)\n", n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n); }
    long calls = 0;                                                 // (the first eight submits warm buffers and pages up: not counted)
    void start() { if (on) { t = std::chrono::steady_clock::now(); calls++; if (calls > 8) n++; } }
    void lap(int k) { if (on) { const auto u = std::chrono::steady_clock::now(); if (calls > 8) acc[k] += std::chrono::duration<double, std::micro>(u - t).count(); t = u; } }
};
static jda_submit_clock g_submit_clock;

int jda_pipeline_submit(jda_pipeline *p, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                        const int32_t *pixel_types, const int32_t *options, int32_t *ticket)
{
    return jda_pipeline_submit_ex(p, n, jpegs, lens, outputs, pixel_types, options, 0, ticket);
}

// Files below this size still go through the page-locked mirror when the input is page-locked already: a copy command per file costs
// the submitting thread a few microseconds, which for a 100 KB file is more than the workers' copy of it
#ifndef JDA_PIPE_DIRECT_MIN_BYTES
#define JDA_PIPE_DIRECT_MIN_BYTES (128u << 10)
#endif

int jda_pipeline_submit_ex(jda_pipeline *p, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                           const int32_t *pixel_types, const int32_t *options, int32_t flags, int32_t *ticket)
{
    if (!p) return JDA_ERROR_NO_DEVICE;
    g_submit_clock.start();
    if (n <= 0 || n > p->max_images || !jpegs || !lens || !outputs || !ticket) return JDA_INVALID_PARAMETER;
    jda_ctx *ctx = p->ctx;
    (void)hipSetDevice(ctx->device);
    const int t = p->next_ticket;
    jda_pipeline::Slot &S = p->slots[t % p->depth];
    if (S.in_flight) return JDA_INVALID_PARAMETER;          // the batch that used this slot has not been waited for
    S.imgs.assign((size_t)n, Img());
    S.jpegs.assign(jpegs, jpegs + n); S.lens.assign(lens, lens + n); S.outs.assign(outputs, outputs + n);
    S.pts.resize((size_t)n); S.opts.resize((size_t)n);
    for (int i = 0; i < n; i++) { S.pts[(size_t)i] = pixel_types ? pixel_types[i] : JDA_RGB8888; S.opts[(size_t)i] = options ? options[i] : 0; }
    memset(&S.st, 0, sizeof(S.st));
    S.st.images = n;
    S.flat_max_items = 0;

    // ---- control blob layout, part 1: the tables (the workers write them straight into the page-locked buffer)
    const size_t tab_stride = a16(JDA_TABLE_BYTES);
    size_t ctl = 0;
    for (int i = 0; i < n; i++) { S.imgs[(size_t)i].ctl_tables = ctl; ctl += tab_stride; }
    // the page-locked buffer must hold the tables before the strips are counted: size it generously for them now
    size_t pin_need_min = ctl + a16((size_t)n * sizeof(jda_filter_params)) + a16((size_t)n * sizeof(jda_segscan_params)) + a16((size_t)n * sizeof(jda_dev_desc)) +
                          a16((size_t)n * sizeof(jda_strips_params)) + (size_t)n * JDA_PIPE_STATS_BYTES + 4096;
    if (S.pin_cap < pin_need_min) {
        if (S.pin) (void)hipHostFree(S.pin);
        S.pin = NULL; S.pin_cap = 0;
        const size_t want = a256(pin_need_min * 2 + ((size_t)8 << 20));
        if (hipHostMalloc((void **)&S.pin, want, hipHostMallocDefault) != hipSuccess) { S.pin = NULL; return JDA_ERROR_MEMORY; }
        S.pin_cap = want;
    }

    g_submit_clock.lap(0);
    // ---- host: parse + tables, in parallel
    p->workers->run(n, [&](int i) {
        Img &im = S.imgs[(size_t)i];
        im.err = (jpegs[i] && lens[i] > 0) ? jda_front_prepare(jpegs[i], lens[i], S.pin + im.ctl_tables, &im.f) : JDA_INVALID_PARAMETER;
        im.tab_owner = i; im.tab_hash = 0;
        if (im.err == JDA_SUCCESS) {                          // FNV-1a over the tables (8 bytes a step) and the table ids
            uint64_t h = 0xcbf29ce484222325ull;
            const uint64_t *w = (const uint64_t *)(S.pin + im.ctl_tables);
            for (size_t k = 0; k < JDA_TABLE_BYTES / 8; k++) h = (h ^ w[k]) * 0x100000001b3ull;
            for (int c = 0; c < 3; c++) h = (h ^ (uint64_t)(im.f.dc_id[c] | im.f.ac_id[c] << 8 | im.f.q_id[c] << 16)) * 0x100000001b3ull;
            im.tab_hash = h ^ (uint64_t)im.f.info.ncomp;
        }
    });
    // Images with the same tables (a camera's, an encoder's: most batches have one or two sets) share ONE copy of them: 10.8 KB less
    // over the bus per image (a tenth of a 1280x720 file, a quarter of a 640x480 one), one set of walk tables to build and to keep in the
    // L2 instead of one per image.  The owners' copies are moved together at the front of the blob.
    static_assert(JDA_TABLE_BYTES % 8 == 0, "hashed 8 bytes a step");
    size_t n_tab = 0;
    {
        auto same_tables = [&](const Img &om, const Img &im) {
            return !memcmp(S.pin + om.ctl_tables, S.pin + im.ctl_tables, JDA_TABLE_BYTES) && !memcmp(om.f.dc_id, im.f.dc_id, 3) && !memcmp(om.f.ac_id, im.f.ac_id, 3) &&
                   !memcmp(om.f.q_id, im.f.q_id, 3) && om.f.info.ncomp == im.f.info.ncomp;
        };
        // the first image with a hash value is the candidate owner of everybody with that value; the byte compares (10.8 KB each) run on
        // the workers; an image whose compare fails (two table sets under one 64-bit hash value: never seen) becomes an owner itself
        std::unordered_map<uint64_t, int> first;
        first.reserve((size_t)n);
        for (int i = 0; i < n; i++) {
            Img &im = S.imgs[(size_t)i];
            if (im.err != JDA_SUCCESS) continue;
            auto it = first.find(im.tab_hash);
            if (it == first.end()) first.emplace(im.tab_hash, i); else im.tab_owner = it->second;
        }
        p->workers->run(n, [&](int i) {
            Img &im = S.imgs[(size_t)i];
            if (im.err == JDA_SUCCESS && im.tab_owner != i && !same_tables(S.imgs[(size_t)im.tab_owner], im)) im.tab_owner = -1 - im.tab_owner;
        });
        std::vector<int> odd;                                  // (owners made by a failed compare, in order)
        for (int i = 0; i < n; i++) {
            Img &im = S.imgs[(size_t)i];
            if (im.err != JDA_SUCCESS) continue;
            if (im.tab_owner < 0) {                             // a collision: among the images that collided before, or an owner of its own
                im.tab_owner = i;
                for (int o : odd) if (S.imgs[(size_t)o].tab_hash == im.tab_hash && same_tables(S.imgs[(size_t)o], im)) { im.tab_owner = o; break; }
                if (im.tab_owner == i) odd.push_back(i);
            }
            if (im.tab_owner != i) { im.ctl_tables = S.imgs[(size_t)im.tab_owner].ctl_tables; continue; }
            const size_t to = n_tab * tab_stride;              // (owners in increasing order: a copy only ever moves towards the front)
            if (to != im.ctl_tables) memmove(S.pin + to, S.pin + im.ctl_tables, JDA_TABLE_BYTES);
            im.ctl_tables = to; n_tab++;
        }
    }
    ctl = n_tab * tab_stride;
    const size_t off_fparams = a16(ctl); ctl = off_fparams + a16((size_t)n * sizeof(jda_filter_params));
    const size_t off_sparams = ctl; ctl += a16((size_t)n * sizeof(jda_segscan_params));
    S.off_descs = ctl; ctl += a16((size_t)n * sizeof(jda_dev_desc));
    const size_t off_tparams = ctl; ctl += a16((size_t)n * sizeof(jda_strips_params));      // the tile lists are written on the device (jda_fill_strips)

    g_submit_clock.lap(1);
    // ---- launch plan and arena layout
    std::vector<jda_dev_desc> descs((size_t)n);
    uint32_t list_tiles[JDA_N_LISTS], list_ord[JDA_N_LISTS];
    int list_tab[JDA_N_LISTS];                                // whose tables the list's last image has
    memset(list_tiles, 0, sizeof(list_tiles)); memset(list_ord, 0, sizeof(list_ord));
    for (int m = 0; m < JDA_N_LISTS; m++) list_tab[m] = -1;
    size_t arena = 0;
    auto take = [&](size_t bytes) { const size_t o = arena; arena += a256(bytes); return o; };
    // regions: [control blob][raw][dc][work][ ZERO: scan | index | zero ][stats]; laid out by region so that one memset and
    // one read-back cover all images
    std::vector<size_t> raw_sz((size_t)n, 0);
    int n_dev = 0;
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        im.device = false;
        if (im.err != JDA_SUCCESS) continue;
        const jda_image_info &I = im.f.info;
        jda_dev_desc &D = descs[(size_t)i];
        int bpp = 0;
        im.n_blocks = (uint32_t)((size_t)I.mcus_x * I.mcus_y * I.blocks_per_mcu);
        im.fast = im.f.fast_provable != 0;
        const int rc = jda_fill_launch_desc(D, I, im.f.dc_id, im.f.ac_id, im.f.q_id, im.fast ? 1 : 0, im.f.general_p1, (uint32_t)(I.mcus_x * I.mcus_y), 0,
                                            outputs[i], S.pts[(size_t)i], S.opts[(size_t)i], &bpp);
        if (rc != JDA_SUCCESS) { im.err = rc; continue; }
        if (!im.f.device_ok) continue;                       // valid, but the serial host pre-scan has to make its index (jda_pipeline_wait)
        // the block records of the device pre-scan are rec_cap slots per 256-byte segment: 6 x the scan with the usual tables, up to
        // 16 x with a DHT whose shortest codes are one or two bits.  Past 8 x (+ 16 MB of grace) the image takes the serial path: a
        // batch of such files must not ask for an arena several times what the same files needed before the records existed
        if ((uint64_t)(im.f.raw_len / JDA_SEG_BYTES + 1u) * im.f.rec_cap * 4u > 8ull * im.f.raw_len + (16ull << 20)) continue;
        im.device = true; n_dev++;
        const int variant = jda_plain_variant(D);
        // (window size: the filtered length is not known yet; the unfiltered one is at most a few percent larger)
        D.scan_len = im.f.raw_len;
        const int big = D.scale_shift == 3 ? 0 : jda_big_window(D, variant);
        im.list = (uint32_t)jda_list_index(D, variant, big, 0, true);
        im.n_tiles = JDA_LIST_IS_THUMB_FLAT((int)im.list) ? 1u : count_tiles(D.mcus_x, D.mcus_y, D.mode, big);      // (a whole gray image at 1/8: one record)
        if (JDA_LIST_IS_THUMB_FLAT((int)im.list)) S.flat_max_items = std::max(S.flat_max_items, jda_flat_items(D));
        im.strip_off = list_tiles[im.list]; list_tiles[im.list] += im.n_tiles;
        // a tile record's `ord` counts the TABLE SETS of its list, not its images: the decode kernel's workgroups restage the tables in LDS
        // (three barriers) when it moves on -- images that share their tables (above) pass from one to the next without a barrier
        if (list_tab[im.list] >= 0 && list_tab[im.list] != im.tab_owner) list_ord[im.list]++;
        list_tab[im.list] = im.tab_owner;
        im.ord = list_ord[im.list];
        S.st.source_pixels += (int64_t)I.width * I.height;
        S.st.compressed_bytes += lens[i];
    }
    S.ctl_bytes = a256(ctl);
    arena = S.ctl_bytes;
    // The unfiltered scans.  Input that is page-locked where it lies (JDA_SUBMIT_PINNED_INPUT) is read there by the copy engine: no
    // host core touches those bytes.  Files that lie next to one another in the caller's memory (a loader's arena, a ring of receive
    // buffers) travel as ONE copy command -- a command per file costs the submitting thread 25-30 us, 2 ms for a batch of 64 --: the
    // covering range goes to the arena as it is (headers and gaps of up to 64 KB included) and a file's scan keeps its place in it
    // (raw_skip: its alignment).  A command that would carry less than 128 KB is not worth its cost: those files, and all of them when
    // the input is pageable, go through the pipeline's page-locked mirror (the workers copy them, one command takes them all).
    static const uint32_t direct_min = []() { const char *e = JDA_LAB_ENV("JDA_PIPE_DIRECT_MIN"); return e ? (uint32_t)atoi(e) : (uint32_t)JDA_PIPE_DIRECT_MIN_BYTES; }();
    struct Run { const uint8_t *src; size_t bytes, off; int files; uintptr_t range; };
    std::vector<Run> runs;
    std::vector<size_t> delta((size_t)n, 0);
    std::vector<int> run_of((size_t)n, -1);
    for (int i = 0; i < n; i++) { S.imgs[(size_t)i].direct = false; S.imgs[(size_t)i].raw_skip = 0; }
    if (flags & JDA_SUBMIT_PINNED_INPUT) {
        // a file is read where it lies when its entropy-coded bytes lie inside ONE page-locked range the library knows (jda_host_alloc /
        // jda_host_register); the others take the mirror.  A command copies [start rounded down to 16 bytes -- still the file's own
        // header --, the last file's last byte): nothing in front of, behind or between page-locked objects is read
        std::vector<int> dix;
        std::vector<uintptr_t> range_of((size_t)n, 0);
        for (int i = 0; i < n; i++) {
            Img &im = S.imgs[(size_t)i];
            uintptr_t rb = 0; size_t rl = 0;
            if (im.device && im.f.raw_off >= 16u && jda_host_range_of(jpegs[i] + im.f.raw_off - 15u, (size_t)im.f.raw_len + 15u, &rb, &rl)) { range_of[(size_t)i] = rb; dix.push_back(i); }
        }
        std::sort(dix.begin(), dix.end(), [&](int a, int b) { return jpegs[a] + S.imgs[(size_t)a].f.raw_off < jpegs[b] + S.imgs[(size_t)b].f.raw_off; });
        const uint8_t *last_end = NULL;
        for (int i : dix) {
            Img &im = S.imgs[(size_t)i];
            const uint8_t *b = jpegs[i] + im.f.raw_off, *e = b + im.f.raw_len;
            if (runs.empty() || runs.back().range != range_of[(size_t)i] || b < last_end || (size_t)(b - last_end) > ((size_t)64 << 10) || (size_t)(e - runs.back().src) > ((size_t)1 << 30)) {
                Run r;
                r.src = (const uint8_t *)((uintptr_t)b & ~(uintptr_t)15); r.bytes = 0; r.off = 0; r.files = 0; r.range = range_of[(size_t)i];
                runs.push_back(r);
            }
            Run &r = runs.back();
            r.bytes = (size_t)(e - r.src); r.files++;
            delta[(size_t)i] = (size_t)(b - r.src); run_of[(size_t)i] = (int)runs.size() - 1;
            last_end = e;
        }
        for (int i : dix) S.imgs[(size_t)i].direct = runs[(size_t)run_of[(size_t)i]].bytes >= direct_min;
    }
    for (int i = 0; i < n; i++) { Img &im = S.imgs[(size_t)i]; if (im.device && !im.direct) im.off_raw = take(a16(im.f.raw_len) + 16); }
    const size_t raw_end = arena;                             // [0, raw_end) = control blob + the mirrored scans: one H2D copy from the page-locked mirror
    for (Run &r : runs) if (r.bytes >= direct_min) r.off = take(a16(r.bytes) + 16);
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        if (!im.direct) continue;
        im.off_raw = runs[(size_t)run_of[(size_t)i]].off + (delta[(size_t)i] & ~(size_t)15);
        im.raw_skip = (uint32_t)(delta[(size_t)i] & 15u);
    }
    for (int m = 0; m < JDA_N_LISTS; m++) { S.list_n[m] = list_tiles[m]; S.list_off[m] = list_tiles[m] ? take((size_t)list_tiles[m] * sizeof(jda_strip)) : 0; }
    for (int i = 0; i < n; i++) { Img &im = S.imgs[(size_t)i]; if (im.device) im.off_dc = take((size_t)im.n_blocks * 2); }
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        if (!im.device) continue;
        // seg_sum | seg_start | two work lists | where the restart intervals start (+ the sentinel)
        im.n_segs_ub = im.f.raw_len / JDA_SEG_BYTES + 1u;
        im.off_rpos = a16((size_t)im.n_segs_ub * 4 * JDA_SEG_SUM_WORDS) + a16((size_t)im.n_segs_ub * 20) + a16((size_t)im.n_segs_ub * 8);
        im.off_fwork = im.off_rpos + (im.f.n_intervals ? a16(((size_t)im.f.n_intervals + 1) * 4) : 0);          // .. | the filter's chunk functions
        im.off_wt = im.off_fwork + a16(JDA_FILTER_WORK_BYTES(im.f.raw_len + 16u));                                 // .. | the walk's tables
        im.work_bytes = im.off_wt + JDA_WT_BYTES;
        // .. | RECORD mode: the segments' block records, the truncation candidates
        im.record = true;                                    // (device_ok streams have record slots: front_common)
        im.off_recs = im.off_cands = 0; im.cand_cap = 0;
        if (im.record) {
            // (images of one size would put their record regions a constant stride apart: a skew per image keeps the walkers of a
            // batch of like images -- all at the same place of their scans at the same time -- off each other's memory channels)
            static const size_t skew = []() { const char *e = JDA_LAB_ENV("JDA_PIPE_REC_SKEW"); return e ? (size_t)atoi(e) : (size_t)0; }();
            im.off_recs = a256(im.work_bytes) + a256(skew * (size_t)(i % 16));
            im.off_cands = im.off_recs + a16((size_t)im.n_segs_ub * im.f.rec_cap * 4);
            im.cand_cap = std::max<uint32_t>(1024u, im.n_segs_ub * 16u);      // (a high-quality photograph: five candidates per segment, most of them of walks that were redone)
            im.work_bytes = im.off_cands + (size_t)im.cand_cap * 16;
        }
        im.off_work = take(im.work_bytes);
    }
    const size_t zero_begin = arena;
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        if (!im.device) continue;
        im.scan_bytes = a16(std::max((size_t)im.f.raw_len + JDA_SCAN_PAD, (size_t)im.n_segs_ub * JDA_SEG_BYTES + 16));
        im.off_scan = take(im.scan_bytes);
    }
    for (int i = 0; i < n; i++) { Img &im = S.imgs[(size_t)i]; if (im.device) im.off_index = take(((size_t)im.n_blocks + 1) * 4); }
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        if (!im.device) continue;
        im.zero_bytes = a16(((size_t)im.n_segs_ub + 1) * 4) + (im.record && im.f.n_intervals ? a16((size_t)im.f.n_intervals * 8) : 0);      // entry states | RECORD mode: who ended which restart interval
        im.off_zero = take(im.zero_bytes);
    }
    S.off_stats_dev = arena;
    for (int i = 0; i < n; i++) { Img &im = S.imgs[(size_t)i]; im.off_stats = arena; arena += JDA_PIPE_STATS_BYTES; }
    S.stats_bytes = arena - S.off_stats_dev;
    arena = a256(arena);
    const size_t zero_end = arena;
    arena += 8192;                                            // slack: readers run a few hundred bytes past a (corrupt) scan

    g_submit_clock.lap(2);
    S.pin_stats = a256(raw_end);                              // the read-back of the result words sits behind the mirror
    const size_t pin_stats = S.pin_stats;
    if (S.pin_cap < pin_stats + S.stats_bytes + 256) {        // (strips and scans did not fit: grow, keeping the tables)
        uint8_t *np = NULL;
        const size_t want = a256((pin_stats + S.stats_bytes) * 5 / 4 + ((size_t)1 << 20));
        if (hipHostMalloc((void **)&np, want, hipHostMallocDefault) != hipSuccess) return JDA_ERROR_MEMORY;
        memcpy(np, S.pin, off_fparams);
        (void)hipHostFree(S.pin);
        S.pin = np; S.pin_cap = want;
    }
    if (S.dev_cap < arena) {
        if (S.dev) (void)hipFree(S.dev);
        S.dev = NULL; S.dev_cap = 0;
        const size_t want = a256(arena + arena / 4);
        hipError_t e = hipMalloc((void **)&S.dev, want);
        if (e != hipSuccess) {                               // give back what the slots that are not in flight hold, and what is not strictly needed: once
            (void)hipGetLastError();
            for (int k = 0; k < p->depth; k++) {
                jda_pipeline::Slot &o = p->slots[k];
                if (&o != &S && !o.in_flight && o.dev) { (void)hipFree(o.dev); o.dev = NULL; o.dev_cap = 0; }
            }
            e = hipMalloc((void **)&S.dev, a256(arena));
            if (e != hipSuccess) { S.dev = NULL; return jda_set_err(ctx, e, "hipMalloc(pipeline arena)"), JDA_ERROR_MEMORY; }
            S.dev_cap = a256(arena);
        } else
        S.dev_cap = want;
    }

    g_submit_clock.lap(3);
    // ---- control blob, part 2: parameters, descriptors, strips (in parallel: the strip lists are the bulk)
    jda_filter_params *fp = (jda_filter_params *)(S.pin + off_fparams);
    jda_segscan_params *sp = (jda_segscan_params *)(S.pin + off_sparams);
    jda_dev_desc *dd = (jda_dev_desc *)(S.pin + S.off_descs);
    std::vector<int> dev_ix;
    for (int i = 0; i < n; i++) if (S.imgs[(size_t)i].device) dev_ix.push_back(i);
    jda_strips_params *tp = (jda_strips_params *)(S.pin + off_tparams);
    uint32_t max_segs = 0, max_raw = 0, max_tiles = 0;
    std::vector<int> wt_first((size_t)n, -1);
    for (size_t k = 0; k < dev_ix.size(); k++) {
        const int i = dev_ix[k];
        Img &im = S.imgs[(size_t)i];
        const jda_image_info &I = im.f.info;
        uint8_t *B = S.dev;
        uint32_t *fres = (uint32_t *)(B + im.off_stats);
        uint32_t *pstats = (uint32_t *)(B + im.off_stats + 16);
        jda_filter_params &F = fp[k];
        F.raw = B + im.off_raw; F.out = B + im.off_scan; F.result = fres; F.raw_skip = im.direct ? im.raw_skip : 0u; F.pad_ = 0;
        F.raw_len = im.f.raw_len + F.raw_skip;
        const uint32_t n_int = im.f.n_intervals;                       // 0: no restart intervals
        F.restart_pos = (uint32_t *)(B + im.off_work + im.off_rpos);
        F.restart_cap = n_int ? n_int + 1u : 0u;
        F.work = (uint32_t *)(B + im.off_work + im.off_fwork);
        max_raw = std::max(max_raw, F.raw_len);
        jda_dev_desc &D = descs[(size_t)i];
        D.tables = B + im.ctl_tables;
        D.blk_index = (const uint32_t *)(B + im.off_index);
        D.blk_dc = (const int16_t *)(B + im.off_dc);
        D.scan = B + im.off_scan;
        D.scan_len = (uint32_t)(im.scan_bytes - JDA_SCAN_PAD);          // upper bound of the filtered length; the bytes behind it are zero
        dd[i] = D;
        jda_segscan_params P;
        memset(&P, 0, sizeof(P));
        P.scan = B + im.off_scan; P.tables = B + im.ctl_tables;
        P.entry_cur = (uint32_t *)(B + im.off_zero); P.entry_nxt = P.entry_cur;
        P.worklist = (uint32_t *)(B + im.off_work + a16((size_t)im.n_segs_ub * 4 * JDA_SEG_SUM_WORDS) + a16((size_t)im.n_segs_ub * 20)); P.worklist_cap = im.n_segs_ub;
        P.seg_sum = (uint32_t *)(B + im.off_work); P.seg_start = (const uint32_t *)(B + im.off_work + a16((size_t)im.n_segs_ub * 4 * JDA_SEG_SUM_WORDS));
        if (im.record) {
            P.records = (uint32_t *)(B + im.off_work + im.off_recs); P.rec_cap = im.f.rec_cap;
            P.cands = (uint32_t *)(B + im.off_work + im.off_cands); P.cand_cap = im.cand_cap;
            if (im.f.n_intervals) P.rst_events = (uint32_t *)(B + im.off_zero + a16(((size_t)im.n_segs_ub + 1) * 4));
        }
        P.blk_index = (uint32_t *)(B + im.off_index); P.blk_dc = (int16_t *)(B + im.off_dc);
        P.stats = pstats;
        int &wo = wt_first[(size_t)im.tab_owner];              // the first image ON THE DEVICE with these tables makes the walk's tables for all of them
        if (wo < 0) wo = i;
        P.walk_tables = B + S.imgs[(size_t)wo].off_work + S.imgs[(size_t)wo].off_wt;
        P.walk_tables_shared = wo != i ? 1u : 0u;
        P.scan_len = im.f.raw_len; P.n_segs = im.n_segs_ub; P.n_blocks_total = im.n_blocks;
        P.nblocks = (uint8_t)I.blocks_per_mcu; P.nluma = (uint8_t)(I.blocks_per_mcu - (I.ncomp == 3 ? 2 : 0));
        for (int c = 0; c < 3; c++) { P.dc_id[c] = im.f.dc_id[c]; P.ac_id[c] = im.f.ac_id[c]; }
        P.filter_result = fres;
        if (n_int) {                                                    // the walk ends intervals where the filter found the markers
            P.restart_pos = F.restart_pos; P.n_intervals = n_int;
            P.interval_blocks = (uint32_t)I.restart_interval * (uint32_t)I.blocks_per_mcu;
            P.round_last = ((uint32_t)(I.mcus_x * I.mcus_y) % (uint32_t)I.restart_interval) == 0 ? 1u : 0u;
        }
        sp[k] = P;
        max_segs = std::max(max_segs, im.n_segs_ub);
        jda_strips_params &TP = tp[k];
        TP.dst = (jda_strip *)(B + S.list_off[im.list]) + im.strip_off; TP.n_padded = im.n_tiles; TP.image = (uint32_t)i; TP.ord = im.ord;
        TP.mcus_x = D.mcus_x; TP.mcus_y = D.mcus_y; TP.per = jda_mcus_per_tile(D.mode);
        if (JDA_LIST_IS_THUMB_FLAT((int)im.list)) { TP.mcus_x = 1; TP.mcus_y = 1; TP.per = 1; }      // (jda_fill_strips writes the one record: the image, first = 1)
        max_tiles = std::max(max_tiles, im.n_tiles);
    }
    g_submit_clock.lap(4);
    p->workers->run((int)dev_ix.size(), [&](int k) {
        const int i = dev_ix[(size_t)k];
        const Img &im = S.imgs[(size_t)i];
        // the file's entropy-coded bytes into the page-locked mirror (the workers' memcpy is the only time the host touches them):
        // the whole batch then travels as ONE asynchronous copy instead of a blocking pageable copy per file
        if (!im.direct) copy_to_mirror(S.pin + im.off_raw, jpegs[i] + im.f.raw_off, im.f.raw_len);
    });

    g_submit_clock.lap(5);
    // ---- enqueue: upload stream
    const int up_ix = t % (p->n_upx + 1);
    hipStream_t s_up = up_ix ? p->s_upx[up_ix - 1] : p->s_up;           // batches take the pre-scan streams in turn
    hipError_t e = hipSuccess;
    uint8_t *B = S.dev;
    if (n_dev) {
        e = hipMemsetAsync(B + zero_begin, 0, zero_end - zero_begin, p->s_copy);
        if (e == hipSuccess) e = hipMemcpyAsync(B, S.pin, raw_end, hipMemcpyHostToDevice, p->s_copy);      // control blob + every mirrored scan
        S.st.h2d_bytes += (int64_t)raw_end;
        for (size_t r = 0; r < runs.size() && e == hipSuccess; r++) {                                      // .. and the others from where they lie
            if (runs[r].bytes < direct_min) continue;
            e = hipMemcpyAsync(B + runs[r].off, runs[r].src, runs[r].bytes, hipMemcpyHostToDevice, p->s_copy);
            S.st.h2d_bytes += (int64_t)runs[r].bytes;
        }
        // The copy stream carries the memset and the copy ONLY: with the filter behind the copy on the same stream, the next batch's
        // 110 MB (2 ms at 55 GB/s) could not start before this batch's filter had run -- and the filter's workgroups wait for the decode
        // kernel of the batch in front to give the CUs' LDS back: the copy stream was the pipeline's period (3.35 ms per batch of 64 x
        // 4096x4096, profiles/r03_pipeline_timeline_filter_on_copy_stream.txt).  The filter opens the batch's pre-scan stream instead.
        static const bool filter_on_copy = []() { const char *v = JDA_LAB_ENV("JDA_PIPE_FILTER_STREAM"); return v && v[0] == 'c'; }();      // (measuring)
        hipStream_t s_f = filter_on_copy ? p->s_copy : s_up;
        if (!filter_on_copy) {
            if (e == hipSuccess) e = hipEventRecord(S.ev_copy, p->s_copy);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_up, S.ev_copy, 0);
        }
        if (e == hipSuccess) e = jda_launch_fill_strips((const jda_strips_params *)(B + off_tparams), (uint32_t)dev_ix.size(), max_tiles, s_f);
        if (e == hipSuccess) e = jda_launch_walk_tables((const jda_segscan_params *)(B + off_sparams), (uint32_t)dev_ix.size(), s_f);
        if (e == hipSuccess) e = jda_launch_filter((const jda_filter_params *)(B + off_fparams), (uint32_t)dev_ix.size(), max_raw, s_f);
        if (filter_on_copy) {
            if (e == hipSuccess) e = hipEventRecord(S.ev_copy, p->s_copy);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_up, S.ev_copy, 0);
        }
        if (e == hipSuccess) {
            const jda_segscan_params *dp = (const jda_segscan_params *)(B + off_sparams);
            const uint32_t ns = (uint32_t)dev_ix.size();
            // round 0, the counting round (RECORD mode: + a record per block), the work-list rounds, sums, then WRITE (restart
            // streams) / finalize + candidates (jda_kernels.hip)
            e = jda_launch_prescan_passes(dp, ns, max_segs, JDA_PIPE_SPEC_ROUNDS, JDA_PIPE_MAX_ROUNDS, 1, s_up);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(S.pin + S.pin_stats, B + S.off_stats_dev, S.stats_bytes, hipMemcpyDeviceToHost, s_up);
    }
    if (e == hipSuccess) e = hipEventRecord(S.ev_up, s_up);
    // ---- enqueue: decode stream
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, S.ev_up, 0);
    for (int m = 0; m < JDA_N_LISTS && e == hipSuccess && n_dev; m++) {
        if (!S.list_n[m]) continue;
        e = jda_launch_decode(JDA_LIST_MODE(m), JDA_LIST_FAST(m), JDA_LIST_VARIANT(m), JDA_LIST_BIG(m), JDA_LIST_CONT(m), (const jda_dev_desc *)(B + S.off_descs), (const jda_strip *)(B + S.list_off[m]), S.list_n[m],
                              JDA_LIST_IS_THUMB_FLAT(m) ? S.flat_max_items : 0u, ctx->stream);
        S.st.launches++;
    }
    if (e == hipSuccess) e = hipEventRecord(S.ev_dec, ctx->stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(p->s_copy); (void)hipStreamSynchronize(s_up); (void)hipStreamSynchronize(ctx->stream); return jda_set_err(ctx, e, "jda_pipeline_submit"); }
    g_submit_clock.lap(6);
    S.ticket = t; S.in_flight = true;
    p->next_ticket++;
    *ticket = t;
    return JDA_SUCCESS;
}

// the serial path for one image: host pre-scan, upload, decode into the caller's surface; returns the image's status
// (On a context -- a stream -- of the pipeline's own: the decodes of the batches submitted after this one are already queued on the
// caller's stream, and the redo's synchronisations would wait for all of them.  It is ordered behind this batch's optimistic decode
// of the same surface by jda_pipeline_wait's wait for the batch's event.)
static int slow_path(jda_pipeline *p, const uint8_t *jpeg, int32_t len, const jda_output &O, int32_t pt, int32_t opt)
{
    if (!p->slow_ctx) { int32_t e = JDA_SUCCESS; p->slow_ctx = jda_create(p->ctx->device, &e); }
    jda_ctx *ctx = p->slow_ctx ? p->slow_ctx : p->ctx;
    int32_t err = JDA_SUCCESS;
    jda_image *img = jda_prepare_ex(jpeg, len, 0, &err);
    if (!img) return err;
    jda_dev_image *d = jda_upload(ctx, img, &err);
    uint32_t nok = 0;
    (void)jda_image_block_index(img, &nok);
    const jda_image_info I = *jda_image_get_info(img);
    jda_image_free(img);
    if (!d) return err;
    int rc = JDA_SUCCESS;
    jda_batch *b = jda_batch_create(ctx, 1, &d, &O, &pt, &opt, &err);
    if (!b) rc = err;
    else {
        const bool complete = nok == (uint32_t)(I.mcus_x * I.mcus_y);
        if (!complete) {                                      // the reference leaves the MCUs behind the bad one undrawn: zeros here
            int bpp, ow, oh, cw, ch;
            if (jda_output_geometry(&I, pt, opt, &bpp, &ow, &oh, &cw, &ch) == JDA_SUCCESS) {
                const size_t wbytes = (size_t)std::min(O.width_px, cw) * bpp;
                (void)hipMemset2DAsync(O.pixels, (size_t)O.pitch_bytes, 0, wbytes, (size_t)std::min(O.rows, ch), ctx->stream);
            }
        }
        rc = jda_batch_decode(ctx, b);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == JDA_SUCCESS) rc = JDA_ERROR_HIP;
        jda_batch_destroy(ctx, b);
        if (rc == JDA_SUCCESS && !complete) rc = JDA_DECODE_ERROR;      // jpeg.inl:5354-5356
    }
    jda_dev_image_free(ctx, d);
    return rc;
}

int jda_pipeline_wait(jda_pipeline *p, int32_t ticket, int32_t *status)
{
    if (!p) return JDA_ERROR_NO_DEVICE;
    if (ticket < 0 || ticket >= p->next_ticket) return JDA_INVALID_PARAMETER;
    jda_pipeline::Slot &S = p->slots[ticket % p->depth];
    if (!S.in_flight || S.ticket != ticket) return JDA_INVALID_PARAMETER;
    jda_ctx *ctx = p->ctx;
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipEventSynchronize(S.ev_dec);
    S.in_flight = false;
    if (e != hipSuccess) return jda_set_err(ctx, e, "jda_pipeline_wait");
    const int n = (int)S.imgs.size();
    int first_err = JDA_SUCCESS;
    for (int i = 0; i < n; i++) {
        Img &im = S.imgs[(size_t)i];
        int st = im.err;
        if (st == JDA_SUCCESS) {
            bool redo = !im.device;
            if (im.device) {
                const uint32_t *rb = (const uint32_t *)(S.pin + S.pin_stats + (im.off_stats - S.off_stats_dev));
                const uint32_t *ps = rb + 4;
                uint32_t max_ac = 0, max_dc = 0;
                // the last round left nothing to walk; enough blocks; no bad code; the closing entry written once; and, with restart
                // intervals, as many markers as the MCU count asks for, each where the count puts it
                bool ok = ps[7] == 1 && ps[6] == 1 && ps[0] == 0 && ps[1] == 1;
                if (im.f.n_intervals && (rb[1] + 1u != im.f.n_intervals || ps[5] != 0)) ok = false;
                max_ac = ps[2]; max_dc = ps[3];
                S.st.spec_rounds_max = std::max<int32_t>(S.st.spec_rounds_max, [&]() { int r = 2; while (r <= JDA_PIPE_MAX_ROUNDS + 1 && ps[8 + r]) r++; return r; }());   // rounds that had something to walk
                {
                    static const bool trace_lists = JDA_LAB_ENV("JDA_PIPE_TRACE_LISTS") != NULL;      // (what the rounds behind round 1 had to walk)
                    if (trace_lists && i == 0) fprintf(stderr, "jda_pipeline: ticket %d image 0: %u segments, work lists of rounds 2.. : %u %u %u %u %u\n", ticket, rb[0] / JDA_SEG_BYTES + 1u, ps[10], ps[11], ps[12], ps[13], ps[14]);
                }
                if (ok && im.fast && !jda_front_fast_mul(S.pin + im.ctl_tables, &im.f, max_ac, (int32_t)max_dc)) ok = false;   // a magnitude no legal stream has
                redo = !ok;
                if (ok) S.st.device_images++;
                else {
                    static const bool trace = JDA_LAB_ENV("JDA_PIPE_TRACE") != NULL;
                    if (trace) fprintf(stderr, "jda_pipeline: image %d (%dx%d, %u restart intervals) of ticket %d goes to the host path: filter %u bytes / %u markers, result words %u %u %u %u %u, [5] %u [6] %u, settled %u, fast %d\n",
                                       i, im.f.info.width, im.f.info.height, im.f.n_intervals, ticket, rb[0], rb[1], ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6],
                                       ps[7], (int)im.fast);
                }
            }
            if (redo) { st = slow_path(p, S.jpegs[(size_t)i], S.lens[(size_t)i], S.outs[(size_t)i], S.pts[(size_t)i], S.opts[(size_t)i]); S.st.host_path_images++; }
        }
        if (status) status[i] = st;
        if (st != JDA_SUCCESS && first_err == JDA_SUCCESS) first_err = st;
        if (st != JDA_SUCCESS) S.st.failed_images++;
    }
    p->total.images += S.st.images; p->total.device_images += S.st.device_images; p->total.host_path_images += S.st.host_path_images;
    p->total.failed_images += S.st.failed_images; p->total.source_pixels += S.st.source_pixels; p->total.compressed_bytes += S.st.compressed_bytes;
    p->total.h2d_bytes += S.st.h2d_bytes; p->total.launches += S.st.launches;
    p->total.spec_rounds_max = std::max(p->total.spec_rounds_max, S.st.spec_rounds_max);
    (void)first_err;
    return JDA_SUCCESS;
}

// After jda_pipeline_wait(ticket) and before the slot is used again: the per-block index and DC predictors the device made for
// image i of that batch (tests: they must equal the serial pre-scan's).  JDA_INVALID_PARAMETER if the image did not take the
// device path.
int jda_pipeline_read_index(jda_pipeline *p, int32_t ticket, int32_t i, uint32_t *index, int16_t *dc, uint32_t *filtered_len)
{
    if (!p) return JDA_ERROR_NO_DEVICE;
    if (ticket < 0 || ticket >= p->next_ticket) return JDA_INVALID_PARAMETER;
    jda_pipeline::Slot &S = p->slots[ticket % p->depth];
    if (S.in_flight || S.ticket != ticket || i < 0 || i >= (int)S.imgs.size() || !S.imgs[(size_t)i].device) return JDA_INVALID_PARAMETER;
    if (!S.dev) return JDA_INVALID_PARAMETER;               // (the slot's arena was given back when another slot ran out of memory)
    const Img &im = S.imgs[(size_t)i];
    (void)hipSetDevice(p->ctx->device);
    hipError_t e = hipSuccess;
    if (index) e = hipMemcpy(index, S.dev + im.off_index, ((size_t)im.n_blocks + 1) * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && dc) e = hipMemcpy(dc, S.dev + im.off_dc, (size_t)im.n_blocks * 2, hipMemcpyDeviceToHost);
    if (filtered_len) *filtered_len = ((const uint32_t *)(S.pin + S.pin_stats + (im.off_stats - S.off_stats_dev)))[0];
    return e == hipSuccess ? JDA_SUCCESS : jda_set_err(p->ctx, e, "jda_pipeline_read_index");
}

int jda_pipeline_get_stats(const jda_pipeline *p, jda_pipeline_stats *out)
{
    if (!p || !out) return JDA_INVALID_PARAMETER;
    *out = p->total;
    return JDA_SUCCESS;
}

} // extern "C"
