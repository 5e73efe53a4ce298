// jda_kernels.hip -- gfx950 (CDNA4) kernels of the decode path.  Written for wave64 only.
//
// jda_decode_tiles_persistent<MODE, FAST, VARIANT, CONT>:
// a workgroup is as many independent wavefronts as fit in a CU's LDS next to one copy of the image's
// tables (16 x 9.8 KB + 6.8 KB for 4:2:0 = one 1024-thread workgroup per CU); each wavefront decodes one
// tile = up to 64 consecutive 8x8 blocks of one MCU row (10 MCUs of 4:2:0).  After the tables are staged
// there is no workgroup barrier: phases are separated by wave-local fences, so wavefronts drift freely.
//   P1  lane = block: Huffman/RLE expand from the per-block index entry into the block's int16[64]
//       in LDS (coefficients never touch HBM); the wave then builds the IDCT work lists with a DPP
//       prefix sum and ballots (non-empty columns, row-variant classes);
//   P2  lane = (block, non-empty column): dequant + IDCT column stage, in place;
//   P3  lane = (block, row): IDCT row stage + range limit, grouped by the reference's row variant;
//   P4  lanes tile the tile's output rows: YCbCr -> RGB and 16-byte stores to consecutive addresses.
// While a wavefront decodes tile n it prefetches tile n+1's index entries and scan slice (HBM -> registers
// -> LDS) and tile n+2's record.
// jda_filter_*, jda_segscan_* : the marker filter and the per-block index on the device.
// No MFMA: the IDCT is shift/add integer work; the decode kernel is bound by VALU issue.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

// A kernel's dynamic-LDS limit is a property of the function ON A DEVICE: set once per device and kernel (a process may drive
// several GPUs -- JPEGDEC::setDevice, jda_node -- from several threads: the flags are atomic, the attribute call is idempotent).
static hipError_t jda_ensure_lds_limit(const void *fn, int bytes, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// Launches per kernel function of this library, counted where they are made (an atomic add per launch): jda_kernel_launch_counts
// reports which kernels of the code object a process has used -- tests/test_gpu_zz_kernel_coverage.py holds the suite to all of them.
struct jda_launch_slot { std::atomic<const void *> fn; std::atomic<unsigned long long> n; };
#define JDA_LAUNCH_SLOTS 128
static jda_launch_slot g_launch_slots[JDA_LAUNCH_SLOTS];
static void jda_count_launch(const void *fn)
{
    const uint32_t h = (uint32_t)((((uintptr_t)fn >> 3) * 0x9E3779B97F4A7C15ull) >> 57);
    for (uint32_t i = 0; i < JDA_LAUNCH_SLOTS; i++) {
        jda_launch_slot &S = g_launch_slots[(h + i) % JDA_LAUNCH_SLOTS];
        const void *cur = S.fn.load(std::memory_order_acquire);
        if (cur == nullptr && S.fn.compare_exchange_strong(cur, fn, std::memory_order_acq_rel)) cur = fn;
        if (cur == fn) { S.n.fetch_add(1, std::memory_order_relaxed); return; }
    }
}
#define JDA_LAUNCH(kernel, grid, block, lds, stream, ...) do { jda_count_launch((const void *)(kernel)); hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); } while (0)
// "<kernel symbol> <launches>\n" for every kernel launched so far; returns the bytes the whole report needs (cap too small: truncated)
extern "C" int jda_kernel_launch_counts(char *buf, int cap)
{
    int need = 0, written = 0;
    bool full = false;                                          // (whole lines only, none behind the first that did not fit)
    for (uint32_t i = 0; i < JDA_LAUNCH_SLOTS; i++) {
        const void *fn = g_launch_slots[i].fn.load(std::memory_order_acquire);
        if (!fn) continue;
        const char *name = hipKernelNameRefByPtr(fn, nullptr);
        char line[512];
        int n = snprintf(line, sizeof(line), "%s %llu\n", name ? name : "?", g_launch_slots[i].n.load(std::memory_order_relaxed));
        if (n >= (int)sizeof(line)) { n = (int)sizeof(line) - 1; line[n - 1] = '\n'; }
        if (buf && !full && written + n < cap) { memcpy(buf + written, line, (size_t)n); written += n; } else full = true;
        need += n;
    }
    if (buf && cap > 0) buf[written] = 0;                       // the string ends where its content does
    return need + 1;
}

__device__ unsigned long long *g_jda_trace = nullptr;
// optional per-wave start/finish stamps of the persistent kernel (profiling aid, tools/wg_balance.py): the constant
// 100 MHz clock (s_memrealtime) at entry and exit of every wave -> how evenly the static tile split loads the CUs
__device__ unsigned long long *g_jda_wgtrace = nullptr;
// inside P1 (lane 0 of wave 0 of every 16th workgroup; the last tile decoded wins).  Costs a global load per
// hook, so only in -DJDA_PROFILE_P1 builds.
#ifdef JDA_PROFILE_P1
#define JDA_P1_TRACE(slot) do { if (g_jda_trace && threadIdx.x == 0 && blockIdx.x % 16 == 0) g_jda_trace[(blockIdx.x / 16 * 4) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#endif
#include "jda_device_core.h"
#include "jda_plan.h"

// optional phase trace (profiling aid, off unless jda_internal_set_trace() was called): per traced
// workgroup and wave, the shader clock at each phase boundary
#define JDA_TRACE_STRIDE 16
#define JDA_TRACE(slot) do { if (trace && lane == 0 && wave < 4) trace[(blockIdx.x / JDA_TRACE_STRIDE * 4 + wave) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)

// the same inside the persistent kernel: the second tile a traced workgroup decodes (steady state)
#ifndef JDA_TRACE_ITER
#define JDA_TRACE_ITER 1        // which tile of a traced wavefront is stamped (1: its second; 60: steady state, the memory system saturated)
#endif
#ifdef JDA_PHASE_TRACE      // (make lib EXTRA=-DJDA_PHASE_TRACE; the stamps cost SGPRs, so they are not in the product build)
#define JDA_PTRACE(slot) do { if (trace && iter == JDA_TRACE_ITER && lane == 0 && wave < 4) trace[(blockIdx.x / JDA_TRACE_STRIDE * 4 + wave) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define JDA_PTRACE(slot) ((void)0)
#endif

// wave-local phase boundary: LDS operations of one wavefront complete in order, so ordering the
// compiler is all that is needed -- no s_barrier
#define JDA_WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// The image descriptor is wave-uniform.  Read through the global pointer the compiler has to assume
// that the kernel's own stores may change it: every field access became a vector load followed by
// s_waitcnt vmcnt(0) -- a full memory round trip that also drains the output stores in flight (one
// per item of the colour stage's loop).  Copy it once per image into a local whose fields are
// readfirstlane'd, i.e. live in SGPRs.
__device__ __forceinline__ uint32_t jda_uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <typename P> __device__ __forceinline__ P jda_uni_ptr(P p)
{
    const uint64_t v = (uint64_t)p;
    return (P)(((uint64_t)jda_uni32((uint32_t)(v >> 32)) << 32) | jda_uni32((uint32_t)v));
}
// VARIANT 1 / 2 / 3 (the plain cases: full size, 24-bit multiplies, output RGB8888 / RGB565 little endian / 8-bit gray -- of a
// colour file: luma only, its chroma blocks never decoded --; the host puts only such images in those launch lists): these
// fields are constants, and everything that tests them folds away (1837 -> 1775 VALU per tile for RGB8888)
template <int VARIANT = 0, int MODE = JDA_MODE_420>
__device__ __forceinline__ jda_dev_desc jda_desc_uniform(const jda_dev_desc *p)
{
    const jda_dev_desc JDA_GLOBAL *g = JDA_G(const jda_dev_desc, p);
    jda_dev_desc L;
    L.scan = jda_uni_ptr(g->scan); L.blk_index = jda_uni_ptr(g->blk_index); L.blk_dc = jda_uni_ptr(g->blk_dc);
    L.tables = jda_uni_ptr(g->tables); L.out = jda_uni_ptr(g->out);
    L.blk_cont_first = jda_uni_ptr(g->blk_cont_first); L.blk_cont = jda_uni_ptr(g->blk_cont);      // (dead in the kernels that do not decode in chunks)
    L.out_pitch = jda_uni32(g->out_pitch); L.out_w = jda_uni32(g->out_w); L.out_rows = jda_uni32(g->out_rows);
    L.mcus_x = jda_uni32(g->mcus_x); L.mcus_y = jda_uni32(g->mcus_y); L.n_mcus_ok = jda_uni32(g->n_mcus_ok);
    L.scan_len = jda_uni32(g->scan_len); L.strip_mcus = jda_uni32(g->strip_mcus);
#pragma unroll
    for (int i = 0; i < 4; i++) L.cfg[i] = jda_uni32(g->cfg[i]);      // mode .. pad_: sixteen byte fields in four SGPRs
    if (VARIANT >= 1) {
        L.scale_shift = 0; L.pad_[0] = 0; L.strip_mcus = 0;
        L.pixel_type = VARIANT == 1 ? JDA_RGB8888 : (VARIANT == 2 ? JDA_RGB565_LITTLE_ENDIAN : JDA_EIGHT_BIT_GRAYSCALE);
        L.gray_from_color = (VARIANT == 3 && MODE != JDA_MODE_GRAY) ? 1 : 0;
    }
    return L;
}

// The same through the CONSTANT address space: the descriptor array is written before the launch and never by a kernel, and the
// address is uniform, so these are scalar loads (s_load_dwordx8: no vector memory instruction, no v_readfirstlane, a scalar-cache
// hit for every wavefront but the first).  What the DC thumbnail kernel uses: its wavefronts live for three dependent round trips,
// and two of them are this.  (The persistent decode kernel re-reading its descriptor per phase instead of holding it in SGPRs
// was measured: 28 -> 13 SGPR spills, 1.5 % SLOWER -- the scalar loads' waits cost more than the v_readlanes they replace.)
typedef const jda_dev_desc __attribute__((address_space(4))) *jda_desc_cptr;
template <int VARIANT = 0, int MODE = JDA_MODE_420>
__device__ __forceinline__ jda_dev_desc jda_desc_const(jda_desc_cptr c)
{
    asm volatile("" : "+s"(c));
    jda_dev_desc L;
    L.scan = (const uint8_t *)c->scan; L.blk_index = (const uint32_t *)c->blk_index; L.blk_dc = (const int16_t *)c->blk_dc;
    L.tables = (const uint8_t *)c->tables; L.out = (uint8_t *)c->out;
    L.blk_cont_first = (const uint32_t *)c->blk_cont_first; L.blk_cont = (const uint32_t *)c->blk_cont;
    L.out_pitch = c->out_pitch; L.out_w = c->out_w; L.out_rows = c->out_rows;
    L.mcus_x = c->mcus_x; L.mcus_y = c->mcus_y; L.n_mcus_ok = c->n_mcus_ok;
    L.scan_len = c->scan_len; L.strip_mcus = c->strip_mcus;
#pragma unroll
    for (int i = 0; i < 4; i++) L.cfg[i] = c->cfg[i];
    if (VARIANT >= 1) {
        L.scale_shift = 0; L.pad_[0] = 0; L.strip_mcus = 0;
        L.pixel_type = VARIANT == 1 ? JDA_RGB8888 : (VARIANT == 2 ? JDA_RGB565_LITTLE_ENDIAN : JDA_EIGHT_BIT_GRAYSCALE);
        L.gray_from_color = (VARIANT == 3 && MODE != JDA_MODE_GRAY) ? 1 : 0;
    }
    return L;
}
__device__ __forceinline__ jda_desc_cptr jda_desc_at(const jda_dev_desc *descs, uint32_t image)
{
    return (jda_desc_cptr)(uintptr_t)(descs + image);
}

// a tile record (16 bytes, wave-uniform address) -> SGPRs
__device__ __forceinline__ jda_strip jda_unpack_record(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    jda_strip S;
    S.image = jda_uni32(w0);
    const uint32_t yx = jda_uni32(w1), cf = jda_uni32(w2);
    S.mcu_y = (uint16_t)(yx & 0xffffu); S.mcu_x0 = (uint16_t)(yx >> 16);
    S.count = (uint8_t)(cf & 0xffu); S.first = (uint8_t)((cf >> 8) & 0xffu); S.pad_ = 0;
    S.ord = jda_uni32(w3);
    return S;
}
__device__ __forceinline__ jda_strip jda_load_record(const jda_strip *tp)
{
    const uint32_t JDA_GLOBAL *w = JDA_G(const uint32_t, tp);
    return jda_unpack_record(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------
// Persistent variant (the default): one workgroup per CU (sized by LDS), each given a contiguous run of the
// batch's tiles.  The wavefronts of a workgroup DRAW their tiles from that run through a counter in LDS: the
// SIMD arbiter does not share issue slots fairly, so with a fixed tile list per wavefront some wavefronts were
// done after 65 % of the kernel's time and their SIMDs ran the rest under-occupied (tools/wg_balance.py).
// While a wavefront decodes tile j it fetches what tile j+1 needs -- per-lane index entries during the entropy
// phase, the scan slice during the column stage -- and draws tile j+2 and loads its record, so a tile never
// waits on the record -> index -> scan chain (three dependent HBM latencies).  The tables in LDS belong to one
// image; tiles are drawn in order, so every wavefront meets an image boundary of the run exactly once: all
// wait at a barrier, the wavefront that drew the new image's first tile restages the tables, second barrier.
// A wavefront that runs out of tiles still walks the remaining boundaries (the barrier counts every wave).
template <int MODE>
__device__ __forceinline__ uint32_t jda_draw_tile(uint32_t *ctr, uint32_t lane)
{
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(ctr, 1u);            // ds_add_rtn_u32
    return jda_uni32(v);
}

// per-lane index / DC entries of a tile + the entry just past it (for the window bounds)
// CONT: also the block's place in the continuation entries (cf0 .. cf1: jda_p1c_*)
template <int MODE, int CONT = 0>
__device__ __forceinline__ void jda_issue_index_loads(const jda_dev_desc &D, const jda_strip &S, uint32_t lane,
                                                      jda_p1_inputs &in, uint32_t &ix_end, uint32_t *cf0 = nullptr, uint32_t *cf1 = nullptr)
{
    typedef jda_mode_traits<MODE> T;
    uint32_t count = S.count;
    const uint32_t first_mcu = S.mcu_y * D.mcus_x + S.mcu_x0;
    if (first_mcu >= D.n_mcus_ok) count = 0;
    else if (first_mcu + count > D.n_mcus_ok) count = D.n_mcus_ok - first_mcu;
    const uint32_t first_block = first_mcu * T::NBLK, nb = count * T::NBLK;
    in.lb = lane; in.ix = 0; in.pred = 0; ix_end = 0;
    in.active = lane < nb;
    if (in.active) {
        in.ix = JDA_G(const uint32_t, D.blk_index)[first_block + lane];
        in.pred = JDA_G(const int16_t, D.blk_dc)[first_block + lane];
    }
    if (CONT) {
        *cf0 = *cf1 = 0;
        if (in.active) { *cf0 = JDA_G(const uint32_t, D.blk_cont_first)[first_block + lane]; *cf1 = JDA_G(const uint32_t, D.blk_cont_first)[first_block + lane + 1u]; }
    }
    if (nb) ix_end = JDA_G(const uint32_t, D.blk_index)[first_block + nb];     // uniform address
}

// Bring the staged tables up to image `target` of the run (see above).  P / DP: the tile this wavefront is about
// to decode and its image's descriptor (have == false: the wavefront has no tile left).  The wavefront that drew the new
// image's first tile publishes where its tables are; then the WHOLE workgroup stages them (one wavefront doing it alone kept
// the other fifteen at the barrier for eleven rounds of loads and entry conversions: 2-4 % of a batch of small images).
#define JDA_ADVANCE_TABLES(target, have, P, DP)                                                         \
    while (staged < (target)) {                                                                           \
        __syncthreads();                              /* nobody reads the old tables any more */          \
        if ((have) && (P).first && (P).ord == staged + 1u && lane == 0) *tab_src = (unsigned long long)(DP).tables;  \
        __syncthreads();                                                                                  \
        jda_p0_tables_from((const uint8_t *)*(volatile unsigned long long *)tab_src, threadIdx.x, 64u * n_waves, tab, L::LONG_LDS != 0); \
        __syncthreads();                                                                                  \
        staged++;                                                                                         \
    }

#ifndef JDA_PRIO_P1
#define JDA_PRIO_P1 3
#define JDA_PRIO_IDCT 1
#define JDA_PRIO_P4 3
#endif
#ifndef JDA_GRID_MULT_DEFAULT
#define JDA_GRID_MULT_DEFAULT 1
#endif
#ifndef JDA_EXP_SKIP
#define JDA_EXP_SKIP 0       // profiling builds (tools/phase_count_libs.sh): 1 no P4, 2 no P3, 4 no P2, 8 no lists, 16 no P1
#endif
// P1 of a tile in chunks (CONT kernels): pass A, the tile's continuation entries in passes of 64, the finish (jda_p1c_*)
#define JDA_P1C_PRELOADED 2          // passes whose entries are asked for before pass A runs
template <int MODE>
__device__ __forceinline__ uint32_t jda_p1_chunked(const jda_dev_desc &D, const jda_tile_ctx &C, const jda_p1_inputs &in, uint32_t cf0, uint32_t cf1, const jda_lane_pre &LP,
                                                   const uint8_t *tab, uint8_t *wl, const uint8_t *win, uint32_t win_cap, uint32_t lane)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t win_len = C.win_len < win_cap ? C.win_len : win_cap;
    const bool chunked = C.win_need <= win_len && !(D.pad_[0] & JDA_DESC_GENERAL_P1);       // uniform
    // the tile's entries are contiguous: from the first block's first to the last block's last
    const uint32_t nb = C.count * (uint32_t)T::NBLK;
    const uint32_t c0 = jda_uni32(cf0);
    const uint32_t c1 = nb ? (uint32_t)__builtin_amdgcn_readlane((int)cf1, (int)(nb - 1u)) : c0;
    const uint32_t n_items = chunked ? c1 - c0 : 0u;
    const uint32_t JDA_GLOBAL *cont = JDA_G(const uint32_t, D.blk_cont) + c0;
    uint32_t ent[JDA_P1C_PRELOADED];
#pragma unroll
    for (uint32_t p = 0; p < JDA_P1C_PRELOADED; p++) { ent[p] = 0; if (64u * p + lane < n_items) ent[p] = cont[64u * p + lane]; }
    const jda_p1c_own own = jda_p1c_block<MODE>(D, C, in, LP, tab, wl, win, win_cap, chunked);
    JDA_WAVE_SYNC();
    for (uint32_t p = 0; 64u * p < n_items; p++) {               // uniform trip count
        uint32_t entry = p == 0u ? ent[0] : (p == 1u ? ent[1] : 0u);
        const bool valid = 64u * p + lane < n_items;
        if (p >= JDA_P1C_PRELOADED && valid) entry = cont[64u * p + lane];
        const uint32_t bl = valid ? (JDA_CONT_G7(entry) - C.first_block) & 63u : lane;
        const uint32_t owner_bits = jda_lane_pull(own.bits, bl, nullptr);                   // (every lane takes part)
        jda_p1c_item<MODE>(D, C, entry, bl, owner_bits, valid, tab, wl, win);
    }
    JDA_WAVE_SYNC();
    return jda_p1c_finish<MODE>(own, in.lb, wl);
}

// big: the workgroup's wavefronts, a wavefront's share of the LDS and its scan window are jda_lds_layout<MODE, 0>'s (as many
// wavefronts as fit) or <MODE, 1>'s (one less, its LDS shared out as window: high-bitrate images), chosen per launch list.
// FAST: 1 = every multiply in 24 bits (the plain-case kernels' lists hold only such images), -1 = as the image's descriptor says
// (the general kernels: a uniform branch around the column stage).
template <int MODE, int FAST, int VARIANT, int CONT = 0>
__global__ __launch_bounds__((64 * jda_lds_layout<MODE, 0>::WAVES))
void jda_decode_tiles_persistent(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ tiles, uint32_t n_quads, uint32_t big)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef jda_lds_layout<MODE, 0> L;
    typedef jda_lds_layout<MODE, 1> LB;
    // (one live SGPR: the three numbers are selects between constants wherever they are used)
#define n_waves (big ? (uint32_t)LB::WAVES : (uint32_t)L::WAVES)
#define wave_bytes (big ? (uint32_t)LB::WAVE_BYTES : (uint32_t)L::WAVE_BYTES)
#define win_bytes (big ? (uint32_t)LB::WIN_BYTES : (uint32_t)L::WIN_BYTES)
    static_assert(jda_lds_layout<MODE, 0>::WIN_CHUNKS == jda_lds_layout<MODE, 1>::WIN_CHUNKS, "both window sizes are staged by the same copy loop");
    typedef jda_chunks<L::WIN_CHUNKS> chunks_t;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // runs of n_quads / grid groups of tiles, the remainder one more each for the first workgroups (rounding the run length up
    // instead left the last workgroups of a 64-image batch with half a run and none: 0.6 % of the kernel's time)
    const uint32_t per = n_quads / gridDim.x, rem = n_quads % gridDim.x;
    const uint32_t q0 = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const uint32_t q_end = q0 + per + (blockIdx.x < rem ? 1u : 0u);
    if (q0 >= q_end) return;
    const uint32_t t_begin = q0 * n_waves, t_end = q_end * n_waves;                       // this workgroup's run of tiles
    unsigned long long *wgtrace = g_jda_wgtrace;
    if (wgtrace && lane == 0) wgtrace[(blockIdx.x * 16u + wave) * 2u] = wall_clock64();
    uint8_t *tab = lds;
    uint8_t *wl = lds + L::TAB_BYTES + wave * wave_bytes;
    uint32_t *ctr = (uint32_t *)(lds + L::TAB_BYTES + n_waves * wave_bytes);                // the run's draw counter
    unsigned long long *tab_src = (unsigned long long *)(ctr + 2);                          // where the next image's tables are (JDA_ADVANCE_TABLES)

    // ---- prologue: the tables of the run's first image, the counter
    if (threadIdx.x == 0) *ctr = t_begin;
    const jda_strip R0 = jda_load_record(tiles + t_begin);
    uint32_t staged = R0.ord;                                                            // image (ordinal) whose tables are in LDS
    const uint32_t last_ord = jda_load_record(tiles + (t_end - 1u)).ord;
    jda_dev_desc Dc = jda_desc_uniform<VARIANT, MODE>(descs + R0.image);
    jda_p0_tables(Dc, threadIdx.x, 64u * n_waves, tab, L::LONG_LDS != 0);
    __syncthreads();                                  // tables staged, counter set

    uint32_t i_cur = jda_draw_tile<MODE>(ctr, lane);
    if (i_cur >= t_end) {                             // more wavefronts than tiles: only keep the barriers company
        jda_strip none = R0;
        JDA_ADVANCE_TABLES(last_ord, false, none, Dc);
        return;
    }
    uint32_t i_nxt = jda_draw_tile<MODE>(ctr, lane);
    jda_strip S = jda_load_record(tiles + i_cur);
    if (S.image != R0.image) Dc = jda_desc_uniform<VARIANT, MODE>(descs + S.image);
    jda_p1_inputs in;
    jda_tile_ctx C;
    uint32_t cf0 = 0, cf1 = 0;                        // CONT: the lane's block among the continuation entries
    // everything a tile needs before its P1, fetched with nothing to overlap it (first tile of a wavefront, first
    // tile after an image boundary): index entries -> window bounds -> scan slice into the LDS window
#define JDA_TILE_COLD_START()                                                                                     \
    do {                                                                                                          \
        uint32_t ixe_;                                                                                            \
        jda_issue_index_loads<MODE, CONT>(Dc, S, lane, in, ixe_, &cf0, &cf1);                                     \
        C = jda_tile_setup_from<MODE>(Dc, S, __builtin_amdgcn_readfirstlane(in.ix), __builtin_amdgcn_readfirstlane(ixe_), win_bytes); \
        C.count = __builtin_amdgcn_readfirstlane(C.count);                                                        \
        C.win_lo = __builtin_amdgcn_readfirstlane(C.win_lo);                                                      \
        C.win_len = __builtin_amdgcn_readfirstlane(C.win_len);                                                    \
        jda_window_store<L::WIN_CHUNKS>(wl + L::WIN_OFF, C.win_len, lane, jda_window_load<L::WIN_CHUNKS>(JDA_G(const uint8_t, Dc.scan), C.win_lo, C.win_len, lane)); \
        asm volatile("" : "+v"(in.ix), "+v"(in.pred));   /* nothing in flight when the loop (re)starts */                                \
        if (CONT) asm volatile("" : "+v"(cf0), "+v"(cf1));                                                        \
    } while (0)
    JDA_TILE_COLD_START();
    jda_p4_pre P4;                                    // the colour stage's item addresses for this image (pitch, pixel size)
    jda_p4_prepare<MODE>(P4, Dc, lane);
    if (lane < 8) ((uint32_t *)(wl + L::CNT_OFF))[lane] = 0;
    jda_strip Sn = S;
    if (i_nxt < t_end) Sn = jda_load_record(tiles + i_nxt);
    JDA_ADVANCE_TABLES(S.ord, true, S, Dc);
    jda_lane_pre LP;                                  // the lane's Huffman LUTs / quantiser table / EOB code in this image (reads the staged tables)
    jda_lane_prepare<MODE>(LP, Dc, lane, tab);
    JDA_WAVE_SYNC();

#ifdef JDA_PHASE_TRACE
    unsigned long long *trace = (blockIdx.x % JDA_TRACE_STRIDE == 0) ? g_jda_trace : nullptr;
    uint32_t iter = 0;
#endif
    for (;;) {
        const jda_dev_desc &D = Dc;
        JDA_PTRACE(0);
        const bool have_next = i_nxt < t_end;
        // the software pipeline runs inside an image; the first tile of the next image starts cold (below) -- keeping a
        // second descriptor in SGPRs for that one tile cost more (SGPR spills) than the overlap was worth
        const bool pipelined = have_next && Sn.image == S.image;
        // stage A: draw the tile after the next one and start loading its record (wave-uniform 16 bytes, consumed at
        // the bottom of the loop)
        uint32_t i_nn = t_end;
        if (have_next) i_nn = jda_draw_tile<MODE>(ctr, lane);
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        if (i_nn < t_end) { const uint32_t JDA_GLOBAL *w = JDA_G(const uint32_t, tiles + i_nn); r0 = w[0]; r1 = w[1]; r2 = w[2]; r3 = w[3]; }
        // stage B: per-lane index entries of the next tile; in flight during this tile's entropy phase
        jda_p1_inputs inn;
        uint32_t ixn_end = 0;
        inn.lb = lane; inn.ix = 0; inn.pred = 0; inn.active = false;
        uint32_t cfn0 = 0, cfn1 = 0;
        if (pipelined) jda_issue_index_loads<MODE, CONT>(D, Sn, lane, inn, ixn_end, &cfn0, &cfn1);

        JDA_PTRACE(1);
        // wave priorities: the phase that is a dependent chain (P1) and the one that feeds the memory pipe (P4) go first,
        // the arithmetic-dense IDCT fills the issue slots they leave (measured: 0.6-1 % over "oldest wave first")
        __builtin_amdgcn_s_setprio(JDA_PRIO_P1);
        const uint32_t p1flags = (JDA_EXP_SKIP & 16) ? 0u : (CONT ? jda_p1_chunked<MODE>(D, C, in, cf0, cf1, LP, tab, wl, wl + L::WIN_OFF, win_bytes, lane)
                                                                  : jda_p1_entropy<MODE>(D, C, in, LP, tab, wl, wl + L::WIN_OFF, win_bytes));
        if (D.scale_shift < 2 && !(JDA_EXP_SKIP & 8)) jda_p1_lists<MODE>(D, LP, lane, p1flags, nullptr, tab, wl);
        __builtin_amdgcn_s_setprio(JDA_PRIO_IDCT);
        JDA_WAVE_SYNC();
        JDA_PTRACE(2);

        // stage C: index entries are here -> window bounds -> the next tile's scan slice (HBM -> registers),
        // in flight during the column stage
        jda_tile_ctx Cn = C;
        // the index loads and the record have landed: settle their waits HERE.  Left to the compiler, the record's wait
        // lands after P4 (where it is consumed) as s_waitcnt vmcnt(0) -- the counter is shared with stores on gfx9, so the
        // wavefront would sit out the write acknowledgements of its own tile before starting the next one
        asm volatile("" : "+v"(inn.ix), "+v"(inn.pred), "+v"(ixn_end), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        if (CONT) asm volatile("" : "+v"(cfn0), "+v"(cfn1));
        if (pipelined) {
            Cn = jda_tile_setup_from<MODE>(D, Sn, __builtin_amdgcn_readfirstlane(inn.ix), __builtin_amdgcn_readfirstlane(ixn_end), win_bytes);
            Cn.count = __builtin_amdgcn_readfirstlane(Cn.count);
            Cn.win_lo = __builtin_amdgcn_readfirstlane(Cn.win_lo);
            Cn.win_len = __builtin_amdgcn_readfirstlane(Cn.win_len);
        }
        // (asked for whether or not a next tile of this image exists -- then it is this tile's slice once more, and nothing is stored:
        // a load under a condition is a load the compiler waits for where it merges the two paths)
        chunks_t chunk = jda_window_load<L::WIN_CHUNKS>(JDA_G(const uint8_t, D.scan), Cn.win_lo, Cn.win_len, lane);

        JDA_PTRACE(3);
        if (D.scale_shift < 2 && !(JDA_EXP_SKIP & 4)) {
            if (FAST > 0 || (FAST < 0 && D.fast_mul)) jda_p2_columns<MODE, true>(D, lane, tab, wl);
            else jda_p2_columns<MODE, false>(D, lane, tab, wl);
            JDA_WAVE_SYNC();
        }
        JDA_PTRACE(4);

        // stage D: scan slice -> the LDS window (this tile's P1, its only reader, is over).  The load is settled for every
        // lane, also those that store nothing: a load the compiler still counts as pending at the loop's back edge costs
        // an s_waitcnt vmcnt(0) at the top of the next tile, i.e. behind this tile's output stores
#pragma unroll
        for (int k = 0; k < L::WIN_CHUNKS; k++) asm volatile("" : "+v"(chunk.c[k].w[0]), "+v"(chunk.c[k].w[1]), "+v"(chunk.c[k].w[2]), "+v"(chunk.c[k].w[3]));
        if (pipelined) jda_window_store<L::WIN_CHUNKS>(wl + L::WIN_OFF, Cn.win_len, lane, chunk);

        JDA_PTRACE(5);
        if (D.scale_shift < 2 && !(JDA_EXP_SKIP & 2)) {
            jda_p3_rows<MODE>(D, lane, tab, wl);
            JDA_WAVE_SYNC();
        }
        JDA_PTRACE(6);
        if (lane < 8) ((uint32_t *)(wl + L::CNT_OFF))[lane] = 0;      // list counters reset for the next tile

        __builtin_amdgcn_s_setprio(JDA_PRIO_P4);
        if (!(JDA_EXP_SKIP & 1)) jda_p4_output<MODE>(D, S, C, lane, wl, P4);
        JDA_PTRACE(7);
#ifdef JDA_PHASE_TRACE
        iter++;
#endif
        if (!have_next) break;
        S = Sn; i_nxt = i_nn;
        Sn = jda_unpack_record(r0, r1, r2, r3);
        if (pipelined) { C = Cn; in = inn; cf0 = cfn0; cf1 = cfn1; }
        else {                                        // image boundary: every wavefront of the workgroup passes here once
            Dc = jda_desc_uniform<VARIANT, MODE>(descs + S.image);
            JDA_ADVANCE_TABLES(S.ord, true, S, Dc);
            JDA_WAVE_SYNC();
            JDA_TILE_COLD_START();
            jda_p4_prepare<MODE>(P4, Dc, lane);
            jda_lane_prepare<MODE>(LP, Dc, lane, tab);
        }
        JDA_WAVE_SYNC();
    }
#undef JDA_TILE_COLD_START
    JDA_ADVANCE_TABLES(last_ord, false, S, Dc);        // boundaries after this wavefront's last tile
    if (wgtrace && lane == 0) wgtrace[(blockIdx.x * 16u + wave) * 2u + 1u] = wall_clock64();
#undef n_waves
#undef wave_bytes
#undef win_bytes
}

template <int MODE, int FAST, int VARIANT, int CONT = 0>
static hipError_t launch_persistent(int big, const jda_dev_desc *descs, const jda_strip *tiles, uint32_t n_tiles, hipStream_t stream)
{
    typedef jda_lds_layout<MODE, 0> L0;
    typedef jda_lds_layout<MODE, 1> L1;
    static_assert(L0::TAB_BYTES + L0::WAVES * L0::WAVE_BYTES + 16 <= 160 * 1024 && L1::TAB_BYTES + L1::WAVES * L1::WAVE_BYTES + 16 <= 160 * 1024, "one workgroup must fit the CU's LDS");
    static_assert(L0::WIN_BYTES >= L0::COLLIST_ENTRIES * 2 + 16 && L0::WIN_BYTES % 16 == 0 && L1::WIN_BYTES % 16 == 0 && L0::WIN_OFF % 16 == 0, "the window covers the column list and its overrun");
    const uint32_t n_waves = big ? L1::WAVES : L0::WAVES, wave_bytes = big ? L1::WAVE_BYTES : L0::WAVE_BYTES, win_bytes = big ? L1::WIN_BYTES : L0::WIN_BYTES;
    const int lds_max = (L0::WAVES * L0::WAVE_BYTES > L1::WAVES * L1::WAVE_BYTES ? L0::WAVES * L0::WAVE_BYTES : L1::WAVES * L1::WAVE_BYTES) + L0::TAB_BYTES + 16;
    const int lds_bytes = L0::TAB_BYTES + (int)(n_waves * wave_bytes) + 16;      // + the draw counter
    static std::atomic<unsigned long long> attr_done(0);
    { const hipError_t e = jda_ensure_lds_limit((const void *)jda_decode_tiles_persistent<MODE, FAST, VARIANT, CONT>, lds_max, attr_done); if (e != hipSuccess) return e; }
    static std::atomic<int> grid_cap_once(0);                   // (the GPUs of a node are alike: the first device's CU count)
    int grid_cap = grid_cap_once.load(std::memory_order_relaxed);
    if (!grid_cap) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int per_cu = (160 * 1024) / lds_max;
        // more workgroups than CUs: the ones that do not fit start as others finish, so the hardware deals the second half of the
        // work out by who is done first (the static split left the CUs finishing up to 4 % apart: profiles/r01_final_wg_balance.txt)
        static const int mult = []() { const char *e = JDA_LAB_ENV("JDA_GRID_MULT"); const int m = e ? atoi(e) : JDA_GRID_MULT_DEFAULT; return m < 1 ? 1 : (m > 16 ? 16 : m); }();
        // (measuring: CUs left to the pre-scan's latency-bound rounds while a decode kernel runs)
        static const int spare = []() { const char *e = JDA_LAB_ENV("JDA_DECODE_SPARE_CUS"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v; }();
        if (spare < cus) cus -= spare;
        grid_cap = cus * (per_cu > 0 ? per_cu : 1) * mult;
        grid_cap_once.store(grid_cap, std::memory_order_relaxed);
    }
    const uint32_t n_quads = n_tiles / n_waves;
    const uint32_t grid = n_quads < (uint32_t)grid_cap ? n_quads : (uint32_t)grid_cap;
    JDA_LAUNCH((jda_decode_tiles_persistent<MODE, FAST, VARIANT, CONT>), dim3(grid), dim3(64 * n_waves), lds_bytes, stream,
               descs, tiles, n_quads, (uint32_t)(big ? 1 : 0));
    (void)win_bytes;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The tile lists of whole images (jda_append_strips' records, jda_plan.h), written on the device: thread = tile
__global__ __launch_bounds__(256)
void jda_fill_strips(const jda_strips_params *__restrict__ params)
{
    const jda_strips_params P = params[blockIdx.y];
    const uint32_t per_row = (P.mcus_x + P.per - 1u) / P.per, n_real = P.mcus_y * per_row;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < P.n_padded; k += gridDim.x * 256u) {
        uint32_t w1 = 0, w2 = 0;
        if (k < n_real) {
            const uint32_t y = k / per_row, x = (k - y * per_row) * P.per;
            const uint32_t count = P.mcus_x - x < P.per ? P.mcus_x - x : P.per;
            w1 = y | (x << 16); w2 = count | (k == 0u ? 1u << 8 : 0u);      // mcu_y | mcu_x0 << 16, count | first << 8
        }
        jda_store_u32x4(P.dst + k, P.image, w1, w2, P.ord);
    }
}
extern "C" hipError_t jda_launch_fill_strips(const jda_strips_params *params, uint32_t n_images, uint32_t max_tiles, hipStream_t stream)
{
    if (n_images == 0 || max_tiles == 0) return hipSuccess;
    uint32_t gx = (max_tiles + 1023u) / 1024u;                        // four records a thread
    if (gx > 64u) gx = 64u;
    JDA_LAUNCH(jda_fill_strips, dim3(gx, n_images), dim3(256), 0, stream, params);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The DC thumbnail: 1/8 scale (every progressive file's DC scan included).  A pixel is its block's DC term -- the range-limited
// (DC x q0) >> 5 of jpeg.inl:5146-5154 -- so with index format 2 (the pre-scan stored every block's own DC value) nothing but
// 2 bytes per block is read: no scan, no index entry, no tables in LDS, no IDCT.  A wavefront takes JDA_THUMB_TILES consecutive
// tiles of the list (a tile = the decode kernel's: <= 64 consecutive blocks of one MCU row; crop-aware lists stay what they are),
// lane = block for the load, lane = output pixel for the store; the samples of a pixel's MCU come from their lanes by ds_bpermute.
// Output as JPEGPutMCU* at bThumbnail: (MCU_W / 8) x (MCU_H / 8) pixels per MCU, chroma shared by the MCU.
#ifndef JDA_THUMB_TILES
#define JDA_THUMB_TILES 8u
#endif
// what a tile's lanes load: the DC value of lane's block (0 behind the tile's decoded blocks); count: the tile's MCUs that are decoded
template <int MODE>
__device__ __forceinline__ int32_t jda_thumb_load(const jda_dev_desc &D, const jda_strip &S, uint32_t lane, uint32_t &count)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t first_mcu = S.mcu_y * D.mcus_x + S.mcu_x0;
    count = S.count;
    if (first_mcu >= D.n_mcus_ok) count = 0;                             // MCUs behind a bad one are not decoded (jpeg.inl:5354-5356)
    else if (first_mcu + count > D.n_mcus_ok) count = D.n_mcus_ok - first_mcu;
    int32_t dc = 0;
    if (lane < count * (uint32_t)T::NBLK) dc = JDA_G(const int16_t, D.blk_dc)[first_mcu * (uint32_t)T::NBLK + lane];
    return dc;
}
// .. and what they store: lane = pixel p of the tile's (count * mw) x mh pixels, row-major
template <int MODE>
__device__ __forceinline__ void jda_thumb_store(const jda_dev_desc &D, const jda_strip &S, uint32_t lane, uint32_t count, int32_t dc, int32_t q0)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t mw = (uint32_t)T::MCU_W >> 3, mh = (uint32_t)T::MCU_H >> 3;      // an MCU's pixels: 1 x 1, 2 x 2, 2 x 1, 1 x 2
    const uint32_t sample = jda_range_limit5(dc * q0);                   // the block's 64 samples, all this one (:5146-5154)
    const uint32_t tile_w = count * mw;
    const uint32_t row = (mh == 2u && lane >= tile_w) ? 1u : 0u, x = lane - row * tile_w;
    const uint32_t m = mw == 2u ? x >> 1 : x, px = mw == 2u ? x & 1u : 0u;
    // which luma block of the MCU (jda_fetch at shift 3: 4:2:0 (py, px) -> 2 py + px; 4:2:2 px; 4:4:0 py)
    const uint32_t q = MODE == JDA_MODE_420 ? 2u * row + px : (MODE == JDA_MODE_422 ? px : (MODE == JDA_MODE_440 ? row : 0u));
    const uint32_t yl = m * (uint32_t)T::NBLK + q;
    const uint32_t y = MODE == JDA_MODE_GRAY ? sample : (uint32_t)__builtin_amdgcn_ds_bpermute((int)(yl << 2), (int)sample);
    uint32_t cb = 128u, cr = 128u;
    if (MODE != JDA_MODE_GRAY) {
        cb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((m * (uint32_t)T::NBLK + (uint32_t)T::NLUMA) << 2), (int)sample);
        cr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((m * (uint32_t)T::NBLK + (uint32_t)T::NLUMA + 1u) << 2), (int)sample);
    }
    const uint32_t X = S.mcu_x0 * mw + x, Y = S.mcu_y * mh + row;
    if (lane >= tile_w * mh || X >= D.out_w || Y >= D.out_rows) return;
    uint8_t JDA_GLOBAL *rowp = JDA_G(uint8_t, D.out) + (size_t)Y * D.out_pitch;
    const int pt = D.pixel_type;
    if (pt == JDA_EIGHT_BIT_GRAYSCALE) rowp[X] = (uint8_t)y;
    else if (MODE == JDA_MODE_GRAY) ((uint16_t JDA_GLOBAL *)rowp)[X] = (uint16_t)jda_gray_565(y, pt != JDA_RGB565_LITTLE_ENDIAN);      // JPEGPutMCUGray
    else {
        jda_ycc p;
        p.y = (int32_t)(y << 12); p.cb = (int32_t)cb; p.cr = (int32_t)cr;
        if (pt == JDA_RGB8888) ((jda_u32_alias JDA_GLOBAL *)rowp)[X] = jda_pixel_rgba(p);
        else ((uint16_t JDA_GLOBAL *)rowp)[X] = (uint16_t)jda_pixel_565(p, pt == JDA_RGB565_BIG_ENDIAN);
    }
}
#ifndef JDA_THUMB_WAVES
#define JDA_THUMB_WAVES 4u
#endif
template <int MODE>
__global__ __launch_bounds__(64 * JDA_THUMB_WAVES)
void jda_dc_thumbnail(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ tiles, uint32_t n_tiles)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t0 = jda_uni32((blockIdx.x * JDA_THUMB_WAVES + (threadIdx.x >> 6)) * JDA_THUMB_TILES);
    if (t0 >= n_tiles) return;
    const uint32_t n_here = n_tiles - t0 < JDA_THUMB_TILES ? n_tiles - t0 : JDA_THUMB_TILES;
    // the run's records through the scalar cache (uniform addresses): 16 bytes each
    const uint32_t __attribute__((address_space(4))) *rw = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(tiles + t0);
    jda_strip S[JDA_THUMB_TILES];
    uint32_t w0[JDA_THUMB_TILES], w1[JDA_THUMB_TILES], w2[JDA_THUMB_TILES];
#pragma unroll
    for (uint32_t k = 0; k < JDA_THUMB_TILES; k++) {                      // (behind the list's end: the last record again -- every load asked for before the first wait)
        const uint32_t kk = k < n_here ? k : n_here - 1u;
        w0[k] = rw[4u * kk]; w1[k] = rw[4u * kk + 1u]; w2[k] = rw[4u * kk + 2u];
    }
    bool one_image = true;
#pragma unroll
    for (uint32_t k = 0; k < JDA_THUMB_TILES; k++) {
        S[k].image = w0[k]; S[k].mcu_y = (uint16_t)(w1[k] & 0xffffu); S[k].mcu_x0 = (uint16_t)(w1[k] >> 16);
        S[k].count = k < n_here ? (uint8_t)(w2[k] & 0xffu) : (uint8_t)0; S[k].first = 0; S[k].pad_ = 0; S[k].ord = 0;
        one_image = one_image && w0[k] == w0[0];
    }
    const uint32_t b_in_mcu = lane % (uint32_t)T::NBLK, comp = b_in_mcu < (uint32_t)T::NLUMA ? 0u : b_in_mcu - (uint32_t)T::NLUMA + 1u;
    uint32_t cur_image = S[0].image;
    jda_dev_desc D = jda_desc_const<0, MODE>(jda_desc_at(descs, cur_image));
    // the lane's quantiser's first entry: one of three scalar loads
#define JDA_THUMB_Q0(D_) do {                                                                                                                  \
        const uint32_t __attribute__((address_space(4))) *qt_ = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)((D_).tables + JDA_TB_QUANT); \
        const int32_t qa_ = (int16_t)qt_[(D_).q_id[0] * 32u], qb_ = (int16_t)qt_[(D_).q_id[1] * 32u], qc_ = (int16_t)qt_[(D_).q_id[2] * 32u];            \
        q0 = comp == 0u ? qa_ : (comp == 1u ? qb_ : qc_);                                                                                      \
    } while (0)
    int32_t q0;
    JDA_THUMB_Q0(D);
    if (one_image) {
        // the run is one image's (all but the runs across an image boundary of the list): every tile's load is asked for before the first is used
        int32_t dcv[JDA_THUMB_TILES];
        uint32_t cnt[JDA_THUMB_TILES];
        if (MODE == JDA_MODE_GRAY && D.pixel_type == JDA_EIGHT_BIT_GRAYSCALE && JDA_THUMB_TILES % 4u == 0u) {
            // a gray file to 8-bit gray: four whole tiles side by side are 256 consecutive DC values and 256 consecutive pixels -- a lane
            // takes four of each (one 8-byte load, one dword store) and makes the samples two at a time in the halves of a word: bits 14:5
            // of DC x q0 are all ucRangeTable[(DC x q0 >> 5) & 0x3ff] looks at (jpeg.inl:5146-5154), so the 16-bit product is enough
            typedef unsigned short jda_us2 __attribute__((ext_vector_type(2)));
            constexpr uint32_t G = JDA_THUMB_TILES / 4u;
            bool quad[G];
            uint32_t first[G];
            uint64_t four[G];
#pragma unroll
            for (uint32_t g = 0; g < G; g++) {
                const jda_strip &A = S[4u * g];
                first[g] = A.mcu_y * D.mcus_x + A.mcu_x0;
                bool ok = (first[g] & 3u) == 0u && first[g] + 256u <= D.n_mcus_ok && A.mcu_x0 + 256u <= D.out_w && A.mcu_y < D.out_rows;
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) ok = ok && S[4u * g + k].count == 64u && S[4u * g + k].mcu_y == A.mcu_y && S[4u * g + k].mcu_x0 == A.mcu_x0 + 64u * k;
                quad[g] = ok;
            }
#pragma unroll
            for (uint32_t g = 0; g < G; g++) {
                four[g] = 0;
                if (quad[g]) four[g] = JDA_G(const uint64_t, D.blk_dc)[(first[g] >> 2) + lane];
                else {
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++) dcv[4u * g + k] = jda_thumb_load<MODE>(D, S[4u * g + k], lane, cnt[4u * g + k]);
                }
            }
            const uint32_t qq = ((uint32_t)q0 & 0xffffu) | ((uint32_t)q0 << 16);
#pragma unroll
            for (uint32_t g = 0; g < G; g++) {
                if (quad[g]) {
                    const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, (uint32_t)four[g]) * __builtin_bit_cast(jda_us2, qq));
                    const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, (uint32_t)(four[g] >> 32)) * __builtin_bit_cast(jda_us2, qq));
                    const uint32_t s01 = jda_sat_pk_u8(jda_pk_add16(jda_pk_sext10_at5(p01), 0x00800080u));
                    const uint32_t s23 = jda_sat_pk_u8(jda_pk_add16(jda_pk_sext10_at5(p23), 0x00800080u));
                    uint8_t JDA_GLOBAL *o = JDA_G(uint8_t, D.out) + ((size_t)S[4u * g].mcu_y * D.out_pitch + S[4u * g].mcu_x0 + 4u * lane);
                    *(jda_u32_alias JDA_GLOBAL *)o = s01 | (s23 << 16);
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++)
                        if (cnt[4u * g + k]) jda_thumb_store<MODE>(D, S[4u * g + k], lane, cnt[4u * g + k], dcv[4u * g + k], q0);
                }
            }
            return;
        }
#pragma unroll
        for (uint32_t k = 0; k < JDA_THUMB_TILES; k++) dcv[k] = jda_thumb_load<MODE>(D, S[k], lane, cnt[k]);
#pragma unroll
        for (uint32_t k = 0; k < JDA_THUMB_TILES; k++)
            if (cnt[k]) jda_thumb_store<MODE>(D, S[k], lane, cnt[k], dcv[k], q0);
        return;
    }
#pragma unroll
    for (uint32_t k = 0; k < JDA_THUMB_TILES; k++) {
        if (S[k].count == 0) continue;                                   // (padding entry, or behind the list's end)
        if (S[k].image != cur_image) {
            cur_image = S[k].image;
            D = jda_desc_const<0, MODE>(jda_desc_at(descs, cur_image));
            JDA_THUMB_Q0(D);
        }
        uint32_t count;
        const int32_t dc = jda_thumb_load<MODE>(D, S[k], lane, count);
        if (count) jda_thumb_store<MODE>(D, S[k], lane, count, dc, q0);
    }
#undef JDA_THUMB_Q0
}
template <int MODE>
static hipError_t launch_dc_thumbnail(const jda_dev_desc *descs, const jda_strip *tiles, uint32_t n_tiles, hipStream_t stream)
{
    const uint32_t per_wg = JDA_THUMB_WAVES * JDA_THUMB_TILES;
    JDA_LAUNCH((jda_dc_thumbnail<MODE>), dim3((n_tiles + per_wg - 1u) / per_wg), dim3(64 * JDA_THUMB_WAVES), 0, stream, descs, tiles, n_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The same for a WHOLE gray image on a row-major surface (no crop, no strips): the image's 1/8 thumbnail is a pointwise map of its DC
// array -- pixel (X, Y) = f(DC[Y mcus_x + X]) -- so there is nothing to look up per tile: the launch list holds ONE record per image,
// a thread takes four neighbouring blocks of a row (8 bytes in, 4 or 8 bytes out) and several such quads, all loads asked for before
// the first is used.  The tile kernel above spent its time on the records (twelve scalar loads and a chain of compares per four
// tiles: 0.32 ms per 256 x 8192x8192, 2.5 TB/s of the 2 + 1 bytes a block); this one is a copy kernel with a multiply in it.
#define JDA_FLAT_QUADS 8u          // quads a thread takes, a workgroup's width apart
__global__ __launch_bounds__(256)
void jda_dc_thumbnail_flat(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ recs)
{
    typedef unsigned short jda_us2 __attribute__((ext_vector_type(2)));
    const uint32_t __attribute__((address_space(4))) *rw = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(recs + blockIdx.y);
    const jda_dev_desc D = jda_desc_const<0, JDA_MODE_GRAY>(jda_desc_at(descs, rw[0]));
    const uint32_t __attribute__((address_space(4))) *qt = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(D.tables + JDA_TB_QUANT);
    const int32_t q0 = (int16_t)qt[D.q_id[0] * 32u];
    const uint32_t Q = (D.mcus_x + 3u) >> 2, n_items = Q * D.mcus_y;          // quads per row, in the image
    const uint32_t base = blockIdx.x * (256u * JDA_FLAT_QUADS) + threadIdx.x;
    if (base >= n_items) return;
    const bool aligned = (D.mcus_x & 3u) == 0u;                          // a quad's four DC values are one aligned 8-byte word
    const float rq = 1.0f / (float)Q;
    const int pt = D.pixel_type;
    uint64_t four[JDA_FLAT_QUADS];
    uint32_t X0[JDA_FLAT_QUADS], Y0[JDA_FLAT_QUADS], nv[JDA_FLAT_QUADS];
#pragma unroll
    for (uint32_t k = 0; k < JDA_FLAT_QUADS; k++) {
        const uint32_t item = base + k * 256u;
        uint32_t y = (uint32_t)((float)item * rq);                       // item / Q (item < 2^24 for every image the index admits; corrected below)
        if (y * Q > item) y--;
        if ((y + 1u) * Q <= item) y++;
        const uint32_t x = (item - y * Q) * 4u;
        X0[k] = x; Y0[k] = y; four[k] = 0; nv[k] = 0;
        if (item >= n_items) continue;
        const uint32_t g = y * D.mcus_x + x;                             // the quad's first block
        uint32_t n = D.mcus_x - x < 4u ? D.mcus_x - x : 4u;              // blocks of the row it holds ..
        n = g >= D.n_mcus_ok ? 0u : (D.n_mcus_ok - g < n ? D.n_mcus_ok - g : n);      // .. that were decoded (jpeg.inl:5354-5356)
        nv[k] = n;
        if (n == 4u && aligned) four[k] = JDA_G(const uint64_t, D.blk_dc)[g >> 2];
        else for (uint32_t e = 0; e < n; e++) four[k] |= (uint64_t)(uint16_t)JDA_G(const int16_t, D.blk_dc)[g + e] << (16u * e);
    }
    const uint32_t qq = ((uint32_t)q0 & 0xffffu) | ((uint32_t)q0 << 16);
#pragma unroll
    for (uint32_t k = 0; k < JDA_FLAT_QUADS; k++) {
        uint32_t n = nv[k];
        if (n == 0u || Y0[k] >= D.out_rows || X0[k] >= D.out_w) continue;
        if (D.out_w - X0[k] < n) n = D.out_w - X0[k];
        // bits 14:5 of DC x q0 are all ucRangeTable[(DC x q0 >> 5) & 0x3ff] looks at (jpeg.inl:5146-5154): the 16-bit product is enough
        const uint32_t p01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, (uint32_t)four[k]) * __builtin_bit_cast(jda_us2, qq));
        const uint32_t p23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, (uint32_t)(four[k] >> 32)) * __builtin_bit_cast(jda_us2, qq));
        const uint32_t s01 = jda_sat_pk_u8(jda_pk_add16(jda_pk_sext10_at5(p01), 0x00800080u));      // samples 0, 1 in bytes 0, 1
        const uint32_t s23 = jda_sat_pk_u8(jda_pk_add16(jda_pk_sext10_at5(p23), 0x00800080u));
        const uint32_t s = s01 | (s23 << 16);
        uint8_t JDA_GLOBAL *rowp = JDA_G(uint8_t, D.out) + (size_t)Y0[k] * D.out_pitch;
        if (pt == JDA_EIGHT_BIT_GRAYSCALE) {
            if (n == 4u) *(jda_u32_alias JDA_GLOBAL *)(rowp + X0[k]) = s;
            else for (uint32_t e = 0; e < n; e++) rowp[X0[k] + e] = (uint8_t)(s >> (8u * e));
        } else {                                                         // JPEGPutMCUGray: usGrayTo565
            const bool be = pt != JDA_RGB565_LITTLE_ENDIAN;
            uint16_t JDA_GLOBAL *o = (uint16_t JDA_GLOBAL *)rowp + X0[k];
            const uint32_t a = jda_gray_565(s & 0xffu, be) | (jda_gray_565((s >> 8) & 0xffu, be) << 16), b = jda_gray_565((s >> 16) & 0xffu, be) | (jda_gray_565(s >> 24, be) << 16);
            if (n == 4u) *(jda_u64_alias JDA_GLOBAL *)o = (uint64_t)a | ((uint64_t)b << 32);
            else for (uint32_t e = 0; e < n; e++) o[e] = (uint16_t)((e < 2u ? a : b) >> (16u * (e & 1u)));
        }
    }
}
static hipError_t launch_dc_thumbnail_flat(const jda_dev_desc *descs, const jda_strip *recs, uint32_t n_images, uint32_t max_items, hipStream_t stream)
{
    const uint32_t per_wg = 256u * JDA_FLAT_QUADS;
    JDA_LAUNCH(jda_dc_thumbnail_flat, dim3((max_items + per_wg - 1u) / per_wg, n_images), dim3(256), 0, stream, descs, recs);
    return hipGetLastError();
}

// .. and a whole 4:2:0 image (every progressive photograph's default decode among them): a thread takes an MCU -- its six DC values are
// twelve consecutive bytes -- and makes the MCU's 2 x 2 pixels as JPEGPutMCU22 does at bThumbnail (jpeg.inl:3627-3663: luma block q = 2 y + x
// of the MCU, chroma shared), two 8-byte stores for RGB8888; consecutive threads take consecutive MCUs of a row.
__global__ __launch_bounds__(256)
void jda_dc_thumbnail_flat420(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ recs)
{
    const uint32_t __attribute__((address_space(4))) *rw = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(recs + blockIdx.y);
    const jda_dev_desc D = jda_desc_const<0, JDA_MODE_420>(jda_desc_at(descs, rw[0]));
    const uint32_t __attribute__((address_space(4))) *qt = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(D.tables + JDA_TB_QUANT);
    const int32_t qy = (int16_t)qt[D.q_id[0] * 32u], qb = (int16_t)qt[D.q_id[1] * 32u], qr = (int16_t)qt[D.q_id[2] * 32u];
    const uint32_t n_items = D.mcus_x * D.mcus_y;
    const uint32_t n_ok = D.n_mcus_ok < n_items ? D.n_mcus_ok : n_items;       // MCUs behind a bad one are not decoded (jpeg.inl:5354-5356)
    const uint32_t base = blockIdx.x * (256u * JDA_FLAT_QUADS) + threadIdx.x;
    if (base >= n_ok) return;
    const float rq = 1.0f / (float)D.mcus_x;
    const int pt = D.pixel_type;
    uint32_t w0[JDA_FLAT_QUADS], w1[JDA_FLAT_QUADS], w2[JDA_FLAT_QUADS];
#pragma unroll
    for (uint32_t k = 0; k < JDA_FLAT_QUADS; k++) {                      // (all loads asked for before the first is used)
        const uint32_t item = base + k * 256u;
        w0[k] = w1[k] = w2[k] = 0;
        if (item < n_ok) { const uint32_t JDA_GLOBAL *d = JDA_G(const uint32_t, D.blk_dc) + item * 3u; w0[k] = d[0]; w1[k] = d[1]; w2[k] = d[2]; }
    }
#pragma unroll
    for (uint32_t k = 0; k < JDA_FLAT_QUADS; k++) {
        const uint32_t item = base + k * 256u;
        if (item >= n_ok) continue;
        uint32_t my = (uint32_t)((float)item * rq);
        if (my * D.mcus_x > item) my--;
        if ((my + 1u) * D.mcus_x <= item) my++;
        const uint32_t mx = item - my * D.mcus_x;
        const uint32_t y00 = jda_range_limit5((int32_t)(int16_t)w0[k] * qy), y01 = jda_range_limit5((int32_t)(int16_t)(w0[k] >> 16) * qy);
        const uint32_t y10 = jda_range_limit5((int32_t)(int16_t)w1[k] * qy), y11 = jda_range_limit5((int32_t)(int16_t)(w1[k] >> 16) * qy);
        const uint32_t cb = jda_range_limit5((int32_t)(int16_t)w2[k] * qb), cr = jda_range_limit5((int32_t)(int16_t)(w2[k] >> 16) * qr);
        const uint32_t X = 2u * mx, Y = 2u * my;
#pragma unroll
        for (uint32_t r = 0; r < 2u; r++) {
            if (Y + r >= D.out_rows || X >= D.out_w) continue;
            const uint32_t ya = r ? y10 : y00, yb = r ? y11 : y01;
            const bool two = X + 1u < D.out_w;
            uint8_t JDA_GLOBAL *rowp = JDA_G(uint8_t, D.out) + (size_t)(Y + r) * D.out_pitch;
            if (pt == JDA_EIGHT_BIT_GRAYSCALE) {                         // (luma only: JPEGPutMCU8BitGray)
                if (two) *(uint16_t JDA_GLOBAL *)(rowp + X) = (uint16_t)(ya | (yb << 8)); else rowp[X] = (uint8_t)ya;
            } else {
                jda_ycc a, b;
                a.y = (int32_t)(ya << 12); a.cb = (int32_t)cb; a.cr = (int32_t)cr;
                b.y = (int32_t)(yb << 12); b.cb = (int32_t)cb; b.cr = (int32_t)cr;
                if (pt == JDA_RGB8888) {
                    const uint32_t pa = jda_pixel_rgba(a), pb = jda_pixel_rgba(b);
                    if (two) *(jda_u64_alias JDA_GLOBAL *)(rowp + 4u * X) = (uint64_t)pa | ((uint64_t)pb << 32); else *(jda_u32_alias JDA_GLOBAL *)(rowp + 4u * X) = pa;
                } else {
                    const bool be = pt == JDA_RGB565_BIG_ENDIAN;
                    const uint32_t pa = jda_pixel_565(a, be) & 0xffffu, pb = jda_pixel_565(b, be) & 0xffffu;
                    if (two) *(jda_u32_alias JDA_GLOBAL *)(rowp + 2u * X) = pa | (pb << 16); else *(uint16_t JDA_GLOBAL *)(rowp + 2u * X) = (uint16_t)pa;
                }
            }
        }
    }
}
static hipError_t launch_dc_thumbnail_flat420(const jda_dev_desc *descs, const jda_strip *recs, uint32_t n_images, uint32_t max_items, hipStream_t stream)
{
    const uint32_t per_wg = 256u * JDA_FLAT_QUADS;
    JDA_LAUNCH(jda_dc_thumbnail_flat420, dim3((max_items + per_wg - 1u) / per_wg, n_images), dim3(256), 0, stream, descs, recs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 1/4 scale (jda_q4_* in jda_device_core.h): a block is its DC value, at most four AC symbols and a 2x2 IDCT -- the persistent decode
// kernel spent its time on what surrounds that (window staging, slots, lists, one workgroup per CU for its LDS).  Here a workgroup is
// four wavefronts with the image's tables in LDS and nothing else, so a CU holds as many as its registers allow; it takes a run of the
// launch list's groups (a group = the tiles the decode kernel's workgroup would take: one image's, so the tables are staged when the
// image changes and not per group), a wavefront every fourth tile of a group.  Per tile: lane = block -- index entry and DC value, the
// five dwords of the scan behind the entry, all four tiles' loads asked for before the first is used --, then lane = output pixel.
#define JDA_Q4_TILES 4u
template <int MODE>
__global__ __launch_bounds__(256)
void jda_quarter_tiles(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ tiles, uint32_t n_groups, uint32_t group_tiles)
{
    typedef jda_mode_traits<MODE> T;
    __shared__ __attribute__((aligned(16))) uint8_t tab[JDA_LT_BYTES];
    const uint32_t wave = jda_uni32(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t per = n_groups / gridDim.x, rem = n_groups % gridDim.x;
    const uint32_t g0 = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const uint32_t g_end = g0 + per + (blockIdx.x < rem ? 1u : 0u);
    uint32_t staged = 0xffffffffu;
    jda_dev_desc D;
    jda_lane_pre LP;
    jda_q4_quant Q;
    Q.q0 = Q.q1 = Q.q8 = Q.q9 = 0;
    bool dc_only = false, skip = false;
    for (uint32_t g = g0; g < g_end; g++) {
        const uint32_t __attribute__((address_space(4))) *rw = (const uint32_t __attribute__((address_space(4))) *)(uintptr_t)(tiles + (size_t)g * group_tiles);
        const uint32_t image = rw[0];                                    // (a group is one image's: jda_append_strips pads per image)
        if (image != staged) {                                           // the same for every wavefront of the workgroup
            __syncthreads();                                             // nobody reads the old tables any more
            D = jda_desc_const<0, MODE>(jda_desc_at(descs, image));
            jda_p0_tables(D, threadIdx.x, 256, tab, true);
            __syncthreads();
            jda_lane_prepare<MODE>(LP, D, lane, tab);
            const int16_t *quant = (const int16_t *)(tab + LP.quant_off);
            Q.q0 = quant[0]; Q.q1 = quant[1]; Q.q8 = quant[8]; Q.q9 = quant[9];
            dc_only = (D.pad_[0] & JDA_DESC_DC_ONLY) != 0;               // the DC scan of a progressive file: no AC symbol exists
            skip = MODE != JDA_MODE_GRAY && D.gray_from_color && LP.chroma;      // :5225-5233 chroma never decoded
            staged = image;
        }
        // this wavefront's tiles of the group: wave, wave + 4, ..
        jda_strip S[JDA_Q4_TILES];
        uint32_t cnt[JDA_Q4_TILES], ix[JDA_Q4_TILES];
        int32_t dc[JDA_Q4_TILES];
        jda_q4_bits B[JDA_Q4_TILES];
#pragma unroll
        for (uint32_t k = 0; k < JDA_Q4_TILES; k++) {
            const uint32_t t = wave + 4u * k, tt = t < group_tiles ? t : 0u;
            const uint32_t w1 = rw[4u * tt + 1u], w2 = rw[4u * tt + 2u];
            S[k].image = image; S[k].mcu_y = (uint16_t)(w1 & 0xffffu); S[k].mcu_x0 = (uint16_t)(w1 >> 16);
            S[k].count = t < group_tiles ? (uint8_t)(w2 & 0xffu) : (uint8_t)0; S[k].first = 0; S[k].pad_ = 0; S[k].ord = 0;
        }
#pragma unroll
        for (uint32_t k = 0; k < JDA_Q4_TILES; k++) {
            const uint32_t first_mcu = S[k].mcu_y * D.mcus_x + S[k].mcu_x0;
            cnt[k] = S[k].count;
            if (first_mcu >= D.n_mcus_ok) cnt[k] = 0;                    // MCUs behind a bad one are not decoded (jpeg.inl:5354-5356)
            else if (first_mcu + cnt[k] > D.n_mcus_ok) cnt[k] = D.n_mcus_ok - first_mcu;
            ix[k] = 0; dc[k] = 0;
            if (lane < cnt[k] * (uint32_t)T::NBLK && !skip) {
                ix[k] = JDA_G(const uint32_t, D.blk_index)[first_mcu * (uint32_t)T::NBLK + lane];
                dc[k] = JDA_G(const int16_t, D.blk_dc)[first_mcu * (uint32_t)T::NBLK + lane];
            }
        }
        // (every lane loads, the idle ones wherever the clamp sends them: a load under a lane condition is merged with the other lanes'
        // zeros by a copy, and the copy waits.  All twenty dwords are settled in one place, before the first tile is worked on:
        // loads and stores share a counter, and a wait the compiler places later would sit behind the stores of the tiles before)
        if (!dc_only) {
#pragma unroll
            for (uint32_t k = 0; k < JDA_Q4_TILES; k++) B[k] = jda_q4_load(D.scan, D.scan_len, ix[k]);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < JDA_Q4_TILES; k++)
#pragma unroll
                for (int i = 0; i < 5; i++) B[k].d[i] = 0;
        }
        asm volatile("" : "+v"(B[0].d[0]), "+v"(B[0].d[1]), "+v"(B[0].d[2]), "+v"(B[0].d[3]), "+v"(B[0].d[4]),
                          "+v"(B[1].d[0]), "+v"(B[1].d[1]), "+v"(B[1].d[2]), "+v"(B[1].d[3]), "+v"(B[1].d[4]));
        asm volatile("" : "+v"(B[2].d[0]), "+v"(B[2].d[1]), "+v"(B[2].d[2]), "+v"(B[2].d[3]), "+v"(B[2].d[4]),
                          "+v"(B[3].d[0]), "+v"(B[3].d[1]), "+v"(B[3].d[2]), "+v"(B[3].d[3]), "+v"(B[3].d[4]));
        const uint16_t *ac = (const uint16_t *)(tab + LP.ac_off);
#pragma unroll
        for (uint32_t k = 0; k < JDA_Q4_TILES; k++) {
            if (cnt[k] == 0) continue;                                   // (uniform: padding, behind the group's end, behind a bad MCU)
            uint32_t px = 0;
            const bool active = lane < cnt[k] * (uint32_t)T::NBLK && !skip;
            const bool trunc = active && (ix[k] & JDA_INDEX_TRUNC) != 0u;
            // (every lane runs the block code, the idle ones on zeros: it has no branch to skip with)
            if (__builtin_amdgcn_ballot_w64(trunc) != 0ull) px = jda_q4_block<true>(ix[k], dc[k], B[k], ac, Q, dc_only, trunc);
            else px = jda_q4_block<false>(ix[k], dc[k], B[k], ac, Q, dc_only, false);
            jda_q4_store<MODE>(D, S[k], cnt[k], lane, px, nullptr);
        }
    }
}
template <int MODE>
static hipError_t launch_quarter(const jda_dev_desc *descs, const jda_strip *tiles, uint32_t n_tiles, uint32_t group_tiles, hipStream_t stream)
{
    const uint32_t n_groups = n_tiles / group_tiles;
    // two rounds of as many workgroups as the GPU holds at once (the registers decide: six or seven of four wavefronts a CU)
    static const uint32_t resident = []() {
        int dev = 0, cus = 256, per_cu = 6;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)jda_quarter_tiles<MODE>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 6;
        return (uint32_t)cus * (uint32_t)per_cu;
    }();
    uint32_t grid = resident * 2u;
    if (grid > n_groups) grid = n_groups;
    if (grid == 0) return hipSuccess;
    JDA_LAUNCH((jda_quarter_tiles<MODE>), dim3(grid), dim3(256), 0, stream, descs, tiles, n_groups, group_tiles);
    return hipGetLastError();
}

// wave-wide maximum / sum of a value of every ACTIVE lane (inactive lanes contribute nothing), the same in every lane
__device__ __forceinline__ uint32_t jda_wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t jda_wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    return v;
}
// ------------------------------------------------------------------------------------------------
// The per-block index on the device (the algorithm is described at jda_seg_walk): one lane per 256-byte segment
// of the filtered scan, read where it lies; a workgroup of four wavefronts around one copy of the walk's tables.
// The walk's tables, once per image (every walker's workgroup converted the blob itself before: 16 dependent rounds of byte
// loads in front of each of them -- a fifth of the latency-bound rounds)
__global__ __launch_bounds__(1024)
void jda_walk_tables_build(const jda_segscan_params *__restrict__ params)
{
    const jda_segscan_params &P = params[blockIdx.x];
    if (P.walk_tables_shared) return;
    jda_walk_tables_from(P.tables, jda_wt_dc_follow(P), threadIdx.x, 1024u, P.walk_tables);
}
extern "C" hipError_t jda_launch_walk_tables(const jda_segscan_params *params, uint32_t n_images, hipStream_t stream)
{
    if (n_images == 0) return hipSuccess;
    JDA_LAUNCH(jda_walk_tables_build, dim3(n_images), dim3(1024), 0, stream, params);
    return hipGetLastError();
}

// one segment of a round: walk it, store its sums, and if its exit state is not what the next segment was entered with, replace that
// and put the next segment on the next round's list (one atomic per wavefront for the places on it)
template <int OP>
__device__ __forceinline__ void jda_fused_item(const jda_segscan_params &P, const uint8_t *tab, uint32_t seg, uint32_t round, uint32_t lane,
                                               uint32_t JDA_GLOBAL *E, uint32_t JDA_GLOBAL *wl_out)
{
    jda_seg_sum S;
    jda_seg_stats ST;
    ST.bad = 0; ST.terminal = 0; ST.max_ac_bits = 0; ST.max_abs_dc = 0; ST.trunc_events = 0; ST.mismatch = 0;
    const uint32_t JDA_GLOBAL *segw = JDA_G(const uint32_t, P.scan) + (size_t)seg * (JDA_SEG_BYTES / 4u);
    const uint32_t entry = (seg == 0 || round == 0) ? 0u : E[seg];       // the scan starts at a block start (jpeg.inl:4996-4998)
    const uint32_t x = P.restart_pos ? jda_seg_walk<OP, true>(P, seg, entry, segw, tab, S, ST, round) : jda_seg_walk<OP, false>(P, seg, entry, segw, tab, S, ST, round);
    if (OP == JDA_SEG_RECORD) {
        uint32_t *o = P.seg_sum + (size_t)seg * JDA_SEG_SUM_WORDS;
        jda_store_u32x4(o, S.nblk, (uint32_t)S.dcsum[0], (uint32_t)S.dcsum[1], (uint32_t)S.dcsum[2]);
        jda_store_u32x4(o + 4, S.phase_map, S.bad | (S.max_ac << 4), S.lag_last, round);
    }
    if (round == 0) { if (seg + 1u < P.n_segs) E[seg + 1u] = x; }       // (nobody reads the entry states in round 0)
    else {
        const bool changed = seg + 1u < P.n_segs && x != E[seg + 1u];
        const uint64_t who = __builtin_amdgcn_ballot_w64(changed);
        if (who) {
            const uint32_t first = (uint32_t)__builtin_ctzll(who);
            uint32_t at0 = 0;
            if (lane == first) at0 = atomicAdd(&P.stats[8u + round + 1u], (uint32_t)__builtin_popcountll(who));
            at0 = (uint32_t)__builtin_amdgcn_readlane((int)at0, (int)first);
            if (changed) {
                E[seg + 1u] = x;
                const uint32_t at = at0 + (uint32_t)__builtin_popcountll(who & ((1ull << lane) - 1ull));
                if (at < P.worklist_cap) wl_out[at] = seg + 1u;
            }
        }
    }
}

// JDA_SEG_SPEC: round 0 (every segment from the guess "a block starts here": exit states only); JDA_SEG_RECORD: the rest.
// LDS_TABLES: the walk's tables staged in LDS (rounds 0 and 1: every segment walks, the lookups are what the round's time is made of)
// or read where jda_walk_tables_build left them, through the L2 (the rounds behind: a few percent of the segments, each a chain of
// dependent steps whose length nothing shortens -- what matters there is what the round keeps OTHERS from doing: a workgroup that holds
// 32 KB of LDS keeps a decode workgroup, which needs the CU's whole LDS, off its CU for the length of a walk; a wavefront of <= 64
// registers and no LDS runs beside one).
#ifndef JDA_WALK_THREADS
#define JDA_WALK_THREADS 256u      // a walker workgroup: the wavefronts that share one copy of the tables in LDS
#endif
#define JDA_ROUND_ALL 0x80000000u     // jda_segscan_fused's round argument: every segment, whatever the round; jda_segscan_tail's first round: exit states only (SPEC)
template <int OP, bool LDS_TABLES>
__global__ __launch_bounds__(JDA_WALK_THREADS)
void jda_segscan_fused(const jda_segscan_params *__restrict__ params, uint32_t round_and_flag)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const jda_segscan_params P = jda_segscan_resolve(params[blockIdx.y]);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t JDA_GLOBAL *stats = JDA_G(uint32_t, P.stats);
    const uint32_t round = round_and_flag & ~JDA_ROUND_ALL;
    const bool all = round <= 1u || (round_and_flag & JDA_ROUND_ALL) != 0u;      // rounds 0 and 1 walk every segment (round 1 to make everybody's sums); the closing RECORD round of the states-first order (jda_launch_prescan_passes_ex) too
    const uint32_t count = all ? P.n_segs : stats[8u + round];
    if (blockIdx.x * JDA_WALK_THREADS >= count) return;                          // (uniform per workgroup)
    const uint32_t JDA_GLOBAL *wl_in = JDA_G(const uint32_t, P.worklist) + ((round & 1u) ? P.worklist_cap : 0u);
    uint32_t JDA_GLOBAL *wl_out = JDA_G(uint32_t, P.worklist) + ((round & 1u) ? 0u : P.worklist_cap);
    uint32_t JDA_GLOBAL *E = JDA_G(uint32_t, P.entry_cur);
    const uint8_t *tab = LDS_TABLES ? (const uint8_t *)lds : (const uint8_t *)JDA_G(const uint8_t, P.walk_tables);
    if (LDS_TABLES) {
        jda_walk_tables_stage(P.walk_tables, threadIdx.x, JDA_WALK_THREADS, lds);
        __syncthreads();                                             // the tables: all that is staged (a walk reads its segment from memory)
    }
    for (uint32_t base = blockIdx.x * JDA_WALK_THREADS + wave * 64u; base < count; base += gridDim.x * JDA_WALK_THREADS) {
        const uint32_t item = base + lane;
        if (item >= count) continue;
        jda_fused_item<OP>(P, tab, all ? item : wl_in[item], round, lane, E, wl_out);
    }
}

// The rounds behind the first few, in ONE launch: one workgroup per image goes round after round (a barrier and a fence between two)
// until a round leaves its list empty -- lists of a handful of segments by then; a launch per round cost more than its walk, and a
// fixed number of launches was a limit on the rounds.  stats[7] = 1: settled (0: max_round reached).
// LDS_TABLES: sixteen wavefronts around the tables in LDS (large images: dozens of segments a round) / four wavefronts, tables through
// the L2 (batches of small images: see jda_segscan_fused).
template <bool LDS_TABLES>
__global__ __launch_bounds__(LDS_TABLES ? 1024 : 256)
void jda_segscan_tail(const jda_segscan_params *__restrict__ params, uint32_t first_round_and_flag, uint32_t max_round)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t nthreads = LDS_TABLES ? 1024u : 256u;
    const jda_segscan_params P = jda_segscan_resolve(params[blockIdx.x]);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *stats = P.stats;
    const bool states_only = (first_round_and_flag & JDA_ROUND_ALL) != 0u;      // (uniform) the states-first order: these rounds settle the entry states, one RECORD round over every segment follows
    uint32_t round = first_round_and_flag & ~JDA_ROUND_ALL;
    uint32_t count = __hip_atomic_load(&stats[8u + round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count == 0) { if (threadIdx.x == 0) stats[7] = 1; return; }       // (the usual case: nothing is staged)
    uint32_t JDA_GLOBAL *E = JDA_G(uint32_t, P.entry_cur);
    const uint8_t *tab = LDS_TABLES ? (const uint8_t *)lds : (const uint8_t *)JDA_G(const uint8_t, P.walk_tables);
    if (LDS_TABLES) {
        jda_walk_tables_stage(P.walk_tables, threadIdx.x, nthreads, lds);
        __syncthreads();
    }
    while (count != 0 && round < max_round) {
        const uint32_t JDA_GLOBAL *wl_in = JDA_G(const uint32_t, P.worklist) + ((round & 1u) ? P.worklist_cap : 0u);
        uint32_t JDA_GLOBAL *wl_out = JDA_G(uint32_t, P.worklist) + ((round & 1u) ? 0u : P.worklist_cap);
        if (count > P.worklist_cap) count = P.worklist_cap;
        for (uint32_t base = wave * 64u; base < count; base += nthreads) {
            const uint32_t item = base + lane;
            if (item >= count) continue;
            if (states_only) jda_fused_item<JDA_SEG_SPEC>(P, tab, wl_in[item], round, lane, E, wl_out);
            else jda_fused_item<JDA_SEG_RECORD>(P, tab, wl_in[item], round, lane, E, wl_out);
        }
        __threadfence();                                             // this round's entry states, sums and list, for every wavefront of the next
        __syncthreads();
        round++;
        count = __hip_atomic_load(&stats[8u + round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) stats[7] = count == 0 ? 1u : 0u;
}

// batches whose longest scan has at most this many segments (512 KB) take the late rounds without LDS
#define JDA_SMALL_SCAN_SEGS 2048u
extern "C" hipError_t jda_launch_segscan_fused(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t round, hipStream_t stream)
{
    if (n_images == 0 || max_segs == 0) return hipSuccess;
    // Round 0 (exit states only: the lightest walk) ran FASTER with four workgroups per CU than with the eight that 16 KB of tables
    // allowed (455 -> 365 us per 64-image batch).  The tables are 32 KB now (pair halves): five fit; round 0 is launched with 8 KB more
    // than it uses, which makes it four (251 us at five or four, 350 at two; the counting rounds: 331 at five or four, 401 at two --
    // profiles/r03_walk_sq_counters.txt).  JDA_WALK_LDS_R0 / JDA_WALK_LDS_R1: extra bytes, for measuring.
    static const int lds_extra0 = []() { const char *e = JDA_LAB_ENV("JDA_WALK_LDS_R0"); return e ? atoi(e) : 8192; }();
    static const int lds_extra1 = []() { const char *e = JDA_LAB_ENV("JDA_WALK_LDS_R1"); return e ? atoi(e) : 0; }();
    const int lds_max = JDA_WT_BYTES + (lds_extra0 > lds_extra1 ? lds_extra0 : lds_extra1);
    const int lds_bytes = JDA_WT_BYTES + (round == 0 ? lds_extra0 : lds_extra1);
    static std::atomic<unsigned long long> attr_done0(0), attr_done1(0);
    {
        hipError_t e = jda_ensure_lds_limit((const void *)jda_segscan_fused<JDA_SEG_SPEC, true>, lds_max, attr_done0);
        if (e == hipSuccess) e = jda_ensure_lds_limit((const void *)jda_segscan_fused<JDA_SEG_RECORD, true>, lds_max, attr_done1);
        if (e != hipSuccess) return e;
    }
    const uint32_t full = (max_segs + JDA_WALK_THREADS - 1u) / JDA_WALK_THREADS;
    // later rounds walk a few percent of the segments: a few workgroups per image, each stepping through the list
    const dim3 grid(round <= 1 ? full : (full < 8u ? full : 8u), n_images), block(JDA_WALK_THREADS);
    if (round == 0) JDA_LAUNCH((jda_segscan_fused<JDA_SEG_SPEC, true>), grid, block, lds_bytes, stream, params, round);
    else if (round == 1) JDA_LAUNCH((jda_segscan_fused<JDA_SEG_RECORD, true>), grid, block, lds_bytes, stream, params, round);
    else if (max_segs > JDA_SMALL_SCAN_SEGS) JDA_LAUNCH((jda_segscan_fused<JDA_SEG_RECORD, true>), grid, block, lds_bytes, stream, params, round);
    else JDA_LAUNCH((jda_segscan_fused<JDA_SEG_RECORD, false>), grid, block, 0, stream, params, round);
    return hipGetLastError();
}
// The states-first order's rounds (a batch too small to fill the GPU: one image at a time): every round in front of the last walks for
// exit states only (the SPEC walk: half a RECORD walk's instructions, and a lone wavefront's round is as long as its chain of them), the
// last one -- all_record -- records every segment from its settled entry state.
static hipError_t jda_launch_segscan_states_first(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t round, bool all_record, hipStream_t stream)
{
    const int lds_bytes = JDA_WT_BYTES;
    static std::atomic<unsigned long long> attr_s(0), attr_r(0);
    hipError_t e = jda_ensure_lds_limit((const void *)jda_segscan_fused<JDA_SEG_SPEC, true>, JDA_WT_BYTES + 8192, attr_s);
    if (e == hipSuccess) e = jda_ensure_lds_limit((const void *)jda_segscan_fused<JDA_SEG_RECORD, true>, JDA_WT_BYTES + 8192, attr_r);
    if (e != hipSuccess) return e;
    const uint32_t full = (max_segs + JDA_WALK_THREADS - 1u) / JDA_WALK_THREADS;
    const dim3 grid((round <= 1 || all_record) ? full : (full < 8u ? full : 8u), n_images), block(JDA_WALK_THREADS);
    if (all_record) JDA_LAUNCH((jda_segscan_fused<JDA_SEG_RECORD, true>), grid, block, lds_bytes, stream, params, round | JDA_ROUND_ALL);
    else JDA_LAUNCH((jda_segscan_fused<JDA_SEG_SPEC, true>), grid, block, lds_bytes, stream, params, round);
    return hipGetLastError();
}

extern "C" hipError_t jda_launch_segscan_tail(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t first_round, uint32_t max_round, hipStream_t stream)
{
    if (n_images == 0) return hipSuccess;                              // (first_round | JDA_ROUND_ALL: exit states only)
    if (max_segs > JDA_SMALL_SCAN_SEGS) JDA_LAUNCH(jda_segscan_tail<true>, dim3(n_images), dim3(1024), JDA_WT_BYTES, stream, params, first_round, max_round);
    else JDA_LAUNCH(jda_segscan_tail<false>, dim3(n_images), dim3(256), 0, stream, params, first_round, max_round);
    return hipGetLastError();
}

// One element of the scan over an image's segments, and its (associative) combination "A, then B": block starts add up; the
// window's byte lag is a map (field j = exit lag for entry lag j) and maps compose; the DC sums add up, but start over where an
// interval ended inside a segment (flag bit 0: B's sums count from its last restart).
struct jda_sum_el { uint32_t nblk, map, flag; uint32_t d0, d1, d2; };
#define JDA_MAP_IDENTITY 0x2c688u                                   // 3-bit fields: j -> j, j = 0..5
__device__ __forceinline__ uint32_t jda_map_compose(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t j = 0; j < 6u; j++) r |= ((b >> (3u * ((a >> (3u * j)) & 7u))) & 7u) << (3u * j);
    return r;
}
__device__ __forceinline__ jda_sum_el jda_sum_identity() { jda_sum_el e; e.nblk = 0; e.map = JDA_MAP_IDENTITY; e.flag = 0; e.d0 = e.d1 = e.d2 = 0; return e; }
__device__ __forceinline__ jda_sum_el jda_sum_combine(const jda_sum_el &A, const jda_sum_el &B)
{
    jda_sum_el R;
    const bool rst = (B.flag & 1u) != 0;
    R.nblk = A.nblk + B.nblk; R.map = jda_map_compose(A.map, B.map); R.flag = A.flag | B.flag;
    R.d0 = rst ? B.d0 : A.d0 + B.d0; R.d1 = rst ? B.d1 : A.d1 + B.d1; R.d2 = rst ? B.d2 : A.d2 + B.d2;
    return R;
}
__device__ __forceinline__ jda_sum_el jda_sum_shfl_up(const jda_sum_el &v, int d)
{
    jda_sum_el r;
    r.nblk = (uint32_t)__shfl_up((int)v.nblk, d, 64); r.map = (uint32_t)__shfl_up((int)v.map, d, 64); r.flag = (uint32_t)__shfl_up((int)v.flag, d, 64);
    r.d0 = (uint32_t)__shfl_up((int)v.d0, d, 64); r.d1 = (uint32_t)__shfl_up((int)v.d1, d, 64); r.d2 = (uint32_t)__shfl_up((int)v.d2, d, 64);
    return r;
}
__device__ __forceinline__ jda_sum_el jda_sum_wave_scan(jda_sum_el v, uint32_t lane)      // inclusive, all 64 lanes active
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const jda_sum_el o = jda_sum_shfl_up(v, d);
        if (lane >= (uint32_t)d) v = jda_sum_combine(o, v);
    }
    return v;
}
// a chunk = the 64 segments of one wavefront step: every lane's element (what is behind the chunk's first bad segment is dead:
// identity) and the chunk-local verdict
__device__ __forceinline__ jda_sum_el jda_sum_load_chunk(const jda_segscan_params &P, uint32_t chunk, uint32_t lane, uint32_t &first_bad, uint32_t &max_ac)
{
    const uint32_t seg = chunk * 64u + lane;
    jda_sum_el e = jda_sum_identity();
    uint32_t bad = 0;
    max_ac = 0;
    if (seg < P.n_segs) {
        const uint32_t JDA_GLOBAL *su = JDA_G(const uint32_t, P.seg_sum) + (size_t)seg * JDA_SEG_SUM_WORDS;
        e.nblk = su[0]; e.d0 = su[1]; e.d1 = su[2]; e.d2 = su[3]; e.map = su[4]; bad = su[5] & 1u; e.flag = (su[5] & JDA_SEG_HAS_RESTART) ? 1u : 0u;
        max_ac = (su[5] >> 4) & 15u;
    }
    const uint64_t badmask = __builtin_amdgcn_ballot_w64(bad != 0);
    first_bad = badmask ? (uint32_t)__builtin_ctzll(badmask) : 64u;
    if (lane > first_bad) { e = jda_sum_identity(); max_ac = 0; }   // (the bad segment's own block count still counts; nothing behind it does)
    return e;
}

// Exclusive scan over the segments of one image: first block ordinal, DC predictors and the reference window's byte lag at every
// segment's entry.  One workgroup of 16 wavefronts per image: every wavefront scans chunks of 64 segments (chunk, chunk + 16, ..)
// for their aggregates, wavefront 0 scans the aggregates (<= 2,048 of them: 32 MB of scan), every wavefront scans its chunks again
// with the chunk's carry-in and writes seg_start.  (One wavefront walking the segments in order -- 64 serial steps per chunk for
// the lag, then also for the predictors -- took 0.23-0.49 ms per 64-image batch with nothing beside it on the GPU.)
// A segment that met an invalid code ends the sums: harmless only behind the image's last block.
// Result word: stats[6] = 1 when the index can be written (enough blocks, no bad code before the end).
#define JDA_SUMS_WAVES 16u
// (32 MB of scan -- the index packs byte positions in 25 bits --, or what fits the CU's LDS with smaller segments: a longer scan goes
// to the serial pre-scan, stats[6] = 0)
#define JDA_SUMS_MAX_CHUNKS (((1u << 25) / (64u * JDA_SEG_BYTES)) < 6144u ? ((1u << 25) / (64u * JDA_SEG_BYTES)) : 6144u)
__global__ __launch_bounds__(64 * JDA_SUMS_WAVES)
void jda_segscan_sums(const jda_segscan_params *__restrict__ params)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sums_lds[];
    jda_sum_el *agg = (jda_sum_el *)sums_lds;                       // JDA_SUMS_MAX_CHUNKS x: a chunk's aggregate, then its carry-in; flag bit 1: the chunk has a bad segment, bit 2: dead
    const jda_segscan_params P = jda_segscan_resolve(params[blockIdx.x]);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_chunks = (P.n_segs + 63u) / 64u;
    if (n_chunks > JDA_SUMS_MAX_CHUNKS) { if (threadIdx.x == 0) P.stats[6] = 0; return; }    // (the front end admits no scan that long)
    for (uint32_t c = wave; c < n_chunks; c += JDA_SUMS_WAVES) {    // ---- chunk aggregates
        uint32_t first_bad, mac;
        const jda_sum_el incl = jda_sum_wave_scan(jda_sum_load_chunk(P, c, lane, first_bad, mac), lane);
        if (lane == 63u) { agg[c] = incl; agg[c].flag = (incl.flag & 1u) | (first_bad < 64u ? 2u : 0u); }
    }
    __syncthreads();
    if (wave == 0) {                                                // ---- every chunk's carry-in
        jda_sum_el carry = jda_sum_identity();
        bool ended = false;
        for (uint32_t base = 0; base < n_chunks; base += 64u) {
            const uint32_t c = base + lane;
            jda_sum_el e = (c < n_chunks && !ended) ? agg[c] : jda_sum_identity();
            const uint64_t bm = __builtin_amdgcn_ballot_w64((e.flag & 2u) != 0);
            const uint32_t fb = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
            if (lane > fb) e = jda_sum_identity();                  // chunks behind the first bad one are dead
            e.flag &= 1u;
            const jda_sum_el incl = jda_sum_wave_scan(e, lane);
            jda_sum_el excl = jda_sum_shfl_up(incl, 1);
            if (lane == 0) excl = jda_sum_identity();
            jda_sum_el pre = jda_sum_combine(carry, excl);
            pre.flag = (ended || lane > fb) ? 4u : 0u;
            if (c < n_chunks) agg[c] = pre;
            jda_sum_el last;                                        // the whole step, from lane 63
            last.nblk = (uint32_t)__shfl((int)incl.nblk, 63, 64); last.map = (uint32_t)__shfl((int)incl.map, 63, 64); last.flag = (uint32_t)__shfl((int)incl.flag, 63, 64);
            last.d0 = (uint32_t)__shfl((int)incl.d0, 63, 64); last.d1 = (uint32_t)__shfl((int)incl.d1, 63, 64); last.d2 = (uint32_t)__shfl((int)incl.d2, 63, 64);
            carry = jda_sum_combine(carry, last);
            if (bm) ended = true;
        }
        if (lane == 0) P.stats[6] = carry.nblk >= P.n_blocks_total + 1u ? 1u : 0u;     // else: the scan ends before the image does
    }
    __syncthreads();
    uint32_t *seg_start = const_cast<uint32_t *>(P.seg_start);
    for (uint32_t c = wave; c < n_chunks; c += JDA_SUMS_WAVES) {    // ---- every segment's entry values
        uint32_t first_bad, mac;
        const jda_sum_el incl = jda_sum_wave_scan(jda_sum_load_chunk(P, c, lane, first_bad, mac), lane);
        jda_sum_el excl = jda_sum_shfl_up(incl, 1);
        if (lane == 0) excl = jda_sum_identity();
        const jda_sum_el in = agg[c];
        if (P.records) {                                            // RECORD mode: the largest AC category of the settled walks (WRITE's result word)
            if (in.flag & 4u) mac = 0;
            mac = jda_wave_max_u32(mac);
            if (lane == 0 && mac) atomicMax(&P.stats[2], mac);
        }
        const jda_sum_el pre = jda_sum_combine(in, excl);
        const uint32_t seg = c * 64u + lane;
        if (seg < P.n_segs) {
            uint32_t *st = seg_start + (size_t)seg * 5;
            const bool dead = (in.flag & 4u) != 0 || lane > first_bad;       // behind a bad code: the write pass skips these
            st[0] = dead ? 0xfffffff0u : (pre.nblk > 0xfffffff0u ? 0xfffffff0u : pre.nblk);
            st[1] = pre.d0; st[2] = pre.d1; st[3] = pre.d2; st[4] = pre.map & 7u;      // (the lag the scan starts with is 0: field 0 of the composed map)
        }
    }
}

extern "C" hipError_t jda_launch_segscan_sums(const jda_segscan_params *params, uint32_t n_images, hipStream_t stream)
{
    if (n_images == 0) return hipSuccess;
    const int lds_bytes = (int)(JDA_SUMS_MAX_CHUNKS * sizeof(jda_sum_el));
    static std::atomic<unsigned long long> attr_done(0);
    { const hipError_t e = jda_ensure_lds_limit((const void *)jda_segscan_sums, lds_bytes, attr_done); if (e != hipSuccess) return e; }
    JDA_LAUNCH(jda_segscan_sums, dim3(n_images), dim3(64 * JDA_SUMS_WAVES), lds_bytes, stream, params);
    return hipGetLastError();
}

// RECORD mode, after the sums: the records of every segment -> index entries (canonical) and DC predictors, block-parallel -- a
// wavefront per segment, sixteen segments one after the other, lane = record: coalesced reads and writes, no walk.  Result words as
// WRITE left them: [0] bad (a predictor out of range, a stream read on into its padding), [1] closing entry written, [3] max |DC|.
#ifndef JDA_FIN_SEGS_PER_WAVE
#define JDA_FIN_SEGS_PER_WAVE 16u
#endif
__global__ __launch_bounds__(256)
void jda_segscan_finalize(const jda_segscan_params *__restrict__ params)
{
    const jda_segscan_params P = jda_segscan_resolve(params[blockIdx.y]);
    if (!P.records || blockIdx.x * (4u * JDA_FIN_SEGS_PER_WAVE) >= P.n_segs) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;      // (uniform: addresses on the scalar unit)
    const uint32_t seg0 = (blockIdx.x * 4u + wave) * JDA_FIN_SEGS_PER_WAVE;
    const uint32_t inv = jda_fin_recip(P.nblocks);
    jda_fin_acc A;
    A.bad = 0; A.terminal = 0; A.max_abs_dc = 0;
    // the segments' headers first, one lane each (one round trip for all of them), then segment after segment
    uint32_t h_g0 = 0xffffffffu, h_n = 0, h_p0 = 0, h_p1 = 0, h_p2 = 0, h_b0 = 0, h_rf = 0xffffffffu;
    if (lane < JDA_FIN_SEGS_PER_WAVE && seg0 + lane < P.n_segs) {
        const uint32_t JDA_GLOBAL *st = JDA_G(const uint32_t, P.seg_start) + (size_t)(seg0 + lane) * 5;
        h_g0 = st[0]; h_p0 = st[1]; h_p1 = st[2]; h_p2 = st[3];
        h_n = JDA_G(const uint32_t, P.seg_sum)[(size_t)(seg0 + lane) * JDA_SEG_SUM_WORDS];
        h_rf = jda_fin_rst_from(JDA_G(const uint32_t, P.seg_sum)[(size_t)(seg0 + lane) * JDA_SEG_SUM_WORDS + 5u]);
        h_b0 = h_g0 % P.nblocks;                                     // (the one division: the records take their place in the MCU from it)
    }
    // every segment's first 64 records are asked for before the first is used: one trip to memory for the wavefront, not one per segment
    // (a segment behind the image, behind a bad code (g0 = 0xfffffff0) or that does not exist has no records)
    if (h_n > P.rec_cap) { h_n = P.rec_cap; A.bad = 1; }
    if (h_g0 > P.n_blocks_total) h_n = 0;
    uint32_t rec[JDA_FIN_SEGS_PER_WAVE];
#pragma unroll
    for (uint32_t k = 0; k < JDA_FIN_SEGS_PER_WAVE; k++) {
        const uint32_t nblk = (uint32_t)__builtin_amdgcn_readlane((int)h_n, (int)k);
        rec[k] = lane < nblk ? jda_finalize_load(P, seg0 + k, lane) : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < JDA_FIN_SEGS_PER_WAVE; k++) {
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)h_g0, (int)k), nblk = (uint32_t)__builtin_amdgcn_readlane((int)h_n, (int)k);
        const int32_t pr0 = __builtin_amdgcn_readlane((int)h_p0, (int)k), pr1 = __builtin_amdgcn_readlane((int)h_p1, (int)k), pr2 = __builtin_amdgcn_readlane((int)h_p2, (int)k);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)h_b0, (int)k), rf = (uint32_t)__builtin_amdgcn_readlane((int)h_rf, (int)k);
        if (lane < nblk) jda_finalize_apply(P, seg0 + k, lane, rec[k], g0, b0, inv, pr0, pr1, pr2, rf, A);
        for (uint32_t i = lane + 64u; i < nblk; i += 64u) jda_finalize_item(P, seg0 + k, i, g0, b0, inv, pr0, pr1, pr2, rf, A);      // (the densest streams)
    }
    const uint32_t m_dc = jda_wave_max_u32(A.max_abs_dc), n_term = jda_wave_sum_u32(A.terminal);
    const bool any_bad = __builtin_amdgcn_ballot_w64(A.bad != 0) != 0;
    if (lane == 0) {
        if (any_bad) atomicOr(&P.stats[0], 1u);
        if (n_term) atomicAdd(&P.stats[1], n_term);
        if (m_dc) atomicMax(&P.stats[3], m_dc);
    }
}
// .. and the candidates: the truncated reads of the true lag flag their blocks (behind jda_segscan_finalize: it overwrites entries)
__global__ __launch_bounds__(256)
void jda_segscan_resolve_cands(const jda_segscan_params *__restrict__ params)
{
    const jda_segscan_params P = jda_segscan_resolve(params[blockIdx.y]);
    if (!P.records) return;
    if (P.restart_pos) {                                             // every marker where the MCU count puts it?  (result word [5], as WRITE sets it)
        bool mis = false;
        for (uint32_t nr = 1u + blockIdx.x * 256u + threadIdx.x; nr < P.n_intervals; nr += gridDim.x * 256u) mis |= jda_rst_event_item(P, nr) != 0u;
        if (__builtin_amdgcn_ballot_w64(mis) != 0 && (threadIdx.x & 63u) == 0u) atomicOr(&P.stats[5], 1u);
    }
    const uint32_t n = P.stats[JDA_ST_NCAND];
    if (blockIdx.x * 256u >= n) return;
    if (n > P.cand_cap) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&P.stats[0], 1u); return; }     // more than the list holds: the serial pre-scan
    uint32_t hits = 0;
    for (uint32_t ci = blockIdx.x * 256u + threadIdx.x; ci < n; ci += gridDim.x * 256u) hits += jda_resolve_item(P, ci);
    hits = jda_wave_sum_u32(hits);
    if ((threadIdx.x & 63u) == 0u && hits) atomicAdd(&P.stats[4], hits);
}

// every pass of the device pre-scan behind the filter, on one stream: round 0 and the counting round over every segment, the
// work-list rounds, the sums, then WRITE (streams with restart intervals) / finalize + candidates (RECORD mode)
// states_first: the order for a batch that cannot fill the GPU (see jda_launch_segscan_states_first): rounds 0 .. list_rounds - 1 and the
// tail's settle the entry states with the SPEC walk, round max_round (no list round uses that number) records every segment once.
extern "C" hipError_t jda_launch_prescan_passes_ex(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t list_rounds, uint32_t max_round,
                                                   int any_record, int states_first, hipStream_t stream)
{
    if (n_images == 0 || max_segs == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (states_first && any_record) {
        for (uint32_t r = 0; r < list_rounds && e == hipSuccess; r++) e = jda_launch_segscan_states_first(params, n_images, max_segs, r, false, stream);
        if (e == hipSuccess) e = jda_launch_segscan_tail(params, n_images, max_segs, list_rounds | JDA_ROUND_ALL, max_round, stream);
        if (e == hipSuccess) e = jda_launch_segscan_states_first(params, n_images, max_segs, max_round, true, stream);
    } else {
        for (uint32_t r = 0; r < list_rounds && e == hipSuccess; r++) e = jda_launch_segscan_fused(params, n_images, max_segs, r, stream);
        if (e == hipSuccess) e = jda_launch_segscan_tail(params, n_images, max_segs, list_rounds, max_round, stream);
    }
    if (e == hipSuccess) e = jda_launch_segscan_sums(params, n_images, stream);
    if (e == hipSuccess && any_record) {
        JDA_LAUNCH(jda_segscan_finalize, dim3((max_segs + 4u * JDA_FIN_SEGS_PER_WAVE - 1u) / (4u * JDA_FIN_SEGS_PER_WAVE), n_images), dim3(256), 0, stream, params);
        JDA_LAUNCH(jda_segscan_resolve_cands, dim3(16, n_images), dim3(256), 0, stream, params);
        e = hipGetLastError();
    }
    return e;
}
extern "C" hipError_t jda_launch_prescan_passes(const jda_segscan_params *params, uint32_t n_images, uint32_t max_segs, uint32_t list_rounds, uint32_t max_round,
                                                int any_record, hipStream_t stream)
{
    return jda_launch_prescan_passes_ex(params, n_images, max_segs, list_rounds, max_round, any_record, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// The marker / byte-stuffing filter on the device (JPEGFilter, jpeg.inl:1431-1540, over the whole scan: FF 00 -> FF, FF xx
// (xx != 0, also FF FF) -> both bytes dropped, a trailing FF dropped; the offsets at which RSTn markers stood are kept).
// It is a two-state machine over the bytes -- "normal" / "the previous byte was an unpaired FF" -- so it parallelises as a
// scan of state-transition functions: every thread runs its 16 bytes from both possible incoming states (end state,
// bytes emitted, markers seen) and the functions compose.  Three launches over 16 KB chunks of every image of a batch:
//   jda_filter_count   one workgroup per chunk: the chunk's function
//   jda_filter_carry   one wavefront per image: scan of its chunks' functions -> state / output offset / marker count at every
//                      chunk's entry, the image's totals, the sentinel behind the restart positions
//   jda_filter_write   one workgroup per chunk: the scan inside the chunk, output bytes staged in LDS at the alignment they
//                      will have in memory and copied out 16 bytes per thread
// (Round 1's filter was one workgroup per image walking its chunks in order, a byte store per output byte: 1.0-1.6 ms per batch
// of 64 on 64 CUs -- which a decode kernel launched meanwhile could not use: its workgroups need a CU's whole LDS, and the ones
// that had to wait started a millisecond late with a full share of the tiles.)
#define JDA_FILTER_CHUNK 16384u
// a chunk's function as jda_filter_count_v2 leaves it (two words): w0 = end state for incoming 0 | for incoming 1 << 1 | emitted (in 0) << 2 | emitted (in 1) << 17; w1 = markers (in 0) | (in 1) << 16
// work, per image: [chunk] function (2 words) | [chunk] entry state, output offset, marker count (3 words)
__device__ __forceinline__ uint32_t jda_filter_chunks(uint32_t raw_len) { return (raw_len + JDA_FILTER_CHUNK - 1u) / JDA_FILTER_CHUNK; }

struct jda_fsm_wide { uint32_t s, n0, n1, r0, r1; };                // s: end state for incoming 0 | for incoming 1 << 1; counts unpacked (sums over chunks)
__device__ __forceinline__ jda_fsm_wide jda_fsm_wide_compose(const jda_fsm_wide &A, const jda_fsm_wide &B)
{
    const uint32_t m0 = A.s & 1u, m1 = (A.s >> 1) & 1u;
    jda_fsm_wide R;
    R.s = ((B.s >> m0) & 1u) | (((B.s >> m1) & 1u) << 1);
    R.n0 = A.n0 + (m0 ? B.n1 : B.n0); R.n1 = A.n1 + (m1 ? B.n1 : B.n0);
    R.r0 = A.r0 + (m0 ? B.r1 : B.r0); R.r1 = A.r1 + (m1 ? B.r1 : B.r0);
    return R;
}
__device__ __forceinline__ jda_fsm_wide jda_fsm_wide_shfl(const jda_fsm_wide &v, int src, bool up)
{
    jda_fsm_wide r;
    if (up) { r.s = (uint32_t)__shfl_up((int)v.s, src, 64); r.n0 = (uint32_t)__shfl_up((int)v.n0, src, 64); r.n1 = (uint32_t)__shfl_up((int)v.n1, src, 64); r.r0 = (uint32_t)__shfl_up((int)v.r0, src, 64); r.r1 = (uint32_t)__shfl_up((int)v.r1, src, 64); }
    else { r.s = (uint32_t)__shfl((int)v.s, src, 64); r.n0 = (uint32_t)__shfl((int)v.n0, src, 64); r.n1 = (uint32_t)__shfl((int)v.n1, src, 64); r.r0 = (uint32_t)__shfl((int)v.r0, src, 64); r.r1 = (uint32_t)__shfl((int)v.r1, src, 64); }
    return r;
}
__global__ __launch_bounds__(64)
void jda_filter_carry(const jda_filter_params *__restrict__ params)
{
    const jda_filter_params P = params[blockIdx.x];
    const uint32_t lane = threadIdx.x, n_chunks = jda_filter_chunks(P.raw_len);
    uint32_t state = 0, out_base = 0, rst_base = 0;               // the scan starts in state 0 at output offset 0 (uniform)
    uint32_t *carry = P.work + 2u * n_chunks;
    for (uint32_t base = 0; base < n_chunks; base += 64u) {
        const uint32_t c = base + lane;
        jda_fsm_wide v; v.s = 2u; v.n0 = v.n1 = v.r0 = v.r1 = 0;     // identity
        if (c < n_chunks) {
            const uint32_t w0 = P.work[2u * c], w1 = P.work[2u * c + 1u];
            v.s = w0 & 3u; v.n0 = (w0 >> 2) & 0x7fffu; v.n1 = (w0 >> 17) & 0x7fffu; v.r0 = w1 & 0xffffu; v.r1 = w1 >> 16;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const jda_fsm_wide o = jda_fsm_wide_shfl(v, d, true);
            if (lane >= (uint32_t)d) v = jda_fsm_wide_compose(o, v);
        }
        jda_fsm_wide ex = jda_fsm_wide_shfl(v, 1, true);
        if (lane == 0) { ex.s = 2u; ex.n0 = ex.n1 = ex.r0 = ex.r1 = 0; }
        if (c < n_chunks) {                                       // the functions of the chunks before this one, applied to the step's entry
            carry[3u * c] = (ex.s >> state) & 1u;
            carry[3u * c + 1u] = out_base + (state ? ex.n1 : ex.n0);
            carry[3u * c + 2u] = rst_base + (state ? ex.r1 : ex.r0);
        }
        const jda_fsm_wide T = jda_fsm_wide_shfl(v, 63, false);   // the whole step
        out_base += state ? T.n1 : T.n0; rst_base += state ? T.r1 : T.r0; state = (T.s >> state) & 1u;
    }
    if (lane == 0) {
        P.result[0] = out_base; P.result[1] = rst_base;
        if (P.restart_cap) {
            P.restart_pos[0] = 0;
            if (rst_base + 1u < P.restart_cap) P.restart_pos[rst_base + 1u] = JDA_RST_SENTINEL;      // behind the last interval start (the segment walk's search ends there)
        }
    }
}

// ---- the same two kernels without the scan of functions (round 3, late) -------------------------------------------------------
// After any byte that is not FF the machine is in state 0, so a thread's sixteen bytes either FIX the state behind them (they hold a
// byte that is not FF: type K, end state known from a run that starts in state 0) or -- all FF, or none at all behind the end of the
// data -- FLIP it by their parity (type X).  A thread's incoming state is therefore the K-thread nearest in front of it, flipped by the
// X-threads between: two ballots and some mask arithmetic per wavefront (one shuffle when every lane is a K, which is nearly always),
// sixteen 2-bit wavefront summaries per workgroup, and then ONE run of the machine per thread with its true incoming state and plain
// sums of what it emits -- where the scan composed a 2 x 2 table of (end state, bytes, markers) per thread in six shuffle steps and
// ran the machine three times.  (What a chunk emits for the OTHER incoming state, which the carry kernel wants, differs only in the
// chunk's first K-thread: every thread behind it starts from the same state either way, and all-FF threads emit nothing.)
// this thread's 16 bytes of the chunk (valid: how many of them exist)
__device__ __forceinline__ void jda_filter_load(const jda_filter_params &P, uint32_t chunk, uint32_t tid, uint32_t b[4], uint32_t &valid)
{
    const uint32_t off = chunk * JDA_FILTER_CHUNK + tid * 16u;
    valid = off >= P.raw_len ? 0u : (P.raw_len - off < 16u ? P.raw_len - off : 16u);
    b[0] = b[1] = b[2] = b[3] = 0;
    if (valid) {                                                  // (the raw buffer is padded to a multiple of 16 bytes)
        const jda_chunk16_alias v = *(const jda_chunk16_alias JDA_GLOBAL *)(JDA_G(const uint8_t, P.raw) + off);
        b[0] = v.w[0]; b[1] = v.w[1]; b[2] = v.w[2]; b[3] = v.w[3];
    }
}
struct jda_fx { uint32_t is_k, bit; };                              // a thread's / wavefront's / chunk's effect on the state: K: -> bit, X: ^= bit
__device__ __forceinline__ uint32_t jda_fx_apply(uint32_t f, uint32_t st) { return (f & 1u) ? (f >> 1) & 1u : st ^ ((f >> 1) & 1u); }      // f = is_k | bit << 1
// one thread's part: masks, the run from state 0, its type; the wavefront's ballots; returns the wavefront's summary (uniform)
struct jda_fthread { jda_filter_masks M; jda_filter_bits F0; uint32_t valid, is_k, bit; unsigned long long C, B; };
__device__ __forceinline__ uint32_t jda_filter_v2_thread(const jda_filter_params &P, uint32_t chunk, uint32_t tid, uint32_t b[4], jda_fthread &T)
{
    jda_filter_load(P, chunk, tid, b, T.valid);
    T.M = jda_filter_classify(b);
    // bytes in front of the stream's first (raw_skip, in the image's first thread): plain bytes that are not emitted -- the machine
    // starts in state 0 and a byte that is not FF leaves it there
    const uint32_t sk = (chunk == 0u && tid == 0u) ? (1u << P.raw_skip) - 1u : 0u;
    T.M.ff &= ~sk; T.M.zero &= ~sk; T.M.rst &= ~sk;
    T.F0 = jda_filter_run(T.M, T.valid, 0u);
    T.F0.E &= ~sk;
    const uint32_t V = (1u << T.valid) - 1u;
    T.is_k = (T.M.ff & V) != V ? 1u : 0u;                          // (valid == 0: X with parity 0, the identity)
    T.bit = T.is_k ? (T.F0.S >> T.valid) & 1u : T.valid & 1u;
    T.C = __builtin_amdgcn_ballot_w64(T.is_k != 0u);
    T.B = __builtin_amdgcn_ballot_w64(T.bit != 0u);
    const unsigned long long Vm = T.B & T.C, Tm = T.B & ~T.C;
    if (T.C != 0ull) {
        const uint32_t j = 63u - (uint32_t)__builtin_clzll(T.C);
        const unsigned long long above = j == 63u ? 0ull : Tm & ~((2ull << j) - 1ull);
        return 1u | (((uint32_t)((Vm >> j) & 1ull) ^ ((uint32_t)__builtin_popcountll(above) & 1u)) << 1);
    }
    return ((uint32_t)__builtin_popcountll(Tm) & 1u) << 1;
}
// the state a lane starts from, given the state its wavefront starts from
__device__ __forceinline__ uint32_t jda_filter_v2_cin(const jda_fthread &T, uint32_t lane, uint32_t w_in)
{
    if (T.C == ~0ull) {                                            // every lane fixes the state: the lane in front's end state (uniform branch)
        const uint32_t up = (uint32_t)__shfl_up((int)T.bit, 1, 64);
        return lane ? up : w_in;
    }
    const unsigned long long lt = (1ull << lane) - 1ull, below = T.C & lt, Vm = T.B & T.C, Tm = T.B & ~T.C;
    if (below != 0ull) {
        const uint32_t j = 63u - (uint32_t)__builtin_clzll(below);
        const unsigned long long between = Tm & lt & ~((2ull << j) - 1ull);
        return (uint32_t)((Vm >> j) & 1ull) ^ ((uint32_t)__builtin_popcountll(between) & 1u);
    }
    return w_in ^ ((uint32_t)__builtin_popcountll(Tm & lt) & 1u);
}

__global__ __launch_bounds__(1024)
void jda_filter_count_v2(const jda_filter_params *__restrict__ params)
{
    __shared__ uint32_t wf[16], wn[16], alt[2];
    const jda_filter_params P = params[blockIdx.y];
    const uint32_t chunk = blockIdx.x, n_chunks = jda_filter_chunks(P.raw_len), tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (chunk >= n_chunks) return;
    uint32_t b[4];
    jda_fthread T;
    const uint32_t mine = jda_filter_v2_thread(P, chunk, tid, b, T);
    if (lane == 0u) wf[wave] = mine;
    if (tid < 2u) alt[tid] = 0u;
    __syncthreads();
    uint32_t w_in = 0u, prev_all_x = 1u;                             // the chunk entered in state 0; are all wavefronts in front X?
    for (uint32_t w = 0; w < wave; w++) { const uint32_t f = wf[w]; w_in = jda_fx_apply(f, w_in); prev_all_x &= (f & 1u) ^ 1u; }
    const uint32_t cin = jda_filter_v2_cin(T, lane, w_in);
    jda_filter_bits F = T.F0;
    if (cin) F = jda_filter_run(T.M, T.valid, 1u);                    // (rare: the thread in front ended on an unpaired FF)
    uint32_t nr = (uint32_t)__builtin_popcount(F.E) | ((uint32_t)__builtin_popcount(F.R) << 16);
    // the chunk's first K-thread: what it emits from the other state
    if (prev_all_x && T.C != 0ull && lane == (uint32_t)__builtin_ctzll(T.C)) {
        const jda_filter_bits G = cin ? T.F0 : jda_filter_run(T.M, T.valid, 1u);
        alt[0] = (uint32_t)__builtin_popcount(G.E) - (nr & 0xffffu) + 0x8000u;      // (biased: a difference of -16 .. 16)
        alt[1] = (uint32_t)__builtin_popcount(G.R) - (nr >> 16) + 0x8000u;
    }
    nr = jda_wave_sum_u32(nr);
    if (lane == 0u) wn[wave] = nr;
    __syncthreads();
    if (tid == 0u) {
        uint32_t n0 = 0, r0 = 0, f = 2u * 0u;                         // f: the chunk's effect, composed: starts as X with parity 0
        uint32_t s0 = 0u, s1 = 1u;
        for (uint32_t w = 0; w < 16u; w++) { n0 += wn[w] & 0xffffu; r0 += wn[w] >> 16; s0 = jda_fx_apply(wf[w], s0); s1 = jda_fx_apply(wf[w], s1); }
        (void)f;
        const uint32_t n1 = alt[0] ? n0 + alt[0] - 0x8000u : n0, r1 = alt[1] ? r0 + alt[1] - 0x8000u : r0;
        P.work[2u * chunk] = s0 | (s1 << 1) | (n0 << 2) | (n1 << 17);
        P.work[2u * chunk + 1u] = r0 | (r1 << 16);
    }
}

__global__ __launch_bounds__(1024)
void jda_filter_write_v2(const jda_filter_params *__restrict__ params)
{
    __shared__ uint32_t wf[16], wn[16];
    __shared__ __attribute__((aligned(16))) uint8_t stage[JDA_FILTER_CHUNK + 32];
    const jda_filter_params P = params[blockIdx.y];
    const uint32_t chunk = blockIdx.x, n_chunks = jda_filter_chunks(P.raw_len), tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (chunk >= n_chunks) return;
    const uint32_t *carry = P.work + 2u * n_chunks + 3u * chunk;
    const uint32_t state = carry[0], out_base = carry[1], rst_base = carry[2];
    uint32_t b[4];
    jda_fthread T;
    const uint32_t mine = jda_filter_v2_thread(P, chunk, tid, b, T);
    if (lane == 0u) wf[wave] = mine;
    __syncthreads();
    uint32_t w_in = state;
    for (uint32_t w = 0; w < wave; w++) w_in = jda_fx_apply(wf[w], w_in);
    const uint32_t st = jda_filter_v2_cin(T, lane, w_in);
    jda_filter_bits F = T.F0;
    if (st) F = jda_filter_run(T.M, T.valid, 1u);
    // bytes and markers in front of this thread: an inclusive sum over the wavefront, the wavefronts in front from LDS
    const uint32_t nr = (uint32_t)__builtin_popcount(F.E) | ((uint32_t)__builtin_popcount(F.R) << 16);
    uint32_t inc = nr;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63u) wn[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, total = 0;
    for (uint32_t w = 0; w < 16u; w++) { const uint32_t t = wn[w]; if (w < wave) pre += t; total += t; }
    const uint32_t before = pre + inc - nr;
    const uint32_t mis = out_base & 15u;                          // the staged bytes sit at the alignment they will have in memory
    uint32_t o = mis + (before & 0xffffu);
    uint32_t rp = rst_base + (before >> 16);
    uint32_t JDA_GLOBAL *rpos = JDA_G(uint32_t, P.restart_pos);
    for (uint32_t R = F.R; R != 0u; R &= R - 1u) {                 // RSTn: the next interval starts here (rare)
        const uint32_t k = (uint32_t)__builtin_ctz(R);
        rp++;
        if (rp < P.restart_cap) rpos[rp] = out_base + (o - mis) + (uint32_t)__builtin_popcount(F.E & ((1u << k) - 1u));
    }
    // every byte is stored -- to its place, or to a dump byte behind the buffer -- so that the sixteen steps are straight-line code.
    // FF 00 -> FF: the 00 that leaves in state 1 becomes the FF
    const uint32_t sz = F.S & T.M.zero;
#pragma unroll
    for (uint32_t d = 0; d < 4; d++) {                               // nibble -> 0xff in the bytes of its set bits
        const uint32_t one = jda_umul24((sz >> (4u * d)) & 15u, 0x00204081u) & 0x01010101u;
        b[d] |= (one << 8) - one;
    }
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const bool emit = ((F.E >> k) & 1u) != 0u;
        stage[emit ? o : (uint32_t)(JDA_FILTER_CHUNK + 31u)] = (uint8_t)(b[k >> 2] >> (8 * (k & 3)));
        o += emit ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t n_out = total & 0xffffu, end = mis + n_out;
    uint8_t JDA_GLOBAL *gout = JDA_G(uint8_t, P.out) + (out_base - mis);                   // 16-byte aligned (P.out is)
    for (uint32_t piece = tid; piece * 16u < end; piece += 1024u) {
        const uint32_t lo = piece * 16u, hi = lo + 16u;
        if (lo >= mis && hi <= end) *(jda_chunk16_alias JDA_GLOBAL *)(gout + lo) = *(const jda_chunk16_alias *)(stage + lo);
        else for (uint32_t i = lo < mis ? mis : lo; i < (hi < end ? hi : end); i++) gout[i] = stage[i];        // (the chunk's first and last 16 bytes: a neighbour writes the rest)
    }
}

// max_raw_len: the longest raw_len among the images (the grid's width)
extern "C" hipError_t jda_launch_filter(const jda_filter_params *params, uint32_t n_images, uint32_t max_raw_len, hipStream_t stream)
{
    if (n_images == 0) return hipSuccess;
    const uint32_t chunks = (max_raw_len + JDA_FILTER_CHUNK - 1u) / JDA_FILTER_CHUNK;
    if (chunks) JDA_LAUNCH(jda_filter_count_v2, dim3(chunks, n_images), dim3(1024), 0, stream, params);
    JDA_LAUNCH(jda_filter_carry, dim3(n_images), dim3(64), 0, stream, params);
    if (chunks) JDA_LAUNCH(jda_filter_write_v2, dim3(chunks, n_images), dim3(1024), 0, stream, params);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// A position-dependent 64-bit checksum of a decoded surface (rows x row_bytes at pitch), made where the pixels are: the
// multi-GPU driver all-reduces one of these per image to prove that every image was decoded exactly once and identically on
// whichever GPU took it (bench.py, jpegdec_amd/sharding.py); tests use it for surfaces too large to copy back.
//   sum over the surface's dwords d at linear index i (row * dwords_per_row + column; a row's tail bytes zero-extended) of
//   (uint64)((d ^ (i * 0x9E3779B1)) * 0x85EBCA6B mod 2^32) * (2 i + 1)        mod 2^64
__global__ __launch_bounds__(256)
void jda_surface_checksum(const uint8_t *__restrict__ base, uint32_t pitch, uint32_t row_bytes, uint32_t rows, unsigned long long *out)
{
    const uint32_t dpr = (row_bytes + 3u) / 4u;
    const uint64_t total = (uint64_t)dpr * rows;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t r = (uint32_t)(i / dpr), c = (uint32_t)(i - (uint64_t)r * dpr);
        const uint8_t JDA_GLOBAL *p = JDA_G(const uint8_t, base) + (size_t)r * pitch + (size_t)c * 4u;
        uint32_t d;
        if (c * 4u + 4u <= row_bytes) d = *(const jda_u32_alias JDA_GLOBAL *)p;      // (pitch and base are 16-byte aligned)
        else { d = 0; for (uint32_t k = 0; c * 4u + k < row_bytes; k++) d |= (uint32_t)p[k] << (8u * k); }
        const uint32_t m = (d ^ ((uint32_t)i * 0x9E3779B1u)) * 0x85EBCA6Bu;
        acc += (unsigned long long)m * (2ull * i + 1ull);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63u) == 0) atomicAdd(out, acc);
}
extern "C" hipError_t jda_launch_checksum(const void *base, uint32_t pitch, uint32_t row_bytes, uint32_t rows, unsigned long long *out, hipStream_t stream)
{
    if (rows == 0 || row_bytes == 0) return hipSuccess;
    const uint64_t total = (uint64_t)((row_bytes + 3u) / 4u) * rows;
    uint64_t g = (total + 256u * 8u - 1u) / (256u * 8u);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    JDA_LAUNCH(jda_surface_checksum, dim3((uint32_t)g), dim3(256), 0, stream, (const uint8_t *)base, pitch, row_bytes, rows, out);
    return hipGetLastError();
}

extern "C" hipError_t jda_internal_set_wgtrace(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_jda_wgtrace), &dev_buf, sizeof(dev_buf));
}

extern "C" hipError_t jda_internal_set_trace(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_jda_trace), &dev_buf, sizeof(dev_buf));
}

// Launch entry used by jda_runtime.cpp.  n_tiles is a multiple of jda_lds_layout<MODE>::WAVES (padded per image).
extern "C" hipError_t jda_launch_decode(int mode, int fast_mul, int variant, int big, int cont, const jda_dev_desc *descs, const jda_strip *tiles,
                                        uint32_t n_tiles, uint32_t aux, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    if (cont && !fast_mul && variant == 3) {          // JDA_LIST_THUMB_FLAT: whole gray / 4:2:0 images at 1/8, a record per image
        if ((mode != JDA_MODE_GRAY && mode != JDA_MODE_420) || big) return hipErrorInvalidValue;
        if (!aux) return hipSuccess;
        return mode == JDA_MODE_GRAY ? launch_dc_thumbnail_flat(descs, tiles, n_tiles, aux, stream) : launch_dc_thumbnail_flat420(descs, tiles, n_tiles, aux, stream);
    }
    if (cont) {                                       // P1 in chunks (jda_use_cont decides who gets here): the RGB8888 plain case of 4:2:0 and 4:4:4
        if (!fast_mul || variant != 1) return hipErrorInvalidValue;
        switch (mode) {
        case JDA_MODE_444: return launch_persistent<JDA_MODE_444, 1, 1, 1>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_420: return launch_persistent<JDA_MODE_420, 1, 1, 1>(big, descs, tiles, n_tiles, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (!fast_mul && variant == 3 && big) {           // JDA_LIST_QUARTER: 1/4 scale, a kernel of its own (the list is padded as the large-window kernels')
        switch (mode) {
        case JDA_MODE_GRAY: return launch_quarter<JDA_MODE_GRAY>(descs, tiles, n_tiles, jda_lds_layout<JDA_MODE_GRAY, 1>::WAVES, stream);
        case JDA_MODE_444: return launch_quarter<JDA_MODE_444>(descs, tiles, n_tiles, jda_lds_layout<JDA_MODE_444, 1>::WAVES, stream);
        case JDA_MODE_420: return launch_quarter<JDA_MODE_420>(descs, tiles, n_tiles, jda_lds_layout<JDA_MODE_420, 1>::WAVES, stream);
        case JDA_MODE_422: return launch_quarter<JDA_MODE_422>(descs, tiles, n_tiles, jda_lds_layout<JDA_MODE_422, 1>::WAVES, stream);
        case JDA_MODE_440: return launch_quarter<JDA_MODE_440>(descs, tiles, n_tiles, jda_lds_layout<JDA_MODE_440, 1>::WAVES, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (!fast_mul && variant == 3) {                  // JDA_LIST_THUMB: 1/8 scale, the DC values are the pixels
        switch (mode) {
        case JDA_MODE_GRAY: return launch_dc_thumbnail<JDA_MODE_GRAY>(descs, tiles, n_tiles, stream);
        case JDA_MODE_444: return launch_dc_thumbnail<JDA_MODE_444>(descs, tiles, n_tiles, stream);
        case JDA_MODE_420: return launch_dc_thumbnail<JDA_MODE_420>(descs, tiles, n_tiles, stream);
        case JDA_MODE_422: return launch_dc_thumbnail<JDA_MODE_422>(descs, tiles, n_tiles, stream);
        case JDA_MODE_440: return launch_dc_thumbnail<JDA_MODE_440>(descs, tiles, n_tiles, stream);
        default: return hipErrorInvalidValue;
        }
    }
    // big: the larger scan window for high-bitrate images, one wavefront less per workgroup (jda_big_window in jda_runtime.cpp decides)
    if (variant >= 1 && fast_mul) {                   // the plain-case kernels (jda_plain_variant decides who gets here)
        switch (mode * 4 + variant) {
        case JDA_MODE_444 * 4 + 1: return launch_persistent<JDA_MODE_444, 1, 1>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_420 * 4 + 1: return launch_persistent<JDA_MODE_420, 1, 1>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_422 * 4 + 1: return launch_persistent<JDA_MODE_422, 1, 1>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_444 * 4 + 2: return launch_persistent<JDA_MODE_444, 1, 2>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_420 * 4 + 2: return launch_persistent<JDA_MODE_420, 1, 2>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_420 * 4 + 3: return launch_persistent<JDA_MODE_420, 1, 3>(big, descs, tiles, n_tiles, stream);
        case JDA_MODE_GRAY * 4 + 3: return launch_persistent<JDA_MODE_GRAY, 1, 3>(big, descs, tiles, n_tiles, stream);
        default: return hipErrorInvalidValue;
        }
    }
    if (variant != 0) return hipErrorInvalidValue;
    switch (mode) {                                   // the general kernels: 24- or 32-bit multiplies as the image's descriptor says
    case JDA_MODE_GRAY: return launch_persistent<JDA_MODE_GRAY, -1, 0>(big, descs, tiles, n_tiles, stream);
    case JDA_MODE_444: return launch_persistent<JDA_MODE_444, -1, 0>(big, descs, tiles, n_tiles, stream);
    case JDA_MODE_420: return launch_persistent<JDA_MODE_420, -1, 0>(big, descs, tiles, n_tiles, stream);
    case JDA_MODE_422: return launch_persistent<JDA_MODE_422, -1, 0>(big, descs, tiles, n_tiles, stream);
    case JDA_MODE_440: return launch_persistent<JDA_MODE_440, -1, 0>(big, descs, tiles, n_tiles, stream);
    default: return hipErrorInvalidValue;
    }
}
