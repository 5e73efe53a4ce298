// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE: a lane-by-lane CPU emulation of one wavefront.
//
// Compiles the very same per-lane decode logic the HIP kernels use (jpegdec_amd/csrc/
// jda_device_core.h) with g++ and runs every strip the way the kernel does: all 64 lanes through
// phase A, then all 64 lanes through phase B, over a byte array standing in for the wave's LDS.
// It exists so the kernel logic can be checked against the oracle on machines without a GPU
// (`pytest -m "not gpu"`).  It is not part of libjpegdec_amd.so and nothing in the product calls it.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <vector>
static unsigned g_seg_steps = 0;
static unsigned long long g_all_steps = 0;
static bool g_pair_off = false;
#define JDA_SEG_STEP_HOOK() (g_seg_steps++, g_all_steps++)
#define JDA_SEG_PAIR_OFF() g_pair_off
static int g_trace_seg = -1;
#define JDA_SEG_TRACE_HOOK(OP, seg, p, k, kk, b2, bn, ends, iend, next_bit, nr, nblk, e, inval) do { if ((int)(seg) == g_trace_seg) fprintf(stderr, "  op %d seg %u: p %u (abs %u) k %u kk %u b2 %u bn %u ends %d iend %d next_bit %d nr %u nblk %u e %04x inval %d\n", (int)(OP), (unsigned)(seg), (unsigned)(p), (unsigned)((seg) * 2048u + (p)), (unsigned)(k), (unsigned)(kk), (unsigned)(b2), (unsigned)(bn), (int)(ends), (int)(iend), (int)(next_bit), (unsigned)(nr), (unsigned)(nblk), (unsigned)(e), (int)(inval)); } while (0)
#include "../../jpegdec_amd/csrc/jda_device_core.h"
#include "../../jpegdec_amd/csrc/jda_plan.h"

static unsigned long long g_chunk_items = 0;      // continuation entries decoded as chunks since the last call of hostsim_chunk_items
extern "C" unsigned long long hostsim_chunk_items(void) { const unsigned long long n = g_chunk_items; g_chunk_items = 0; return n; }
static int g_use_cont = 0;        // 1: decode through P1's chunked mode (the serial pre-scan's continuation entries)
extern "C" void hostsim_set_chunked(int on) { g_use_cont = on; }      // (the serial pre-scan then writes entries for every image: JDA_PREPARE_CONT_ALWAYS)
extern "C" const uint32_t *jda_image_block_cont(const jda_image *img, const uint32_t **cont_first, uint32_t *n_cont);
static int g_reverse_tiles = 0;   // tests run the tiles in reverse order too: results must not depend on which wave finishes first
extern "C" void hostsim_set_reverse(int on) { g_reverse_tiles = on; }
static uint32_t g_window_bytes = 1024;   // tests shrink it to exercise the HBM fall-back of the bit reader (each layout caps it at its WIN_BYTES)
extern "C" void hostsim_set_window(uint32_t bytes) { g_window_bytes = bytes > 1024 ? 1024 : (bytes & ~15u); }

// one wavefront = 64 lanes stepping through the kernel's phases; a phase runs for every lane before
// the next one starts (= the wave-local fence between them)
template <int MODE, bool FAST>
static void run_tiles(const jda_dev_desc &D, const std::vector<jda_strip> &tiles)
{
    typedef jda_lds_layout<MODE> L;
    std::vector<uint64_t> tab_store((JDA_LT_BYTES + 7) / 8), lds_store((L::WAVE_BYTES + 7) / 8);
    uint8_t *tab = (uint8_t *)tab_store.data(), *wl = (uint8_t *)lds_store.data();
    for (uint32_t tid = 0; tid < 256; tid++) jda_p0_tables(D, tid, 256, tab, L::LONG_LDS != 0);
    for (size_t ii = 0; ii < tiles.size(); ii++) {
        const size_t i = g_reverse_tiles ? tiles.size() - 1 - ii : ii;
        const jda_strip &S = tiles[i];
        memset(wl, 0xA5, L::WAVE_BYTES);         // poison: LDS is not zero-initialised on the GPU either
        const jda_tile_ctx C = jda_tile_setup<MODE>(D, S);
        jda_p1_inputs in[JDA_TILE_THREADS];
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) in[t] = jda_p1_prefetch<MODE>(D, C, t);
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p0_stage<MODE>(D, C, t, wl, g_window_bytes);
        uint32_t flags[JDA_TILE_THREADS];
        jda_lane_pre LP[JDA_TILE_THREADS];
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_lane_prepare<MODE>(LP[t], D, t, tab);
        if (D.blk_cont_first && D.scale_shift < 2) {       // P1 in chunks (jda_p1c_*): pass A for every lane, the tile's continuation entries, the finish
            typedef jda_mode_traits<MODE> T;
            const uint32_t win_len = C.win_len < g_window_bytes ? C.win_len : g_window_bytes;
            const bool chunked = C.win_need <= win_len && !(D.pad_[0] & JDA_DESC_GENERAL_P1);
            jda_p1c_own own[JDA_TILE_THREADS];
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) own[t] = jda_p1c_block<MODE>(D, C, in[t], LP[t], tab, wl, wl + L::WIN_OFF, g_window_bytes, chunked);
            if (chunked && C.count) {
                const uint32_t nb = C.count * (uint32_t)T::NBLK, c0 = D.blk_cont_first[C.first_block], c1 = D.blk_cont_first[C.first_block + nb];
                for (uint32_t e = c0; e < c1; e++) {
                    const uint32_t entry = D.blk_cont[e], bl = (JDA_CONT_G7(entry) - C.first_block) & 127u;
                    if (bl >= nb) { fprintf(stderr, "hostsim: continuation entry %u of tile at block %u names lane %u\n", e, C.first_block, bl); abort(); }
                    g_chunk_items++;
                    jda_p1c_item<MODE>(D, C, entry, bl, own[bl].bits, true, tab, wl, wl + L::WIN_OFF);
                }
            }
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) flags[t] = jda_p1c_finish<MODE>(own[t], t, wl);
        } else
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) flags[t] = jda_p1_entropy<MODE>(D, C, in[t], LP[t], tab, wl, wl + L::WIN_OFF, g_window_bytes);
        if (D.scale_shift < 2) {
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p1_lists<MODE>(D, LP[t], t, flags[t], flags, tab, wl);
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p2_columns<MODE, FAST>(D, t, tab, wl);
            for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_p3_rows<MODE>(D, t, tab, wl);
        }
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) {
            jda_p4_pre P4;
            jda_p4_prepare<MODE>(P4, D, t);
            jda_p4_output<MODE>(D, S, C, t, wl, P4);
        }
    }
}

// the 1/4-scale kernel (jda_quarter_tiles): lane = block for the decode, lane = output pixel for the store
template <int MODE>
static void run_quarter_tiles(const jda_dev_desc &D, const std::vector<jda_strip> &tiles)
{
    typedef jda_lds_layout<MODE> L;
    typedef jda_mode_traits<MODE> T;
    std::vector<uint64_t> tab_store((JDA_LT_BYTES + 7) / 8);
    uint8_t *tab = (uint8_t *)tab_store.data();
    for (uint32_t tid = 0; tid < 256; tid++) jda_p0_tables(D, tid, 256, tab, L::LONG_LDS != 0);
    const bool dc_only = (D.pad_[0] & JDA_DESC_DC_ONLY) != 0;
    for (size_t ii = 0; ii < tiles.size(); ii++) {
        const jda_strip &S = tiles[g_reverse_tiles ? tiles.size() - 1 - ii : ii];
        const uint32_t first_mcu = S.mcu_y * D.mcus_x + S.mcu_x0;
        uint32_t count = S.count;
        if (first_mcu >= D.n_mcus_ok) count = 0;
        else if (first_mcu + count > D.n_mcus_ok) count = D.n_mcus_ok - first_mcu;
        if (count == 0) continue;
        uint32_t px[JDA_TILE_THREADS];
        bool any_trunc = false;
        for (uint32_t t = 0; t < count * (uint32_t)T::NBLK; t++) any_trunc = any_trunc || (D.blk_index[first_mcu * T::NBLK + t] & JDA_INDEX_TRUNC);
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) {
            px[t] = 0;
            jda_lane_pre LP;
            jda_lane_prepare<MODE>(LP, D, t, tab);
            const bool skip = MODE != JDA_MODE_GRAY && D.gray_from_color && LP.chroma;
            if (t >= count * (uint32_t)T::NBLK || skip) continue;
            const int16_t *quant = (const int16_t *)(tab + LP.quant_off);
            jda_q4_quant Q;
            Q.q0 = quant[0]; Q.q1 = quant[1]; Q.q8 = quant[8]; Q.q9 = quant[9];
            const uint32_t ix = D.blk_index[first_mcu * T::NBLK + t];
            const int32_t dc = D.blk_dc[first_mcu * T::NBLK + t];
            jda_q4_bits B;
            memset(&B, 0, sizeof(B));
            if (!dc_only) B = jda_q4_load(D.scan, D.scan_len, ix);
            const bool trunc = (ix & JDA_INDEX_TRUNC) != 0u;
            px[t] = any_trunc ? jda_q4_block<true>(ix, dc, B, (const uint16_t *)(tab + LP.ac_off), Q, dc_only, trunc)
                              : jda_q4_block<false>(ix, dc, B, (const uint16_t *)(tab + LP.ac_off), Q, dc_only, false);
        }
        for (uint32_t t = 0; t < JDA_TILE_THREADS; t++) jda_q4_store<MODE>(D, S, count, t, px[t], px);
    }
}

extern "C" const uint32_t *jda_image_restart_positions(const jda_image *img, uint32_t *n);
extern "C" void jda_image_component_ids(const jda_image *img, uint8_t *dc_id, uint8_t *ac_id, uint8_t *q_id);
extern "C" void jda_image_adopt_prescan(jda_image *img, uint32_t n_mcus_ok, uint32_t max_ac_bits, int32_t max_abs_dc, uint32_t trunc_events);
static int g_device_prescan = 0;     // != 0: make the block index with the device's segment walk (jda_seg_walk), as jda_upload_batch / jda_pipeline do
static int g_prescan_used = 0;
static uint32_t g_round2_list = 0;      // segments the round behind the first recording round had to walk again
extern "C" uint32_t hostsim_round2_list(void) { return g_round2_list; }
static uint32_t g_prescan_cands = 0;     // truncation candidates the last RECORD-mode pre-scan appended
extern "C" void hostsim_trace_segment(int seg) { g_trace_seg = seg; }
// the marker filter's sixteen-byte state machine (jda_filter_classify / jda_filter_run) against the byte-by-byte machine:
// returns 0 when S, E, R agree for this group, valid count and incoming state
extern "C" int hostsim_filter_bits_check(const uint8_t *bytes16, uint32_t valid, uint32_t cin)
{
    uint32_t b[4];
    memcpy(b, bytes16, 16);
    const jda_filter_bits F = jda_filter_run(jda_filter_classify(b), valid, cin);
    uint32_t st = cin, S = 0, E = 0, R = 0;
    for (uint32_t k = 0; k < valid; k++) {
        const uint32_t c = bytes16[k];
        S |= st << k;
        if (st ? c == 0u : c != 0xffu) E |= 1u << k;
        if (st && (c & 0xf8u) == 0xd0u) R |= 1u << k;
        st = st ? 0u : (c == 0xffu ? 1u : 0u);
    }
    S |= st << valid;
    const uint32_t keep = (2u << valid) - 1u;
    return ((F.S & keep) == S && F.E == E && F.R == R) ? 0 : 1;
}
extern "C" uint32_t hostsim_prescan_candidates(void) { return g_prescan_cands; }
extern "C" uint32_t jda_image_record_cap(const jda_image *img);
static uint32_t g_prescan_trunc = 0;
extern "C" void hostsim_set_device_prescan(int on) { g_device_prescan = on; }
extern "C" int hostsim_prescan_used(void) { return g_prescan_used; }
extern "C" void jda_image_run_host_prescan(jda_image *img);
static int g_segscan_rounds = 0;     // speculative rounds of the last marker-less device pre-scan
static int g_states_first = 0;       // the order jda_upload_batch takes for a batch too small to fill the GPU (jda_launch_prescan_passes_ex)
extern "C" void hostsim_set_states_first(int on) { g_states_first = on; }
extern "C" int hostsim_segscan_rounds(void) { return g_segscan_rounds; }
static int g_index_equal = -1;       // after a device-pre-scanned decode: 1 if its index == the serial host pre-scan's
extern "C" int hostsim_index_equal(void) { return g_index_equal; }

extern "C" int hostsim_decode(const uint8_t *jpeg, int len, int pixel_type, int options,
                              uint8_t *out, int pitch_bytes, int width_px, int rows)
{
    int32_t err = 0;
    jda_image *img = jda_prepare_ex(jpeg, len, (g_device_prescan ? JDA_PREPARE_DEVICE_PRESCAN : 0) | (g_use_cont ? JDA_PREPARE_CONT_ALWAYS : 0), &err);
    if (!img) return err ? err : -1;
    std::vector<uint32_t> dev_index;
    std::vector<int16_t> dev_dc;
    g_prescan_used = 0;
    if (jda_image_prescan_pending(img)) {
        // the segment walk (8f N2): what jda_upload_batch / jda_pipeline + jda_segscan do, lane by lane -- streams without restart
        // markers, and (hostsim_set_device_prescan(2), as jda_pipeline does it) streams with them
        const jda_image_info *I = jda_image_get_info(img);
        const size_t nb = (size_t)I->mcus_x * I->mcus_y * I->blocks_per_mcu;
        dev_index.assign(nb + 1, 0xdeadbeefu); dev_dc.assign(nb, 0x7777);      // (finalize stores whole entries: nothing relies on what was there)
        uint32_t sl = 0, tb = 0;
        const uint8_t *scan = jda_image_scan(img, &sl);
        const uint8_t *tables = jda_image_tables(img, &tb);
        const uint32_t n_segs = sl / JDA_SEG_BYTES + 1u;
        std::vector<uint32_t> padded(((size_t)n_segs * JDA_SEG_BYTES + 16) / 4 + 1, 0);
        memcpy(padded.data(), scan, sl);
        std::vector<uint64_t> lt_store((JDA_WT_BYTES + 7) / 8);
        uint8_t *lt = (uint8_t *)lt_store.data();

        std::vector<uint32_t> ea(n_segs + 1, 0), eb(n_segs + 1, 0), seg_sum((size_t)n_segs * JDA_SEG_SUM_WORDS), seg_start((size_t)n_segs * 5, 0);
        jda_segscan_params P;
        memset(&P, 0, sizeof(P));
        P.scan = (const uint8_t *)padded.data(); P.tables = tables;
        P.seg_sum = seg_sum.data(); P.seg_start = seg_start.data();
        P.blk_index = dev_index.data(); P.blk_dc = dev_dc.data();
        P.scan_len = sl; P.n_segs = n_segs; P.n_blocks_total = (uint32_t)nb;
        P.nblocks = (uint8_t)I->blocks_per_mcu; P.nluma = (uint8_t)(I->blocks_per_mcu - (I->ncomp == 3 ? 2 : 0));
        uint8_t q_id[3];
        jda_image_component_ids(img, P.dc_id, P.ac_id, q_id);
        std::vector<uint32_t> rp;                               // restart intervals: where they start in the filtered scan + the sentinel
        if (I->restart_interval) {
            uint32_t n_int = 0;
            const uint32_t *r0 = jda_image_restart_positions(img, &n_int);
            rp.assign(r0, r0 + n_int); rp.push_back(JDA_RST_SENTINEL);
            P.restart_pos = rp.data(); P.n_intervals = n_int; P.interval_blocks = (uint32_t)I->restart_interval * P.nblocks;
            P.round_last = ((uint32_t)(I->mcus_x * I->mcus_y) % (uint32_t)I->restart_interval) == 0 ? 1u : 0u;
        }
        for (uint32_t tid = 0; tid < 256; tid++) jda_walk_tables_from(tables, jda_wt_dc_follow(P), tid, 256, lt);
        const bool rst = P.restart_pos != nullptr;
        // the passes as jda_pipeline / jda_upload_batch run them
        std::vector<uint32_t> records, cands, stats(80, 0), rst_events;
        P.rec_cap = jda_image_record_cap(img); P.cand_cap = std::max<uint32_t>(1024u, n_segs * 16u);
        records.assign((size_t)n_segs * P.rec_cap, 0xdeadbeefu); cands.assign((size_t)P.cand_cap * 4, 0);
        P.records = records.data(); P.cands = cands.data(); P.stats = stats.data();
        if (rst) { rst_events.assign((size_t)P.n_intervals * 2 + 2, 0); P.rst_events = rst_events.data(); }
        jda_seg_sum S;
        jda_seg_stats ST;
        memset(&ST, 0, sizeof(ST));
        uint32_t rounds = 0;
        bool settled = false;
        uint32_t *cur = ea.data();
        {
            // round 0: every segment from the guess; round 1: every segment from what round 0 handed it, recording; from then on the
            // segments whose entry state changed (their records, sums and stamps are overwritten; candidates of earlier walks go stale)
            std::vector<uint32_t> E(n_segs + 1, 0), list, next;
            for (uint32_t seg = 0; seg < n_segs; seg++) {
                const uint32_t *slot = padded.data() + (size_t)seg * (JDA_SEG_BYTES / 4);
                E[seg + 1] = rst ? jda_seg_walk<JDA_SEG_SPEC, true>(P, seg, 0u, slot, lt, S, ST) : jda_seg_walk<JDA_SEG_SPEC, false>(P, seg, 0u, slot, lt, S, ST);
            }
            for (uint32_t seg = 0; seg < n_segs; seg++) list.push_back(seg);
            rounds = 1;
            while (!list.empty() && rounds < 57) {
                const std::vector<uint32_t> snap(E);            // (a round's lanes read what the round before left)
                next.clear();
                for (uint32_t seg : list) {
                    const uint32_t *slot = padded.data() + (size_t)seg * (JDA_SEG_BYTES / 4);
                    const uint32_t entry = seg == 0 ? 0u : snap[seg];
                    uint32_t x;
                    if (g_states_first) x = rst ? jda_seg_walk<JDA_SEG_SPEC, true>(P, seg, entry, slot, lt, S, ST, rounds) : jda_seg_walk<JDA_SEG_SPEC, false>(P, seg, entry, slot, lt, S, ST, rounds);
                    else {
                        x = rst ? jda_seg_walk<JDA_SEG_RECORD, true>(P, seg, entry, slot, lt, S, ST, rounds) : jda_seg_walk<JDA_SEG_RECORD, false>(P, seg, entry, slot, lt, S, ST, rounds);
                        uint32_t *o = &seg_sum[(size_t)seg * JDA_SEG_SUM_WORDS];
                        o[0] = S.nblk; o[1] = (uint32_t)S.dcsum[0]; o[2] = (uint32_t)S.dcsum[1]; o[3] = (uint32_t)S.dcsum[2]; o[4] = S.phase_map;
                        o[5] = S.bad | (S.max_ac << 4); o[6] = S.lag_last; o[7] = rounds;
                    }
                    if (seg + 1 < n_segs && x != E[seg + 1]) { E[seg + 1] = x; next.push_back(seg + 1); }
                }
                if (rounds == 1) g_round2_list = (uint32_t)next.size();
                list.swap(next);
                rounds++;
            }
            settled = list.empty();
            if (g_states_first) {                               // .. and ONE recording round over every segment, from its settled entry state (round number 56)
                for (uint32_t seg = 0; seg < n_segs; seg++) {
                    const uint32_t *slot = padded.data() + (size_t)seg * (JDA_SEG_BYTES / 4);
                    const uint32_t entry = seg == 0 ? 0u : E[seg];
                    const uint32_t x = rst ? jda_seg_walk<JDA_SEG_RECORD, true>(P, seg, entry, slot, lt, S, ST, 56u) : jda_seg_walk<JDA_SEG_RECORD, false>(P, seg, entry, slot, lt, S, ST, 56u);
                    uint32_t *o = &seg_sum[(size_t)seg * JDA_SEG_SUM_WORDS];
                    o[0] = S.nblk; o[1] = (uint32_t)S.dcsum[0]; o[2] = (uint32_t)S.dcsum[1]; o[3] = (uint32_t)S.dcsum[2]; o[4] = S.phase_map;
                    o[5] = S.bad | (S.max_ac << 4); o[6] = S.lag_last; o[7] = 56u;
                    if (seg + 1 < n_segs && x != E[seg + 1]) settled = false;      // (a recording walk leaves the state a SPEC walk leaves)
                }
            }
            for (uint32_t i = 0; i <= n_segs; i++) ea[i] = E[i];
        }
        g_segscan_rounds = (int)rounds;
        if (getenv("HOSTSIM_SEGDEBUG")) {       // how fast do the states become the true ones?
            std::vector<uint32_t> truth(cur, cur + n_segs + 1), a(n_segs + 1, 0), b2(n_segs + 1, 0);
            for (auto &t : truth) t &= ~JDA_SEG_CHANGED;
            uint32_t *c2 = a.data(), *n2 = b2.data();
            for (uint32_t r = 0; r < 12; r++) {
                for (uint32_t seg = 0; seg < n_segs; seg++) {
                    const uint32_t *slot = padded.data() + (size_t)seg * (JDA_SEG_BYTES / 4);
                    n2[seg + 1] = rst ? jda_seg_walk<JDA_SEG_SPEC, true>(P, seg, c2[seg], slot, lt, S, ST) : jda_seg_walk<JDA_SEG_SPEC, false>(P, seg, c2[seg], slot, lt, S, ST);
                }
                n2[0] = 0;
                std::swap(c2, n2);
                uint32_t good = 0, dead = 0;
                for (uint32_t i = 0; i <= n_segs; i++) { good += c2[i] == truth[i]; dead += c2[i] == JDA_SEG_DEAD; }
                fprintf(stderr, "round %u: %u of %u states true, %u dead\n", r, good, n_segs + 1, dead);
            }
        }
        bool ok = settled;
        {   // the host's sums (jda_upload_batch)
            uint64_t g = 0;
            int32_t pred[3] = { 0, 0, 0 };
            uint32_t j = 0;
            for (uint32_t i = 0; i < n_segs; i++) {
                uint32_t *st = &seg_start[(size_t)i * 5];
                const uint32_t *su = &seg_sum[(size_t)i * JDA_SEG_SUM_WORDS];
                st[0] = g > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)g; st[1] = (uint32_t)pred[0]; st[2] = (uint32_t)pred[1]; st[3] = (uint32_t)pred[2]; st[4] = j;
                g += su[0];
                if (su[5] & 1u) {
                    if (g < (uint64_t)nb + 1) ok = false;
                    for (uint32_t r = i + 1; r < n_segs; r++) seg_start[(size_t)r * 5] = 0xfffffff0u;
                    break;
                }
                if (su[5] & JDA_SEG_HAS_RESTART) { pred[0] = (int32_t)su[1]; pred[1] = (int32_t)su[2]; pred[2] = (int32_t)su[3]; }   // an interval ended inside: its sums count from there
                else { pred[0] += (int32_t)su[1]; pred[1] += (int32_t)su[2]; pred[2] += (int32_t)su[3]; }
                j = (su[4] >> (3u * j)) & 7u;
            }
            if (g < (uint64_t)nb + 1) ok = false;
        }
        memset(&ST, 0, sizeof(ST));
        uint32_t terminal = 0;
        {                                                       // finalize + candidates (jda_segscan_finalize, jda_segscan_resolve_cands)
            jda_fin_acc A;
            A.bad = 0; A.terminal = 0; A.max_abs_dc = 0;
            for (uint32_t seg = 0; seg < n_segs; seg++) {
                const uint32_t *st = &seg_start[(size_t)seg * 5];
                if (st[0] > P.n_blocks_total) break;
                uint32_t nblk = seg_sum[(size_t)seg * JDA_SEG_SUM_WORDS];
                if (nblk > P.rec_cap) { nblk = P.rec_cap; A.bad = 1; }
                for (uint32_t i = 0; i < nblk; i++) jda_finalize_item(P, seg, i, st[0], st[0] % P.nblocks, jda_fin_recip(P.nblocks), (int32_t)st[1], (int32_t)st[2], (int32_t)st[3], jda_fin_rst_from(seg_sum[(size_t)seg * JDA_SEG_SUM_WORDS + 5]), A);
                const uint32_t mac = (seg_sum[(size_t)seg * JDA_SEG_SUM_WORDS + 5] >> 4) & 15u;
                if (mac > ST.max_ac_bits) ST.max_ac_bits = mac;
            }
            g_prescan_cands = stats[JDA_ST_NCAND];
            if (stats[JDA_ST_NCAND] > P.cand_cap) A.bad = 1;
            else for (uint32_t ci = 0; ci < stats[JDA_ST_NCAND]; ci++) ST.trunc_events += jda_resolve_item(P, ci);
            if (rst) for (uint32_t nr = 1; nr < P.n_intervals; nr++) {                                          // every marker where the MCU count puts it
                const uint32_t mis = jda_rst_event_item(P, nr);
                if (mis && getenv("HOSTSIM_DEBUG")) fprintf(stderr, "restart event %u of %u: ev %u/%u round %u, seg round %u g0 %u want %u\n", nr, P.n_intervals, rst_events[2 * nr] >> 11, rst_events[2 * nr] & 2047u, rst_events[2 * nr + 1],
                                                            seg_sum[(size_t)(rst_events[2 * nr] >> 11) * JDA_SEG_SUM_WORDS + 7], seg_start[(size_t)(rst_events[2 * nr] >> 11) * 5], nr * P.interval_blocks);
                ST.bad |= mis;
            }
            if (getenv("HOSTSIM_DEBUG")) fprintf(stderr, "finalize: bad %u terminal %u cands %u\n", A.bad, A.terminal, stats[JDA_ST_NCAND]);
            ST.bad |= A.bad; terminal = A.terminal; ST.max_abs_dc = A.max_abs_dc;
        }
        if (ok && !ST.bad && terminal == 1) {
            g_prescan_trunc = ST.trunc_events;
            jda_image_adopt_prescan(img, (uint32_t)(I->mcus_x * I->mcus_y), ST.max_ac_bits, (int32_t)ST.max_abs_dc, ST.trunc_events);
            g_prescan_used = 2;
            int32_t e2 = 0;
            jda_image *ref = jda_prepare(jpeg, len, &e2);
            uint32_t nn = 0;
            const uint32_t *hi = ref ? jda_image_block_index(ref, &nn) : NULL;
            // (the serial pre-scan writes the reference reader's phase into every entry, RECORD mode a canonical one into the entries of
            // blocks without a truncated read: equal = the same bit position and flag everywhere, the same entry where flagged)
            bool same_index = ref != NULL;
            for (size_t i = 0; i < nb && same_index; i++) {
                const uint32_t a = hi[i], b = dev_index[i];
                const uint32_t pa = (a >> JDA_INDEX_OFF_BITS) * 8u + (a & (JDA_INDEX_TRUNC - 1u)), pb = (b >> JDA_INDEX_OFF_BITS) * 8u + (b & (JDA_INDEX_TRUNC - 1u));
                same_index = pa == pb && (a & JDA_INDEX_TRUNC) == (b & JDA_INDEX_TRUNC) && (!(a & JDA_INDEX_TRUNC) || a == b);
            }
            if (same_index) {                                       // the closing entry: the device's bounds the serial one's from above, by 34 bits at most (+ 7 of an interval's rounding)
                const uint32_t a = hi[nb], b = dev_index[nb];
                const uint32_t pa = (a >> JDA_INDEX_OFF_BITS) * 8u + (a & 127u), pb = (b >> JDA_INDEX_OFF_BITS) * 8u + (b & 127u);
                same_index = pb >= pa && pb <= pa + 41u;
            }
            g_index_equal = same_index && memcmp(jda_image_block_dc(ref), dev_dc.data(), nb * 2) == 0 &&
                            jda_image_truncation_events(ref) == ST.trunc_events && jda_image_fast_mul(ref) == jda_image_fast_mul(img) ? 1 : 0;
            if (ref && getenv("HOSTSIM_DEBUG")) {
                for (size_t i = 0; i <= nb; i++) if (hi[i] != dev_index[i] || (i < nb && jda_image_block_dc(ref)[i] != dev_dc[i])) { fprintf(stderr, "first diff at block %zu of %zu: host %u/%u dc %d, dev %u/%u dc %d\n", i, nb, hi[i] >> 7, hi[i] & 127, i < nb ? jda_image_block_dc(ref)[i] : 0, dev_index[i] >> 7, dev_index[i] & 127, i < nb ? dev_dc[i] : 0); break; }
                fprintf(stderr, "rounds %u trunc host %u dev %u\n", rounds, jda_image_truncation_events(ref), ST.trunc_events);
                for (size_t i = 0; i < nb; i++) {                   // the first entry that differs in what counts (position, flag, flagged entry, DC value)
                    const uint32_t a = hi[i], b = dev_index[i];
                    const uint32_t pa = (a >> JDA_INDEX_OFF_BITS) * 8u + (a & (JDA_INDEX_TRUNC - 1u)), pb = (b >> JDA_INDEX_OFF_BITS) * 8u + (b & (JDA_INDEX_TRUNC - 1u));
                    const bool eq = pa == pb && (a & JDA_INDEX_TRUNC) == (b & JDA_INDEX_TRUNC) && (!(a & JDA_INDEX_TRUNC) || a == b) && (i == nb || jda_image_block_dc(ref)[i] == dev_dc[i]);
                    if (!eq) { fprintf(stderr, "first real difference at block %zu of %zu: host bit %u flag %u dc %d, dev bit %u flag %u dc %d (mcus ok host %u)\n", i, nb, pa, (a >> 6) & 1u, i < nb ? jda_image_block_dc(ref)[i] : 0, pb, (b >> 6) & 1u, i < nb ? dev_dc[i] : 0, nn); break; }
                }
            }
            if (ref) jda_image_free(ref);
        } else {                                                // corrupt / truncated: the serial pre-scan, as jda_upload_batch does
            if (getenv("HOSTSIM_DEBUG")) fprintf(stderr, "segment path rejected: ok %d settled %d rounds %u bad %u terminal %u\n", (int)ok, (int)settled, rounds, ST.bad, terminal);
            jda_image_run_host_prescan(img);
            dev_index.clear(); dev_dc.clear();
        }
    }
    jda_dev_desc D;
    jda_output o;
    o.pixels = out; o.pitch_bytes = pitch_bytes; o.width_px = width_px; o.rows = rows;
    int rc = jda_fill_desc(D, img, pixel_type, options, o);
    if (rc != JDA_SUCCESS) { jda_image_free(img); return rc; }
    uint32_t n;
    D.scan = jda_image_scan(img, &n);
    D.blk_index = jda_image_block_index(img, &n);
    D.blk_dc = jda_image_block_dc(img);
    if (g_prescan_used) { D.blk_index = dev_index.data(); D.blk_dc = dev_dc.data(); }
    if (g_use_cont && !g_prescan_used) { uint32_t nc = 0; D.blk_cont = jda_image_block_cont(img, &D.blk_cont_first, &nc); }
    D.tables = jda_image_tables(img, &n);
    std::vector<jda_strip> strips;
    jda_append_strips(strips, 0, D.mcus_x, D.mcus_y, D.mode);
    if (D.scale_shift == 2 && D.strip_mcus == 0) {                    // JDA_LIST_QUARTER (jda_list_index): the 1/4-scale kernel
        switch (D.mode) {
        case JDA_MODE_GRAY: run_quarter_tiles<JDA_MODE_GRAY>(D, strips); break;
        case JDA_MODE_444: run_quarter_tiles<JDA_MODE_444>(D, strips); break;
        case JDA_MODE_420: run_quarter_tiles<JDA_MODE_420>(D, strips); break;
        case JDA_MODE_422: run_quarter_tiles<JDA_MODE_422>(D, strips); break;
        default: run_quarter_tiles<JDA_MODE_440>(D, strips); break;
        }
    } else
    switch (D.mode * 2 + (D.fast_mul ? 1 : 0)) {
    case JDA_MODE_GRAY * 2: run_tiles<JDA_MODE_GRAY, false>(D, strips); break;
    case JDA_MODE_GRAY * 2 + 1: run_tiles<JDA_MODE_GRAY, true>(D, strips); break;
    case JDA_MODE_444 * 2: run_tiles<JDA_MODE_444, false>(D, strips); break;
    case JDA_MODE_444 * 2 + 1: run_tiles<JDA_MODE_444, true>(D, strips); break;
    case JDA_MODE_420 * 2: run_tiles<JDA_MODE_420, false>(D, strips); break;
    case JDA_MODE_420 * 2 + 1: run_tiles<JDA_MODE_420, true>(D, strips); break;
    case JDA_MODE_422 * 2: run_tiles<JDA_MODE_422, false>(D, strips); break;
    case JDA_MODE_422 * 2 + 1: run_tiles<JDA_MODE_422, true>(D, strips); break;
    case JDA_MODE_440 * 2: run_tiles<JDA_MODE_440, false>(D, strips); break;
    default: run_tiles<JDA_MODE_440, true>(D, strips); break;
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

// steps of every walk since the last call (the pairs of the walk's tables: fewer steps for the same symbols); off != 0: walk symbol by symbol
extern "C" unsigned long long hostsim_walk_steps(int off)
{
    const unsigned long long n = g_all_steps;
    g_all_steps = 0; g_pair_off = off != 0;
    return n;
}

// The pair halves of the walk's tables (jda_wt_pair) against two single look-ups: for every short key of every table and `tries`
// random continuations of the stream behind the key's ten bits, the symbol a walk would decode behind the first one -- with the key a
// walk derives from the stream there -- must be what the pair half says (bits, coefficients moved on by, magnitude size), whatever
// the bits the key does not show.  Returns the number of pairs checked, or -1 - (table << 12 | key) at the first disagreement.
extern "C" long hostsim_walk_pairs_check(const uint8_t *jpeg, int len, uint32_t tries, uint32_t seed)
{
    int32_t err = 0;
    jda_image *img = jda_prepare(jpeg, len, &err);
    if (!img) return -1;
    uint32_t tb = 0;
    const uint8_t *tables = jda_image_tables(img, &tb);
    const jda_image_info *I = jda_image_get_info(img);
    jda_segscan_params P;
    memset(&P, 0, sizeof(P));
    P.nblocks = (uint8_t)I->blocks_per_mcu; P.nluma = (uint8_t)(I->blocks_per_mcu - (I->ncomp == 3 ? 2 : 0));
    uint8_t q_id[3];
    jda_image_component_ids(img, P.dc_id, P.ac_id, q_id);
    const uint32_t follow = jda_wt_dc_follow(P);
    std::vector<uint64_t> store((JDA_WT_BYTES + 7) / 8);
    uint8_t *wt = (uint8_t *)store.data();
    for (uint32_t tid = 0; tid < 256; tid++) jda_walk_tables_from(tables, follow, tid, 256, wt);
    const uint32_t *T = (const uint32_t *)wt;
    long checked = 0;
    uint64_t rng = 0x9e3779b97f4a7c15ull ^ seed;
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t fa = t < 2 ? t : (follow >> (2u * (t - 2u))) & 3u;      // the AC table that decodes the second symbol
        for (uint32_t key = 0; key < 1024; key++) {
            const uint32_t e32 = T[t * 2048u + key], ea = e32 & 0xffffu, pd = e32 >> 16;
            if (!pd) continue;
            if (fa > 1u || JDA_AC_STOPS(ea)) { jda_image_free(img); return -1 - (long)(t << 12 | key); }
            const uint32_t bits_a = (ea >> 12) + 1u + ((ea >> 8) & 15u);
            for (uint32_t i = 0; i < tries; i++) {
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                const uint64_t stream = ((uint64_t)key << 54) | (rng >> 10);   // the key's ten bits, then anything
                const uint32_t w = (uint32_t)((stream << bits_a) >> 32);        // what a walk peeks behind the first symbol
                const uint32_t kb = w >= 0xfc000000u ? 1024u + ((w >> 16) & 1023u) : w >> 22;
                const uint32_t eb = T[fa * 2048u + kb] & 0xffffu;
                if ((eb & 0xffu) == JDA_AC_NONE) { jda_image_free(img); return -1 - (long)(t << 12 | key); }
                const bool eob = (eb & 0xffu) == JDA_AC_EOB;
                const uint32_t len_b = (eb >> 12) + 1u, sz_b = eob ? 0u : (eb >> 8) & 15u, dk = eob ? 0u : ((eb >> 1) & 15u) + 1u;
                if ((pd & 31u) != len_b + sz_b || ((pd >> 5) & 31u) != dk || ((pd >> 10) & 15u) != sz_b || !(pd & JDA_WT_PAIR_VALID)) {
                    jda_image_free(img);
                    return -1 - (long)(t << 12 | key);
                }
                checked++;
            }
        }
    }
    jda_image_free(img);
    return checked;
}
