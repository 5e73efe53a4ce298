// jda_device_core.h -- the per-lane decode logic of the HIP kernels.
//
// Everything here is written once and compiled twice: by hipcc into the gfx950 kernels of
// jda_kernels.hip, and by g++ into tests/hostsim (a lane-by-lane CPU emulation of one wavefront
// used ONLY by the unit tests to check the kernel logic where no GPU exists).  The product path
// never runs this on the CPU.
//
// One wavefront decodes one tile: up to 64 consecutive 8x8 blocks of one MCU row (10 MCUs of 4:2:0).
//   P1  lane = block: Huffman/RLE expand into the block's int16[64] slot in LDS   <- JPEGDecodeMCU  jpeg.inl:2090-2274
//       + the wave builds the IDCT work lists (prefix sum / ballots)
//   P2  lane = (block, non-empty column): dequant + IDCT column stage           <- JPEGIDCT       jpeg.inl:2553-2679
//   P3  lane = (block, row): IDCT row stage + range limit -> 64 bytes in place   <- JPEGIDCT       jpeg.inl:2680-2797
//                                                                                  DC-only bypass jpeg.inl:5146-5154
//   P4  lanes tile the output rows: YCbCr -> RGB565/RGB8888/gray, chroma        <- JPEGPutMCU*    jpeg.inl:2799-4868
//       upsample, 1/2-1/4-1/8, coalesced stores                                    JPEGPixel*     jpeg.inl:3101-3278
// plus the segment walk of the device pre-scan (jda_seg_walk) and the tables it stages (jda_walk_tables_from).
// The bit reader reproduces the reference's 64-bit window and refill rule exactly, so that the
// low-bit truncation of SURVEY.md fact 6 is reproduced by construction.
#ifndef JDA_DEVICE_CORE_H
#define JDA_DEVICE_CORE_H

#include <stdint.h>

#include "jda_internal.h"

#if defined(__HIPCC__)
#define JDA_HD __host__ __device__ __forceinline__
#else
#define JDA_HD static inline __attribute__((always_inline))
#endif

// ---- tile and LDS layout -------------------------------------------------------------------------
// One wavefront (64 threads) decodes one "tile": up to 64 consecutive 8x8 blocks of one MCU row =
// 10 MCUs of 4:2:0 (60 blocks), 20 MCUs of 4:4:4 (60 blocks) or 64 MCUs of a gray image.  The four
// wavefronts of a workgroup share nothing but the image's tables in LDS, so after the tables are
// staged there is no workgroup barrier: every phase boundary is a wave-local fence.
#define JDA_TILE_THREADS 64
#ifndef JDA_P1_TRACE
#define JDA_P1_TRACE(slot) ((void)0)      // profiling hook, defined by jda_kernels.hip
#endif
#define JDA_COEF_STRIDE 136      // bytes per block in LDS: 64 int16 + 8 pad (row reads stay 8-byte aligned)
// per-wave LDS window over the tile's slice of the filtered scan: jda_lds_layout<MODE>::WIN_BYTES.  Only P1 reads it, and the
// column work list is only alive from the end of P1 to the end of P2 -- so the two share their bytes: the next tile's slice
// is stored when this tile's column stage is over.

template <int MODE> struct jda_mode_traits;
template <> struct jda_mode_traits<JDA_MODE_GRAY> { enum { NLUMA = 1, NBLK = 1, MCU_W = 8, MCU_H = 8, MCU_W_LOG2 = 3 }; };
template <> struct jda_mode_traits<JDA_MODE_444>  { enum { NLUMA = 1, NBLK = 3, MCU_W = 8, MCU_H = 8, MCU_W_LOG2 = 3 }; };
template <> struct jda_mode_traits<JDA_MODE_420>  { enum { NLUMA = 4, NBLK = 6, MCU_W = 16, MCU_H = 16, MCU_W_LOG2 = 4 }; };
template <> struct jda_mode_traits<JDA_MODE_422>  { enum { NLUMA = 2, NBLK = 4, MCU_W = 16, MCU_H = 8, MCU_W_LOG2 = 4 }; };   // Y0 Y1 (side by side) Cb Cr
template <> struct jda_mode_traits<JDA_MODE_440>  { enum { NLUMA = 2, NBLK = 4, MCU_W = 8, MCU_H = 16, MCU_W_LOG2 = 3 }; };   // Y0 Y1 (one above the other) Cb Cr

// LDS copy of the table blob (one per workgroup): DC LUTs, the SHORT halves of the AC LUTs (codes
// that do not start with six 1 bits; the long halves are rare and stay in global memory),
// quantisers, zigzag.
#define JDA_LT_DC      0         // 2 x 1024
#define JDA_LT_AC      2048      // 2 tables x (1024 short + 1024 long) uint16, re-laid out while staging (jda_ac_entry): the long half
                                 // (codes starting 111111, the reference's usHuffAC[1024 + ..]) right behind the short one, so that ONE
                                 // 11-bit key finds an entry: (w >> 22) for a short code, (w >> 16) & 0x7ff = 1024 + the next 10 bits for a long one
// The prescaled quantiser tables (4 x 64 int16) sit in the DC LUTs' unused bytes: a DC LUT (the reference's layout, jpeg.inl:1098-1152)
// is indexed 0..61 and 128..255 (+ 512 for the folded values), so bytes 256..511 of each 1024-byte LUT are never read --
// room for two 128-byte tables each.  512 bytes less per workgroup is what lets 16 wavefronts of the 64-block tile layouts fit.
#define JDA_LT_QUANT_OFF(q) ((((q) >> 1) * 1024u) + 256u + (((q) & 1u) * 128u))
#define JDA_LT_NIB     96        // 16 x uint16: the set bits of a nibble, lowest first, as 3-bit fields (jda_p1_lists); same DC LUT bytes
#define JDA_LT_EOB     64        // 2 x uint32: JDA_TB_EOB, in never-read bytes of DC LUT 0
#define JDA_LT_ZZ      10240     // 144 x uint16, built while staging: where the coefficient at zigzag position j goes
#define JDA_ZZ_ENTRIES 144       //   j < 64: (column bit 1 << (n & 7)) << 8 | 2 n (n = natural index: the byte offset in the
                                 //   block); j >= 64 (past the block, or 64 + j for a symbol that stores nothing): 128 = the
                                 //   block's padding, no flags.  j <= 63 + 15 + 64.
#define JDA_ZZ_DUMP    134u      // (the last two of the slot's eight pad bytes; bytes 128..131 are the block's shared flag word in P1's chunked mode)
#define JDA_SLOT_FLAGS  128u      // byte offset in a block's slot: OR of the flag words of the chunks other lanes decoded (jda_p1c_*)
#define JDA_LT_BYTES   10528
#define JDA_LT_LONG_BYTES 0           // (the long halves are part of JDA_LT_AC now)

// AC LUT entry as the kernels keep it in LDS, made from the reference's (length << 8) | RS:
//   (length - 1) << 12 | S << 8 | Z,   Z = 2 (R + 64 nostore)  -- or 0xff for EOB, 0xfe for "no such code" (raw 0)
// (a code has 1..16 bits: four bits hold length - 1)
// nostore = a symbol with S == 0 that is not EOB (ZRL).  Z is the byte offset the symbol adds to the zigzag lookup
// (2-byte entries): R + 64 steers it to the padding entry, so the store needs no condition (jpeg.inl:2246-2256:
// "if (S && k < limit)"), and bits 4:1 of Z are still R for the position update.  One byte compare finds EOB.
#define JDA_AC_EOB 0xffu
#define JDA_AC_NONE 0xfeu
#define JDA_AC_STOPS(e) (((e) & 0xfeu) == 0xfeu)      // EOB or no code: the block's symbols end here
JDA_HD uint32_t jda_ac_entry(uint32_t raw)
{
    const uint32_t rs = raw & 0xffu, len = raw >> 8;
    if (len == 0u) return JDA_AC_NONE;
    if (rs == 0u) return ((len - 1u) << 12) | JDA_AC_EOB;
    const uint32_t r = rs >> 4, sz = rs & 0xfu;
    return ((len - 1u) << 12) | (sz << 8) | ((r + (sz == 0u ? 64u : 0u)) << 1);
}

#ifndef JDA_LONG_LDS_MODES
#define JDA_LONG_LDS_MODES 0x1fu     // bit per JDA_MODE_*: the layouts that stage the long AC halves
#endif
// The per-WAVE region.  BIG = 1: one wavefront less per workgroup, its LDS shared out as a larger scan window -- the kernel
// variant for high-bitrate images (a tile whose slice of the scan does not fit the window takes the general reader, which
// goes to HBM at every refill: 3-4x slower in P1).
template <int MODE, int BIG = 0> struct jda_lds_layout {
    enum {
        // MCUs per tile: 10 (4:2:0) / 20 (4:4:4) / 16 / 64.  4:4:4 takes 20 of the 21 that would fit: 160 pixels = 640-byte rows of
        // RGB8888 (whole 128-byte lines: 168-pixel tiles wrote 1.9 % more than they stored) and 320 colour-stage items = exactly five passes
        MCUS = MODE == JDA_MODE_444 ? 20 : JDA_TILE_THREADS / jda_mode_traits<MODE>::NBLK,
        BLOCKS = MCUS * jda_mode_traits<MODE>::NBLK,                 // blocks per tile: 60 / 63 / 64
        // one 136-byte slot per block: int16[64] coefficients, later (first 64 bytes) its 8x8 samples --
        // the row stage stores its bytes over the block it has just read, as the reference does (:2682)
        COEF_OFF = 0,
        ROWLIST_OFF = COEF_OFF + BLOCKS * JDA_COEF_STRIDE,          // 64 block ids (uint8), grouped by row class
        CNT_OFF = ROWLIST_OFF + JDA_TILE_THREADS,                   // 8 uint32 counters
        // the column list comes last, and the scan window lies over it and over everything that is left of the wavefront's
        // share of the LDS behind it.  (A block writes eight column items whatever it has (jda_p1_lists), up to seven past
        // the list's end: window bytes, dead by then.)
        COLLIST_OFF = (CNT_OFF + 32 + 15) / 16 * 16,                // uint16 items
        COLLIST_ENTRIES = BLOCKS * 8,                               // every column of every block
        WIN_OFF = COLLIST_OFF,
        BASE_BYTES = COLLIST_OFF + COLLIST_ENTRIES * 2 + 16,        // what a wavefront needs at least
        PLANE_OFF = COEF_OFF,
        PLANE_STRIDE = jda_mode_traits<MODE>::NBLK * JDA_COEF_STRIDE, // bytes between consecutive MCUs' samples
        // Codes that start 111111 -- every code of 10 bits and more of the Annex K tables -- are looked up in the long halves of
        // the AC LUTs.  They are some percent of the symbols, i.e. SOME lane of a wavefront meets one in most trips of the
        // entropy loop: from the table blob in HBM that was a global load per trip, behind a wait that also drains the
        // tile's output stores.  So the long halves are staged too
        LONG_LDS = 1,
        TAB_BYTES = JDA_LT_BYTES,
        // wavefronts per workgroup = per CU: as many as fit in the 160 KB of LDS next to one table copy (and the 16-byte draw
        // counter), 16 at most; every wavefront gets an equal share of what is there, and what it does not need is window
        LDS_FREE = 160 * 1024 - TAB_BYTES - 16,
        WAVES_MAX = LDS_FREE / BASE_BYTES > 16 ? 16 : LDS_FREE / BASE_BYTES,
        WAVES = WAVES_MAX - BIG,
        WAVE_BYTES = LDS_FREE / WAVES / 16 * 16,                    // 9,568 B (4:2:0), 10,208 B with BIG
        WIN_BYTES = WAVE_BYTES - WIN_OFF > 2048 ? 2048 : WAVE_BYTES - WIN_OFF,     // 1,312 B (4:2:0), 1,952 B with BIG
        WIN_CHUNKS = (WIN_BYTES + 1023) / 1024                      // 16-byte chunks a lane copies when the window is staged
    };
};

// ---- small helpers ---------------------------------------------------------------------------
// type-punned wide accesses to the int16 / uint8 LDS arrays must not be reordered by TBAA
typedef uint64_t __attribute__((may_alias)) jda_u64_alias;
typedef uint32_t __attribute__((may_alias)) jda_u32_alias;

JDA_HD uint32_t jda_alignbyte(uint32_t hi, uint32_t lo, uint32_t byte_shift)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, byte_shift);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (byte_shift & 3)));
#endif
}

// Hides a value from instruction selection.  Needed because hipcc (ROCm 7.2) fuses
//   clamp255(a >> 12) | clamp255(b >> 12) << 8   into gfx950's v_ashr_pk_u8_i32 and then treats
// the result as a zero-extended 16-bit value, but the instruction leaves garbage in bits 31:16
// (observed on MI355X: low bits of the blue channel corrupted).  Breaking the pattern costs nothing.
#if defined(__HIP_DEVICE_COMPILE__)
#define JDA_OPAQUE(x) asm("" : "+v"(x))
#define JDA_STORE_ORDER() asm volatile("" ::: "memory")      // the compiler keeps memory accesses on their side of it (LDS serves a wavefront's accesses in order)
#else
#define JDA_OPAQUE(x) ((void)0)
#define JDA_STORE_ORDER() ((void)0)
#endif

// ---- packed 16-bit helpers (two pixels per VALU instruction in the colour stage) ---------------
// v_perm_b32: result byte i = byte (sel >> 8i & 0xff) of the 8-byte pool {hi = bytes 4-7, lo = bytes 0-3};
// selector 0x0c gives 0x00 and 0x0d gives 0xff.
JDA_HD uint32_t jda_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t k = (sel >> (8 * i)) & 0xffu;
        const uint32_t b = k < 8 ? (uint32_t)(pool >> (8 * k)) & 0xffu : (k == 0x0c ? 0u : 0xffu);
        r |= b << (8 * i);
    }
    return r;
#endif
}
// two independent 16-bit adds (v_pk_add_u16)
JDA_HD uint32_t jda_pk_add16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short jda_us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, a) + __builtin_bit_cast(jda_us2, b));
#else
    return ((a + b) & 0xffffu) | ((a & 0xffff0000u) + (b & 0xffff0000u));
#endif
}
// two independent 16-bit subtracts (v_pk_sub_u16)
JDA_HD uint32_t jda_pk_sub16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short jda_us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_us2, a) - __builtin_bit_cast(jda_us2, b));
#else
    return ((a - b) & 0xffffu) | (((a & 0xffff0000u) - (b & 0xffff0000u)) & 0xffff0000u);
#endif
}
// a.lo + b.hi and a.hi + b.hi (v_pk_add_u16 with op_sel: the second operand's UPPER half feeds both lanes -- a value that sits in
// bits 31:16 of a word is added to a pixel pair without being copied into the lower half first)
JDA_HD uint32_t jda_pk_add16_bhi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    const uint32_t h = b >> 16;
    return ((a + h) & 0xffffu) | ((a + (h << 16)) & 0xffff0000u);
#endif
}
// a.lo + b.lo, a.lo + b.hi  /  a.hi + b.lo, a.hi + b.hi (v_pk_add_u16 with op_sel on the FIRST operand: one half of it feeds both lanes --
// one pixel's luma sample is added to its red and green terms in one instruction)
JDA_HD uint32_t jda_pk_add16_alo(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    const uint32_t l = a & 0xffffu;
    return ((l + b) & 0xffffu) | (((l << 16) + (b & 0xffff0000u)) & 0xffff0000u);
#endif
}
JDA_HD uint32_t jda_pk_add16_ahi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    const uint32_t h = a >> 16;
    return ((h + b) & 0xffffu) | (((h << 16) + (b & 0xffff0000u)) & 0xffff0000u);
#endif
}
// Two bytes of LDS `delta` apart, at p + step, read through an address of their own (ds_read_u8 x 2): the compiler joins neighbouring
// byte loads into wider loads and spends a VALU instruction per byte on taking them apart again -- the colour stage is bound by
// its VALU instructions more than by its LDS accesses.  The address is made where it is used (one add; asm volatile: a copy kept in
// a register across the tile loop per pass of the colour stage sent the general kernels' registers to scratch memory).
JDA_HD void jda_lds_bytes_apart(const uint8_t *p, uint32_t step, uint32_t delta, uint32_t &b0, uint32_t &b1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t a;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"((uint32_t)(uintptr_t)p), "v"(step));
    const uint8_t __attribute__((address_space(3))) *q = (const uint8_t __attribute__((address_space(3))) *)a;
    b0 = q[0]; b1 = q[delta];
#else
    b0 = p[step]; b1 = p[step + delta];
#endif
}
// a * b + c on operands that fit in 24 signed bits (v_mad_i32_i24, full rate)
JDA_HD int32_t jda_mad24(int32_t a, int32_t b, int32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b) + c;
#else
    return (int32_t)((uint32_t)a * (uint32_t)b + (uint32_t)c);
#endif
}
// per 16-bit lane: the sign-extended 10-bit field at bits 14:5  (v_pk_lshlrev_b16 1, v_pk_ashrrev_i16 6)
JDA_HD uint32_t jda_pk_sext10_at5(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short jda_s2 __attribute__((ext_vector_type(2)));
    jda_s2 v = __builtin_bit_cast(jda_s2, a);
    v = v << 1;
    v = v >> 6;
    return __builtin_bit_cast(uint32_t, v);
#else
    const int32_t lo = (int16_t)(uint16_t)((a & 0xffffu) << 1) >> 6, hi = (int16_t)(uint16_t)(((a >> 16) & 0xffffu) << 1) >> 6;
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
#endif
}
// per 16-bit lane: the range limit of a sample, ucRangeTable[(v >> 5) & 0x3ff] = clamp(sext10(v >> 5) + 128) (jpeg.inl:159-222), in
// three packed instructions: v * 2 + 0x8000 modulo 2^16 (v_pk_mad_u16) puts the 10-bit field at bits 15:6 in OFFSET-BINARY form
// (signed value + 512; bit 15 of v, which the table's index drops, falls out), subtracting 384 << 6 with unsigned saturation
// (v_pk_sub_u16 clamp) leaves max(value + 128, 0) << 6 in 0..639 << 6, and x 4 with unsigned saturation (v_pk_mad_u16 clamp) caps
// it at 255 in the lane's HIGH byte (the low byte is junk; the byte permute that assembles a row picks bytes 1 and 3).
// (Rounds 3-5 added the offset -- 512 << 5 -- to the DC term of the column stage instead: a compare and a select per column item
// that the multiply-add's third operand does for nothing.)
JDA_HD uint32_t jda_pk_limit10(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short jda_us2 __attribute__((ext_vector_type(2)));
    uint32_t r, flip = 0x80008000u, four = 4u;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(flip));      // the field to bits 15:6, its sign bit flipped
    jda_us2 v = __builtin_bit_cast(jda_us2, r);
    const jda_us2 k = { 24576, 24576 };
    v = __builtin_elementwise_sub_sat(v, k);                // max(value + 128, 0) << 6          (v_pk_sub_u16 clamp)
    r = __builtin_bit_cast(uint32_t, v);
    asm("v_pk_mad_u16 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(r), "v"(four));   // x 4, saturating: high byte = min(.., 255)
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
        uint32_t x = (((a >> (16 * h)) << 1) + 0x8000u) & 0xffffu;
        x = x > 24576u ? x - 24576u : 0u;
        x *= 4u;
        if (x > 0xffffu) x = 0xffffu;
        r |= x << (16 * h);
    }
    return r;
#endif
}
// two signed 16-bit values -> two bytes saturated to 0..255 in bits 15:0, bits 31:16 zero (v_sat_pk_u8_i16)
JDA_HD uint32_t jda_sat_pk_u8(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(a));
    return r;
#else
    const int32_t lo = (int16_t)(a & 0xffffu), hi = (int16_t)(a >> 16);
    const uint32_t l = lo < 0 ? 0u : (lo > 255 ? 255u : (uint32_t)lo), h = hi < 0 ? 0u : (hi > 255 ? 255u : (uint32_t)hi);
    return l | (h << 8);
#endif
}

// low 32 bits of the product of two values below 2^24 (v_mul_u32_u24, full rate)
JDA_HD uint32_t jda_umul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (a & 0xffffffu) * (b & 0xffffffu);
#endif
}
// a byte in all four bytes of a word (one v_perm_b32 where `x * 0x01010101` is a quarter-rate v_mul_lo_u32)
JDA_HD uint32_t jda_dup8(uint32_t v) { return jda_perm(0, v, 0x00000000u); }
// sum of the four byte products a.b[i] * b.b[i]  (v_dot4_u32_u8)
JDA_HD uint32_t jda_udot4(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, 0u, false);
#else
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return r;
#endif
}
JDA_HD uint32_t jda_udot4_acc(uint32_t a, uint32_t b, uint32_t c)      // a . b + c (bytes, unsigned)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return jda_udot4(a, b) + c;
#endif
}
JDA_HD uint32_t jda_dup16(int32_t v) { return jda_perm(0, (uint32_t)v, 0x01000100u); }        // low half in both halves
JDA_HD uint32_t jda_pack16(int32_t lo, int32_t hi) { return jda_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

// sign-extended 10-bit field starting at bit `lo` (the reference's "& 0x3ff" table index)
JDA_HD int32_t jda_sext10_at(int32_t v, int lo) { return (int32_t)((uint32_t)v << (22 - lo)) >> 22; }
JDA_HD int32_t jda_clamp255(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// ucRangeTable[(v >> 5) & 0x3ff]  (jpeg.inl:159-222, 2786-2793)
JDA_HD uint32_t jda_range_limit5(int32_t v) { return (uint32_t)jda_clamp255(jda_sext10_at(v, 5) + 128); }

// 64-bit big-endian window at an arbitrary byte position (MOTOLONG, src/JPEGDEC.h:316-318),
// assembled from three aligned dword loads.
JDA_HD uint64_t jda_be64_from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t pos)
{
    // byte alignment and byte swap in one permute per half: result byte 3 = stream byte sh, .. byte 0 = stream byte sh + 3
    // 0x00010203 + (pos & 3) * 0x01010101 without the multiply: four bytes of the descending run 6 5 4 3 2 1 0
    const uint32_t sel = jda_alignbyte(0x00000102u, 0x03040506u, ~pos);
    return ((uint64_t)jda_perm(w1, w0, sel) << 32) | jda_perm(w2, w1, sel);
}

// The workgroup's view of the filtered scan: bytes [win_lo, win_lo + win_len) are staged in LDS (copied
// with coalesced 16-byte loads by jda_window_fill); anything beyond is read from HBM directly.
struct jda_bitreader {
    const uint8_t JDA_GLOBAL *base;     // filtered scan in global memory (4-byte aligned, zero padded)
    const uint8_t *win;      // LDS copy
    uint32_t win_lo;         // 16-byte aligned scan offset of win[0]
    uint32_t win_len;        // bytes staged (multiple of 16)
    uint32_t pos;            // bb.pBuf - start of filtered scan
    uint32_t off;            // bb.ulBitOff
    uint64_t bits;           // bb.ulBits
};

JDA_HD uint64_t jda_load_be64(const jda_bitreader &br, uint32_t pos)
{
    const uint32_t a = pos & ~3u;
    const uint32_t rel = a - br.win_lo;
    if (rel + 12u <= br.win_len) {                      // (a < win_lo wraps to a huge rel: falls through)
        // the LDS window holds the stream as byte-swapped dwords (jda_window_store): only the byte alignment is left
        const jda_u32_alias *p = (const jda_u32_alias *)(br.win + rel);
        const uint32_t sel = jda_alignbyte(0x08070605u, 0x04030201u, ~pos);      // 0x07060504 - (pos & 3) * 0x01010101
        return ((uint64_t)jda_perm(p[0], p[1], sel) << 32) | jda_perm(p[1], p[2], sel);
    }
    const jda_u32_alias JDA_GLOBAL *p = (const jda_u32_alias JDA_GLOBAL *)(br.base + a);
    return jda_be64_from_words(p[0], p[1], p[2], pos);
}

JDA_HD void jda_refill(jda_bitreader &br)
{
    if (br.off > 47) {                                   // jpeg.inl:2110-2114 (REGISTER_WIDTH-17)
        br.pos += br.off >> 3;
        br.off &= 7;
        br.bits = jda_load_be64(br, br.pos);
    }
}

// Cooperative copy of the tile's part of the scan into the wave's LDS window: lane l copies the
// 16-byte chunks l, l+64, ...  Every lane of the wave calls it; a wave fence follows.
struct jda_chunk16 { uint32_t w[4]; };
typedef jda_chunk16 __attribute__((may_alias)) jda_chunk16_alias;
// the same copy split in two so the HBM load can be issued early and the LDS store done late
// (the window is NCH x 64 lanes x 16 bytes at most: lane l copies chunks l, l + 64, ..)
template <int NCH> struct jda_chunks { jda_chunk16 c[NCH]; };
template <int NCH>
JDA_HD jda_chunks<NCH> jda_window_load(const uint8_t JDA_GLOBAL *scan, uint32_t win_lo, uint32_t win_len, uint32_t lane)
{
    jda_chunks<NCH> r;
    const jda_chunk16_alias JDA_GLOBAL *src = (const jda_chunk16_alias JDA_GLOBAL *)(scan + win_lo);
    // Every lane loads: one whose chunk lies behind the slice reads the slice's last chunk again (jda_window_store stores what belongs
    // to the window only).  A load under a lane condition, merged with zeros for the others, came out of the compiler as load -> wait ->
    // copy: the wavefront sat out the load's latency where the column stage was meant to cover it.
    const uint32_t n16 = win_len >> 4, last = n16 ? n16 - 1u : 0u;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const uint32_t i = lane + 64u * (uint32_t)k;
        const jda_chunk16_alias v = src[i < last ? i : last];
        r.c[k].w[0] = v.w[0]; r.c[k].w[1] = v.w[1]; r.c[k].w[2] = v.w[2]; r.c[k].w[3] = v.w[3];
    }
    return r;
}
template <int NCH>
JDA_HD void jda_window_store(uint8_t *win, uint32_t win_len, uint32_t lane, const jda_chunks<NCH> &r)
{
    jda_chunk16_alias *dst = (jda_chunk16_alias *)win;
#pragma unroll
    for (int k = 0; k < NCH; k++)
        if (lane + 64u * (uint32_t)k < (win_len >> 4)) {
            jda_chunk16_alias v;                        // every dword with the stream's first byte on top: P1's bit buffer takes them as they are
            v.w[0] = __builtin_bswap32(r.c[k].w[0]); v.w[1] = __builtin_bswap32(r.c[k].w[1]); v.w[2] = __builtin_bswap32(r.c[k].w[2]); v.w[3] = __builtin_bswap32(r.c[k].w[3]);
            dst[lane + 64u * (uint32_t)k] = v;
        }
}

JDA_HD void jda_window_fill(const uint8_t JDA_GLOBAL *scan, uint32_t win_lo, uint32_t win_len, uint8_t *win, uint32_t lane)
{
    const jda_chunk16_alias JDA_GLOBAL *src = (const jda_chunk16_alias JDA_GLOBAL *)(scan + win_lo);
    jda_chunk16_alias *dst = (jda_chunk16_alias *)win;
    for (uint32_t i = lane; i < (win_len >> 4); i += JDA_TILE_THREADS) {
        jda_chunk16_alias v = src[i];
        v.w[0] = __builtin_bswap32(v.w[0]); v.w[1] = __builtin_bswap32(v.w[1]); v.w[2] = __builtin_bswap32(v.w[2]); v.w[3] = __builtin_bswap32(v.w[3]);
        dst[i] = v;
    }
}

// EXTEND of the next s bits of the (un-refilled) window (jpeg.inl:2249-2252, 2155-2158)
JDA_HD int32_t jda_take_extend(uint64_t bits, uint32_t off, uint32_t s)
{
    const uint32_t top = (uint32_t)((bits << off) >> 32);      // zeros enter when off + s > 64
    const uint32_t v = top >> (32 - s);
    return (top & 0x80000000u) ? (int32_t)v : (int32_t)v - (int32_t)((1u << s) - 1u);
}

// ---- wave-level combine (list building without LDS atomics) -------------------------------------
// On the GPU these are DPP / ballot operations over the 64 lanes of the wavefront.  The host emulator
// steps lanes one after another, so it hands in the value of every lane (all[]) and the same results
// are computed by plain loops.
// Exclusive prefix sum of v over the lanes below `lane`, and the sum over all lanes.
JDA_HD uint32_t jda_wave_excl_sum(uint32_t v, uint32_t lane, const uint32_t *all, uint32_t &total)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)all; (void)lane;
    int x = (int)v;                                  // inclusive scan: row_shr 1,2,4,8 then row_bcast 15 / 31 (gfx9 DPP)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
    return (uint32_t)x - v;
#else
    uint32_t below = 0, sum = 0;
    for (uint32_t l = 0; l < JDA_TILE_THREADS; l++) { if (l < lane) below += all[l]; sum += all[l]; }
    total = sum;
    return below;
#endif
}
// How many lanes below `lane` have cls[l] == c, and how many lanes in all (c = 0..3; cls 4 = none).
JDA_HD uint32_t jda_wave_class_rank(uint32_t my_cls, uint32_t c, uint32_t lane, const uint32_t *all_cls, uint32_t &count)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)all_cls; (void)lane;
    const uint64_t m = __builtin_amdgcn_ballot_w64(my_cls == c);
    count = (uint32_t)__builtin_popcountll(m);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    uint32_t below = 0, n = 0;
    for (uint32_t l = 0; l < JDA_TILE_THREADS; l++) if (all_cls[l] == c) { n++; if (l < lane) below++; }
    count = n;
    return below;
#endif
}

JDA_HD uint32_t jda_popcount8(uint32_t v)
{
    return (uint32_t)__builtin_popcount(v & 0xffu);
}

// ---- Huffman / RLE expand of one 8x8 block (jpeg.inl:2090-2274) ------------------------------
// LIMIT: 64 = store every coefficient; 5 = 1/4 scale (zigzag 1..4 only, jpeg.inl:2117-2119);
//        1 = 1/8 scale: only the DC term is wanted.
// Because every block has its own index entry, decoding may stop as soon as the wanted
// coefficients are known (the reference has to walk to EOB to find the next block).
// coef: the block's int16[64] in LDS (natural order).  Returns the reference's u16MCUFlags.
struct jda_tables {
    const uint8_t *dc;        // LDS: 1024-byte DC LUT of this block's component
    const uint16_t *ac_short; // LDS: 1024 entries in the jda_ac_entry layout
    const uint16_t JDA_GLOBAL *ac_long;  // global: 1024 entries (codes starting 111111), the reference's layout
    const uint16_t *ac_long_lds;         // LDS: the same in the jda_ac_entry layout (layouts with LONG_LDS)
    const uint16_t *zz;       // LDS: JDA_ZZ_ENTRIES entries (see JDA_LT_ZZ)
    uint32_t eob_sh, eob_code;   // the window-only reader finds EOB by comparing stream bits (jda_lane_pre)
};

// The index entry of a block is the reader at its FIRST AC SYMBOL (after the refill at the top of the AC loop, jpeg.inl:2225-2230) and
// its DC value comes with it (index format 2: both pre-scans decode the DC symbol, jpeg.inl:2129-2165, as they pass it), so a block's
// decode starts with coefficient 1.
// exact: the block is flagged JDA_INDEX_TRUNC -- its index entry is the reference reader's true phase, and magnitude reads lose the
// bits the reference's window does not hold.  An unflagged block's entry may be canonical (the device pre-scan's: same bit
// position, another phase), and the reference truncates nothing in it: the reader refills instead of reading short.
template <int LIMIT>
JDA_HD uint32_t jda_decode_block(jda_bitreader &br, const jda_tables &T, int16_t *coef, int32_t dc, bool exact = true)
{
    uint32_t flags = 0;
    if (LIMIT == 64) {
        jda_u64_alias *z = (jda_u64_alias *)coef;       // memset(pMCU, 0, 128)  :2121
#pragma unroll
        for (int i = 0; i < 16; i++) z[i] = 0;
    } else if (LIMIT == 5) {
        coef[1] = 0; coef[8] = 0; coef[9] = 0;          // :2118
    }
    coef[0] = (int16_t)dc;
    // AC  (:2223-2265).  The reference refills at the top AND the bottom of every iteration; the top
    // one is a no-op after a bottom one, so one refill before the loop + one per iteration is identical.
    uint32_t code, e;
    int k = 1;
    jda_refill(br);
    while (k < LIMIT) {
        code = (uint32_t)(br.bits >> (48 - br.off)) & 0xffffu;
        if (code >= 0xfc00u) e = T.ac_long_lds[code & 0x3ffu];               // usHuffAC[1024 + ...]  :2232-2233
        else e = T.ac_short[code >> 6];
        if (JDA_AC_STOPS(e)) break;                     // EOB (no refill follows; the block's reader state is not needed any more)
        br.off += (e >> 12) + 1u;
        k += (int)((e >> 1) & 0xfu);
        const uint32_t ms = (e >> 8) & 0xfu;
        if (!exact && br.off + ms > 64u) { br.pos += br.off >> 3; br.off &= 7u; br.bits = jda_load_be64(br, br.pos); }
        if (k < LIMIT && ms) {
            const uint32_t n = (T.zz[k] & 0xffu) >> 1;
            flags |= (1u << (n & 7u)) | (n << 8);
            coef[n] = (int16_t)jda_take_extend(br.bits, br.off, ms);
        }
        br.off += ms;
        k++;
        jda_refill(br);
    }
    return flags;
}

// ---- the same, for a tile whose whole scan slice is in the LDS window (the normal case) ----------
// No lane can leave the window, so the reader needs no bounds check and no HBM path, and the block
// decode below is written without divergent branches: P1 is bound by the latency of its dependent
// chain (stream bits -> LUT -> bit offset -> stream bits), not by instruction issue, and every
// exec-mask region in that chain costs a VALU->SALU->branch round trip.
// x >> n for n in 0..32 (hardware shifts take n mod 32; a result for n == 32 is never used)
JDA_HD uint32_t jda_shr_upto32(uint32_t x, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return x >> (n & 31u);
#else
    return n >= 32 ? 0u : x >> n;
#endif
}
// EXTEND (jpeg.inl:2249-2252) of the s bits at the top of the 32-bit word t
JDA_HD int32_t jda_extend_top(uint32_t t, uint32_t s)
{
    const uint32_t v = jda_shr_upto32(t, 32u - s);
    const uint32_t neg = ~(uint32_t)((int32_t)t >> 31);          // all ones when the leading bit is 0
    return (int32_t)(v + (neg & ((0xffffffffu << s) + 1u)));
}

// The reader of the window-only path (the window in LDS holds byte-swapped dwords, so they are taken as they are).
// The window-only reader keeps ONE number: the position of the last consumed bit (GPU: counted from LDS address 0, so that the
// dword it sits in is at byte address (m >> 3) & ~3).  A peek reads that dword and the next (one ds_read2_b32) and funnel-shifts;
// consuming n bits is m += n.  Five instructions a symbol where sliding hi / lo / next by selects took nine; the price is a second
// LDS read on the symbol's dependent chain.
struct jda_wreader {
    uint32_t m;              // bits up to and including the last consumed one (window start = LDS address * 8; host: = 0), minus 1
    const uint8_t *base;     // host emulator: the window's first byte
};
// a + (b & 0xff) in one instruction (the byte select rides on the add: SDWA)
JDA_HD uint32_t jda_add_byte0(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return a + (b & 0xffu);
#endif
}
JDA_HD uint32_t jda_bfe(uint32_t v, uint32_t off, uint32_t width)          // v_bfe_u32: (v >> off) & ((1 << width) - 1), off + width <= 32
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return (v >> off) & ((1u << width) - 1u);
#endif
}
JDA_HD uint32_t jda_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)      // low 32 bits of {hi, lo} >> sh, sh = 0..31
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
#endif
}
JDA_HD void jda_wr_init(jda_wreader &R, const uint8_t *wbase, uint32_t pos, uint32_t off)
{
    // (for bit 0 of the window the last consumed bit is in the four bytes in front of it -- other LDS data, all "consumed")
#if defined(__HIP_DEVICE_COMPILE__)
    R.m = ((uint32_t)(uintptr_t)wbase << 3) + (pos << 3) + off - 1u;       // (the window is 16-byte aligned and not at LDS address 0)
    R.base = wbase;
#else
    R.m = (pos << 3) + off - 1u;
    R.base = wbase;
#endif
}
JDA_HD uint32_t jda_wr_peek(const jda_wreader &R)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t a = (R.m >> 3) & ~3u;
    const uint32_t __attribute__((address_space(3))) *p = (const uint32_t __attribute__((address_space(3))) *)a;
    return jda_alignbit(p[0], p[1], ~R.m);                         // (- (m + 1)) & 31 bits of the first dword are consumed
#else
    const uint8_t *p = R.base + (((int32_t)R.m >> 5) * 4);
    return jda_alignbit(*(const jda_u32_alias *)p, *(const jda_u32_alias *)(p + 4), ~R.m);
#endif
}
JDA_HD void jda_wr_consume(jda_wreader &R, uint32_t n) { R.m += n; }
// the reference's refill (jpeg.inl:2110-2114) as far as its ulBitOff is concerned
JDA_HD uint32_t jda_ref_refill(uint32_t roff) { return roff > 47u ? (roff & 7u) : roff; }

// LDS accesses by 32-bit address in the entropy loop (GPU): the zigzag lookup and the deferred coefficient store take their
// address from an SDWA add (a + byte 0 of b), which the compiler can only keep in the LDS address space if it is told so
#if defined(__HIP_DEVICE_COMPILE__)
#define JDA_LDS_A32(p) ((uint32_t)(uintptr_t)(p))
#define JDA_LDS_LOAD_U16(base_ptr, a32) ((uint32_t)*(const uint16_t __attribute__((address_space(3))) *)(a32))
#define JDA_COEF_STORE(coef, t, v) (*(int16_t __attribute__((address_space(3))) *)jda_add_byte0(JDA_LDS_A32(coef), (t)) = (int16_t)(v))
#else
#define JDA_LDS_A32(p) 0u
#define JDA_LDS_LOAD_U16(base_ptr, a32) ((uint32_t)*(const uint16_t *)((const uint8_t *)(base_ptr) + (a32)))
#define JDA_COEF_STORE(coef, t, v) (*(int16_t *)((uint8_t *)(coef) + ((t) & 0xffu)) = (int16_t)(v))
#endif
// EXACT = false leaves the reference's ulBitOff out (five instructions and a branch less per symbol): only right for a block
// in which the reference truncates no magnitude read, i.e. one whose index entry lacks JDA_INDEX_TRUNC.
// zero_fill: clear the block first.
// trunc: this lane's block is the flagged one (EXACT runs for the whole wavefront when any lane's is).
template <int LIMIT, bool EXACT, bool LONG_LDS>
JDA_HD uint32_t jda_decode_block_win(uint32_t pos, uint32_t off, const uint8_t *wbase, const jda_tables &T, int16_t *coef, int32_t dc, bool zero_fill, bool trunc = true)
{
    jda_wreader R;
    jda_wr_init(R, wbase, pos, off);
    uint32_t fl = 0;                                     // OR of zz entries of the stored coefficients
    uint32_t roff = EXACT ? jda_ref_refill(off) : 0u;   // the reference's ulBitOff at the block's first AC symbol
    // (the pre-scan decoded the DC symbol, jpeg.inl:2129-2165: the entry points behind it, the value rides on the block's clearing)
    if (LIMIT == 64 && zero_fill) {
#if defined(__HIP_DEVICE_COMPILE__)
        // (the block's address in ONE register the compiler cannot see through: sixteen stores at offsets 0 .. 120 from it -- left to
        // itself it folds the wavefront's constant part of the address into every store and adds it back seven times)
        uint32_t za = JDA_LDS_A32(coef);
        asm volatile("" : "+v"(za));
        jda_u64_alias __attribute__((address_space(3))) *z = (jda_u64_alias __attribute__((address_space(3))) *)za;
#else
        jda_u64_alias *z = (jda_u64_alias *)coef;
#endif
        z[0] = (uint64_t)(uint16_t)dc;
#pragma unroll
        for (int i = 1; i < 16; i++) z[i] = 0;
    } else {
        if (LIMIT == 5) { coef[1] = 0; coef[8] = 0; coef[9] = 0; }
        coef[0] = (int16_t)dc;
    }
    uint32_t w, e;
    const uint32_t zzb = JDA_LDS_A32(T.zz);              // (GPU: the zigzag table's LDS address rides in k2, so a lookup's address is one add)
    uint32_t k2 = zzb + 2;                               // twice the zigzag position: the byte offset into the zigzag table
    // EOB is recognised on the stream bits themselves -- one code per table, checked by the host (JDA_DESC_GENERAL_P1) --
    // before the symbol is looked up: a block costs one trip per coefficient symbol, none for its EOB, and the wavefront
    // runs as many trips as its longest block needs.  (A block of a decoded MCU holds no invalid code: the pre-scan ends
    // the image at the first bad MCU.)
    // A symbol's value is stored one trip later, while the next symbol's lookup is under way: the zigzag entry it needs was
    // asked for a trip earlier and LDS answers in order, so the trip waits for LDS once, not twice.  (The first trip stores a
    // zero to the block's padding.)
    uint32_t t_prev = JDA_ZZ_DUMP;
    int32_t v_prev = 0;
    w = jda_wr_peek(R);
    if ((w >> T.eob_sh) != T.eob_code) for (;;) {
        // codes starting 111111 (usHuffAC[1024 + ...], :2232-2233) sit right behind the short half: one 11-bit key, one lookup,
        // no branch -- the key is a bit field whose position depends on the code's class
        e = T.ac_short[jda_bfe(w, w >= 0xfc000000u ? 16u : 22u, 11u)];
        fl |= t_prev;
        JDA_COEF_STORE(coef, t_prev, v_prev);
        // the zigzag lookup decides where the value goes: position k + R of the block, or the padding when that is past
        // the block or the symbol carries no value (ZRL: the entry's low byte reads 2 (R + 64))
        const uint32_t kk2 = jda_add_byte0(k2, e);
        uint32_t t = JDA_LDS_LOAD_U16(T.zz, kk2);
        if (LIMIT != 64 && kk2 >= zzb + 2u * (uint32_t)LIMIT) t = JDA_ZZ_DUMP;     // 1/4 scale keeps zigzag 1..4 only (:2117-2119)
        const uint32_t len = (e >> 12) + 1u, ms = (e >> 8) & 0xfu;
        uint32_t m = w << len;
        const uint32_t n = len + ms;
        if (EXACT) {
            roff += n;                                   // the reference's ulBitOff after its magnitude read (:2249-2252)
            if (__builtin_expect(roff > 64u && trunc, 0)) m &= ~(0xffffffffu >> (64u + ms - roff));      // its window ended inside the magnitude
            roff = jda_ref_refill(roff);
        }
        v_prev = jda_extend_top(m, ms);
        t_prev = t;
        jda_wr_consume(R, n);
        k2 += (e & 0x1eu) + 2u;
        w = jda_wr_peek(R);
        if (k2 >= zzb + 2u * (uint32_t)LIMIT || (w >> T.eob_sh) == T.eob_code) break;
    }
    fl |= t_prev;
    JDA_COEF_STORE(coef, t_prev, v_prev);
    // A.2: column bits in 7:0, (n << 8) bits above -- only bit 13 (some n >= 32) is ever tested; fl holds 2n in 7:0
    return (fl >> 8) | ((fl & 0x40u) << 7);
}

// ---- a CHUNK of a block: at most JDA_CONT_SYMS AC symbols from a continuation entry (or from the block's first AC symbol), so that
// the lanes of a wavefront share its long blocks (jda_p1c_*).  The loop is jda_decode_block_win's; it starts at zigzag position k0 and
// ends at EOB, behind coefficient 63 or after max_syms symbols.  Returns the OR of the zigzag entries of what it stored (the caller
// folds the chunks' words into the block's flags: jda_p1c_fold).  Only for blocks the reference reads without truncation.
JDA_HD uint32_t jda_decode_chunk_win(uint32_t bitpos, const uint8_t *wbase, const jda_tables &T, int16_t *coef, uint32_t k0, uint32_t max_syms)
{
    jda_wreader R;
    jda_wr_init(R, wbase, bitpos >> 3, bitpos & 7u);
    uint32_t fl = 0;
    const uint32_t zzb = JDA_LDS_A32(T.zz);
    uint32_t k2 = zzb + 2u * k0;
    uint32_t t_prev = JDA_ZZ_DUMP;
    int32_t v_prev = 0;
    uint32_t w = jda_wr_peek(R), left = max_syms;
    if ((w >> T.eob_sh) != T.eob_code) for (;;) {
        const uint32_t e = T.ac_short[jda_bfe(w, w >= 0xfc000000u ? 16u : 22u, 11u)];
        fl |= t_prev;
        JDA_COEF_STORE(coef, t_prev, v_prev);
        const uint32_t kk2 = jda_add_byte0(k2, e);
        const uint32_t t = JDA_LDS_LOAD_U16(T.zz, kk2);
        const uint32_t len = (e >> 12) + 1u, ms = (e >> 8) & 0xfu;
        const uint32_t m = w << len;
        v_prev = jda_extend_top(m, ms);
        t_prev = t;
        jda_wr_consume(R, len + ms);
        k2 += (e & 0x1eu) + 2u;
        w = jda_wr_peek(R);
        left--;
        if (left == 0u || k2 >= zzb + 128u || (w >> T.eob_sh) == T.eob_code) break;
    }
    fl |= t_prev;
    JDA_COEF_STORE(coef, t_prev, v_prev);
    return fl;
}
// the block's flags (A.2) from the OR of its chunks' words: column bits in 7:0, bit 13 = some coefficient in rows 4-7
JDA_HD uint32_t jda_p1c_fold(uint32_t fl) { return ((fl >> 8) & 0xffu) | ((fl & 0x40u) << 7); }

// Does some lane of the wavefront (of those that are here) say yes?  The host emulator steps the lanes one after another:
// there a lane answers for itself (the exact decoder is right for every block, the short one for every unflagged block).
JDA_HD bool jda_wave_any(bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v) != 0ull;
#else
    return v;
#endif
}

// Multiplication by an IDCT constant.  FAST: both operands are known to fit in 24 signed bits (the
// host checks max|coef| * max|quant| < 2^21 for the image, see jda_frontend.cpp), so the full-rate
// 24-bit multiplier gives the exact low 32 bits; otherwise the full 32-bit multiply (quarter rate).
template <bool FAST> JDA_HD int32_t jda_mulc(int32_t x, int32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) return __mul24(x, c);     // v_mul_i32_i24 (hip_runtime.h is included by the .hip file)
    return x * c;
#else
    if (FAST) return (int32_t)((uint32_t)(((int32_t)((uint32_t)x << 8)) >> 8) * (uint32_t)c);   // emulate the 24-bit operand
    return (int32_t)((uint32_t)x * (uint32_t)c);
#endif
}

// ---- dequant + IDCT + range limit (jpeg.inl:2553-2797) ---------------------------------------
// One 1-D pass of the reference's row stage on eight int32 inputs, block-level variant chosen by
// the occupancy flags (jpeg.inl:2686-2743).  Returns the eight range-limited bytes packed LE.
struct jda_row8 { uint32_t lo, hi; };

// RC: 0 = only columns 0-1 occupied, 1 = columns 0-3, 2 = any (the block-level tests of :2686-2688)
template <int RC>
JDA_HD jda_row8 jda_idct_row(const int32_t s[8])
{
    // The even part's four terms are only ever needed as the packed pairs (t0, t1) and (t3, t2) of the final butterflies, and only
    // modulo 2^16 (below): they are made there -- pair + pair and pair - pair -- instead of four 32-bit sums and two packs.
    int32_t t4, t5, t6, t7;
    uint32_t p01, p32;
    if (RC == 0) {                                                   // :2688-2697
        p01 = p32 = jda_dup16(s[0]);
        t7 = s[1];
        t6 = (t7 * 217) >> 8;
        t5 = (t7 * 145) >> 8;
        t4 = -((t7 * 51) >> 8);
    } else if (RC == 1) {                                            // :2698-2718
        const int32_t c = s[2];
        const int32_t m = (c * 106) >> 8;
        const uint32_t pa = jda_dup16(s[0]), pc = jda_pack16(c, m);
        p01 = jda_pk_add16(pa, pc);                                  // t0 = a + c, t1 = a + m
        p32 = jda_pk_sub16(pa, pc);                                  // t3 = a - c, t2 = a - m
        const int32_t z13 = s[3], z11 = s[1];
        t7 = z11 + z13;
        const int32_t t11 = ((z11 - z13) * 362) >> 8;
        const int32_t z5 = ((z11 - z13) * 473) >> 8;
        const int32_t t10 = ((z11 * 277) >> 8) - z5;
        const int32_t t12 = ((z13 * 669) >> 8) + z5;
        t6 = t12 - t7; t5 = t11 - t6; t4 = t10 + t5;
    } else {                                                         // :2720-2743
        const int32_t t10 = s[0] + s[4], t11 = s[0] - s[4];
        const int32_t t13 = s[2] + s[6];
        const int32_t t12 = (((s[2] - s[6]) * 362) >> 8) - t13;
        const uint32_t pe = jda_pack16(t10, t11), pf = jda_pack16(t13, t12);
        p01 = jda_pk_add16(pe, pf);                                  // t0 = t10 + t13, t1 = t11 + t12
        p32 = jda_pk_sub16(pe, pf);                                  // t3 = t10 - t13, t2 = t11 - t12
        const int32_t z13 = s[5] + s[3], z10 = s[5] - s[3];
        const int32_t z11 = s[1] + s[7], z12 = s[1] - s[7];
        t7 = z11 + z13;
        const int32_t u11 = ((z11 - z13) * 362) >> 8;
        const int32_t z5 = ((z10 + z12) * 473) >> 8;
        const int32_t u10 = ((z12 * 277) >> 8) - z5;
        const int32_t u12 = ((z10 * -669) >> 8) + z5;
        t6 = u12 - t7; t5 = u11 - t6; t4 = u10 + t5;
    }
    // :2745-2793  the eight outputs t_a +- t_b and ucRangeTable[(v >> 5) & 0x3ff], two per instruction.
    // Only bits 14:5 of an output reach the table index, so the final adds can be done modulo 2^16 on
    // packed pairs: (o0,o1) = (t0,t1) + (t7,t6), (o7,o6) = (t0,t1) - (t7,t6), (o3,o2) = (t3,t2) + (-t4,t5),
    // (o4,o5) = (t3,t2) - (-t4,t5).  Then the 10-bit sign-extended field (the table's wrap: << 1, >> 6
    // arithmetic on 16-bit lanes), + 128, saturate to a byte.
    const uint32_t q76 = jda_pack16(t7, t6), q45 = jda_pack16(-t4, t5);
    const uint32_t s01 = jda_pk_limit10(jda_pk_add16(p01, q76));      // samples in bytes 1 and 3
    const uint32_t s76 = jda_pk_limit10(jda_pk_sub16(p01, q76));
    const uint32_t s32 = jda_pk_limit10(jda_pk_add16(p32, q45));
    const uint32_t s45 = jda_pk_limit10(jda_pk_sub16(p32, q45));
    jda_row8 r;
    r.lo = jda_perm(s32, s01, 0x05070301u);          // bytes o0 o1 o2 o3  (s32 = [o3,o2])
    r.hi = jda_perm(s76, s45, 0x05070301u);          // bytes o4 o5 o6 o7  (s45 = [o4,o5], s76 = [o7,o6])
    return r;
}

// Column stage for one column (jpeg.inl:2561-2676): c[r] = raw coefficient of row r, q[r] its
// prescaled quantiser; results truncated to int16 as the reference stores them back.
template <bool FAST, bool HALF>
JDA_HD void jda_idct_col(const int32_t c[8], const int32_t q[8], int32_t out[8])
{
    int32_t t0, t1, t2, t3, t4, t5, t6, t7;
    if (HALF) {                                              // :2561-2601
        const int32_t a = c[0] * q[0];
        const int32_t b = c[2] * q[2];
        const int32_t m = jda_mulc<FAST>(b, 106) >> 8;
        t0 = a + b; t3 = a - b; t1 = a + m; t2 = a - m;
        t4 = c[1] * q[1];
        if (c[3] != 0) {
            const int32_t d = c[3] * q[3];
            t7 = t4 + d;
            const int32_t t11 = jda_mulc<FAST>(t4 - d, 362) >> 8;
            const int32_t z5 = jda_mulc<FAST>(t4 - d, 473) >> 8;
            const int32_t t12 = (jda_mulc<FAST>(d, 669) >> 8) + z5;               // (-tmp5 * -669) >> 8
            t6 = t12 - t7;
            t5 = t11 - t6;
            const int32_t t10 = (jda_mulc<FAST>(t4, 277) >> 8) - z5;
            t4 = t10 + t5;
        } else {
            t7 = t4;
            t5 = jda_mulc<FAST>(t4, 145) >> 8;
            t6 = jda_mulc<FAST>(t4, 217) >> 8;
            t4 = jda_mulc<FAST>(t4, -51) >> 8;
        }
    } else {                                                         // :2602-2676
        // the reference's zero tests on rows 4..7 only skip work; the arithmetic is identical
        const int32_t e0 = c[0] * q[0], e4 = c[4] * q[4];
        const int32_t t10 = e0 + e4, t11 = e0 - e4;
        const int32_t e2 = c[2] * q[2], e6 = c[6] * q[6];
        const int32_t t13 = e2 + e6;
        const int32_t t12 = (jda_mulc<FAST>(e2 - e6, 362) >> 8) - t13;
        t0 = t10 + t13; t3 = t10 - t13; t1 = t11 + t12; t2 = t11 - t12;
        const int32_t o3 = c[3] * q[3], o5 = c[5] * q[5];
        const int32_t z13 = o5 + o3, z10 = o5 - o3;
        const int32_t o1 = c[1] * q[1], o7 = c[7] * q[7];
        const int32_t z11 = o1 + o7, z12 = o1 - o7;
        t7 = z11 + z13;
        const int32_t u11 = jda_mulc<FAST>(z11 - z13, 362) >> 8;
        const int32_t z5 = jda_mulc<FAST>(z10 + z12, 473) >> 8;
        const int32_t u12 = (jda_mulc<FAST>(z10, -669) >> 8) + z5;
        t6 = u12 - t7;
        t5 = u11 - t6;
        const int32_t u10 = (jda_mulc<FAST>(z12, 277) >> 8) - z5;
        t4 = u10 + t5;
    }
    out[0] = (int16_t)(t0 + t7); out[1] = (int16_t)(t1 + t6);
    out[2] = (int16_t)(t2 + t5); out[3] = (int16_t)(t3 - t4);
    out[4] = (int16_t)(t3 + t4); out[5] = (int16_t)(t2 - t5);
    out[6] = (int16_t)(t1 - t6); out[7] = (int16_t)(t0 - t7);
}

// 1/4 scale: 2x2 block from coefficients 0,1,8,9 (jpeg.inl:2305-2326) -> 4 bytes
JDA_HD uint32_t jda_idct_2x2(const int16_t *coef, const int16_t *quant)
{
    const int32_t a = coef[0] * quant[0], b = coef[8] * quant[8];
    const int32_t c = coef[1] * quant[1], d = coef[9] * quant[9];
    const int32_t t0 = a + b, t2 = a - b, t1 = c + d, t3 = c - d;
    return jda_range_limit5(t0 + t1) | (jda_range_limit5(t0 - t1) << 8) |
           (jda_range_limit5(t2 + t3) << 16) | (jda_range_limit5(t2 - t3) << 24);
}

// ---- colour conversion (jpeg.inl:3101-3278) -------------------------------------------------
struct jda_ycc { int32_t y; int32_t cb; int32_t cr; };   // y pre-scaled by 2^12 (or sum<<10)

JDA_HD uint32_t jda_pixel_rgba(jda_ycc p)
{
    const int32_t cb = p.cb - 128, cr = p.cr - 128;
    const int32_t r = jda_clamp255((5742 * cr + p.y) >> 12);
    int32_t g = jda_clamp255((-1409 * cb - 2925 * cr + p.y) >> 12);
    const int32_t b = jda_clamp255((7258 * cb + p.y) >> 12);
    JDA_OPAQUE(g);
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | 0xff000000u;
}

JDA_HD uint32_t jda_pixel_565(jda_ycc p, bool big_endian)
{
    const int32_t cb = p.cb - 128, cr = p.cr - 128;
    // usRangeTableR/G/B[(x >> 12) & 0x3ff]: 10-bit wrap, then clamp
    const int32_t r = jda_clamp255(jda_sext10_at(5742 * cr + p.y, 12));
    const int32_t g = jda_clamp255(jda_sext10_at(-1409 * cb - 2925 * cr + p.y, 12));
    const int32_t b = jda_clamp255(jda_sext10_at(7258 * cb + p.y, 12));
    uint32_t v = (uint32_t)((r >> 3) << 11) | (uint32_t)((g >> 2) << 5) | (uint32_t)(b >> 3);
    if (big_endian) v = ((v & 0xffu) << 8) | (v >> 8);
    return v;
}

JDA_HD uint32_t jda_gray_565(uint32_t y, bool big_endian)
{
    uint32_t v = ((y >> 3) << 11) | ((y >> 2) << 5) | (y >> 3);     // usGrayTo565
    if (big_endian) v = ((v & 0xffu) << 8) | (v >> 8);
    return v;
}

// Sample fetch for output pixel (px,py) of one MCU's output tile, per subsampling mode and scale
// shift.  planes: the MCU's blocks, 64 bytes each, block order Y.. Cb Cr.  `luma` = only the
// luma value is wanted (GRAY8 output / gray JPEG): returned in .y UNscaled (0..255).
template <int MODE>
JDA_HD jda_ycc jda_fetch(const uint8_t *planes, uint32_t px, uint32_t py, int shift, bool luma)
{
    typedef jda_mode_traits<MODE> T;
    jda_ycc o;
    o.cb = 128; o.cr = 128;
    const uint8_t *Y = planes;
    uint32_t cidx = 0;                                  // index into the 8x8 chroma blocks
    if (MODE == JDA_MODE_420) {
        // which luma block, and the position inside it
        const uint32_t bsz = 8u >> shift;               // block edge in output pixels: 8,4,2,1
        const uint32_t q = (py / bsz) * 2 + (px / bsz);
        Y = planes + JDA_COEF_STRIDE * q;
        const uint32_t bx = px & (bsz - 1), by = py & (bsz - 1);
        if (shift == 0) { o.y = Y[by * 8 + bx]; cidx = (py >> 1) * 8 + (px >> 1); }         // :4333-4543
        else if (shift == 1) {                                                              // :3577-3626
            const uint8_t *s = Y + by * 16 + bx * 2;
            o.y = s[0] + s[1] + s[8] + s[9];
            cidx = py * 8 + px;
        } else if (shift == 2) { o.y = Y[by * 2 + bx]; cidx = q; }                          // :3664-3748
        else { o.y = Y[0]; cidx = 0; }                                                      // :3627-3663
        if (!luma) {
            o.cb = planes[JDA_COEF_STRIDE * 4 + cidx];
            o.cr = planes[JDA_COEF_STRIDE * 5 + cidx];
            o.y = (shift == 1) ? (o.y << 10) : (o.y << 12);
        } else if (shift == 1) o.y = (o.y + 2) >> 2;                                        // :2979-2999
    } else if (MODE == JDA_MODE_422 || MODE == JDA_MODE_440) {
        // two luma blocks side by side (4:2:2, JPEGPutMCU21 jpeg.inl:4721-4868) or one above the other (4:4:0,
        // JPEGPutMCU12 :4546-4719); chroma blocks in slots 2, 3.  a = coordinate along which the blocks are stacked
        // and the chroma is subsampled, c = the other one.
        const bool horiz = MODE == JDA_MODE_422;
        const uint32_t bsz = 8u >> shift;
        const uint32_t a = horiz ? px : py, c = horiz ? py : px;
        const uint32_t q = a / bsz, ba = a & (bsz - 1);
        Y = planes + JDA_COEF_STRIDE * q;
        const uint32_t bx = horiz ? ba : c, by = horiz ? c : ba;          // position inside the luma block
        const uint8_t *Cb = planes + 2 * JDA_COEF_STRIDE, *Cr = planes + 3 * JDA_COEF_STRIDE;
        uint32_t cb = 128, cr = 128;
        if (shift == 0) {
            o.y = Y[by * 8 + bx];
            cidx = horiz ? (py * 8 + q * 4 + (bx >> 1)) : ((py >> 1) * 8 + px);
            cb = Cb[cidx]; cr = Cr[cidx];
        } else if (shift == 1) {
            const uint8_t *s = Y + by * 16 + bx * 2;
            o.y = s[0] + s[1] + s[8] + s[9];
            // the chroma of the two source pixels across the non-subsampled direction is averaged, + 1 (:4743-4744, :4567-4568)
            cidx = horiz ? (py * 16 + q * 4 + bx) : (py * 8 + px * 2);
            const uint32_t step = horiz ? 8u : 1u;
            if (!luma) { cb = (Cb[cidx] + Cb[cidx + step] + 1u) >> 1; cr = (Cr[cidx] + Cr[cidx + step] + 1u) >> 1; }
        } else if (shift == 2) {
            o.y = Y[by * 2 + bx];
            cidx = horiz ? (py * 2 + q) : (q * 2 + px);                   // :4789-4838, :4610-4682
            cb = Cb[cidx]; cr = Cr[cidx];
        } else { o.y = Y[0]; cb = Cb[0]; cr = Cr[0]; }
        if (!luma) { o.cb = (int32_t)cb; o.cr = (int32_t)cr; o.y = (shift == 1) ? (o.y << 10) : (o.y << 12); }
        else if (shift == 1) o.y = (o.y + 2) >> 2;                         // JPEGPutMCU8BitGray :2857-2873, :2907-2923
    } else {
        if (shift == 0) { cidx = py * 8 + px; o.y = Y[cidx]; }
        else if (shift == 1) {
            cidx = py * 16 + px * 2;
            o.y = Y[cidx] + Y[cidx + 1] + Y[cidx + 8] + Y[cidx + 9];
        } else if (shift == 2) { cidx = py * 2 + px; o.y = Y[cidx]; }
        else { cidx = 0; o.y = Y[0]; }
        if (MODE == JDA_MODE_444 && !luma) {
            const uint8_t *Cb = planes + JDA_COEF_STRIDE, *Cr = planes + 2 * JDA_COEF_STRIDE;
            if (shift == 1) {                                                               // :3297-3322
                o.cb = (Cb[cidx] + Cb[cidx + 1] + Cb[cidx + 8] + Cb[cidx + 9] + 2) >> 2;
                o.cr = (Cr[cidx] + Cr[cidx + 1] + Cr[cidx + 8] + Cr[cidx + 9] + 2) >> 2;
                o.y <<= 10;
            } else { o.cb = Cb[cidx]; o.cr = Cr[cidx]; o.y <<= 12; }
        } else if (shift == 1) o.y = (o.y + 2) >> 2;                                        // :2812-2827, 3043-3069
        (void)T::NBLK;
    }
    return o;
}

// value of one output pixel in its final format
template <int MODE>
JDA_HD uint32_t jda_output_pixel(const uint8_t *planes, uint32_t px, uint32_t py, int shift, int pixel_type)
{
    if (pixel_type == JDA_EIGHT_BIT_GRAYSCALE) return (uint32_t)jda_fetch<MODE>(planes, px, py, shift, true).y;
    if (MODE == JDA_MODE_GRAY)                     // gray JPEG -> RGB565 (JPEGPutMCUGray, :3037-3099)
        return jda_gray_565((uint32_t)jda_fetch<MODE>(planes, px, py, shift, true).y, pixel_type != JDA_RGB565_LITTLE_ENDIAN);
    const jda_ycc p = jda_fetch<MODE>(planes, px, py, shift, false);
    if (pixel_type == JDA_RGB8888) return jda_pixel_rgba(p);
    return jda_pixel_565(p, pixel_type == JDA_RGB565_BIG_ENDIAN);
}


// ---- device pre-scan ----------------------------------------------------------------
// The filtered scan is cut into segments of JDA_SEG_BYTES; one lane walks one segment's Huffman symbols in skip mode.
// A lane does not know the decoder state at its segment's first bit -- (bit offset of the next symbol, block within the
// MCU, zigzag position) -- so it guesses "a block starts here" and walks: Huffman streams re-synchronise by themselves
// after a few symbols, and an EOB re-synchronises the zigzag position.  The passes (jda_kernels.hip):
//   round 0   every lane walks its segment from the guess and hands the state at the segment's end to the next segment
//   round 1.. walk again from the entry state the segment before handed over, now counting (block starts, per-component DC sums, how
//             the reference's window phase propagates: six byte lags in flight); a walk whose exit differs from what the next segment
//             was entered with puts that segment on the next round's list.  Segment 0's entry state is known, so a round that
//             leaves its list empty is the fixed point E[i+1] = walk(i, E[i]) with E[0] true: the serial decoder's states
//   sums      exclusive scan over the segments: first block ordinal, DC predictors, window phase at every entry
//   WRITE     the walk once more, now writing the per-block index and DC predictors exactly as the serial host pre-scan does.
// State at a symbol boundary (after the reference's bottom-of-loop refill, before its next top-of-loop refill):
//   bits 5:0 bit offset from the segment's first bit (entry: how far the previous segment's last symbol reached in),
//   bits 8:6 block within the MCU, bits 14:9 zigzag position k (0 = the next symbol is a DC code).
#define JDA_SEG_BITS  (JDA_SEG_BYTES * 8u)
#define JDA_SEG_SLOT  (JDA_SEG_BYTES + 12u)           // bytes a lane reads: its segment + 12 bytes of the next; 67 dwords per lane in LDS (odd: no bank conflicts)
#define JDA_SEG_DEAD  0x7fffffffu    // state of a walk that met an invalid code (bit 31 is the rounds' "changed" mark)
#define JDA_SEG_CHANGED 0x80000000u // entry-state word: "differs from the previous round's" (the walker's own bits are 14:0)
enum { JDA_SEG_SPEC = 0 /* round 0: the exit state only */, JDA_SEG_RECORD = 4 /* + the segment's sums, one record per block start, truncation candidates (jda_segscan_finalize) */ };
#define JDA_SEG_SUM_WORDS 8u         // per segment: block starts, DC sums [3], phase map, bad | has-restart | max AC category << 4, lag word at the last block start, round
#define JDA_ST_NCAND 66u             // result word: truncation candidates appended (RECORD)
#define JDA_REC_POS_BITS 12u         // a record: bit position of the block's first AC symbol, counted from its segment's first bit (the block's DC symbol starts in the
                                     // segment: <= 2047 + 27) | running DC sum of its component, the block's own difference included, << 12
#define JDA_SEG_FIRST_RST_SHIFT 8u   // seg_sum word 5, bits 31:8: the ordinal of the segment's first block behind an interval end (its sums count from a restart)

struct jda_segscan_params {          // one per image
    const uint8_t *scan;             // filtered scan (global), zero padded to n_segs * JDA_SEG_BYTES + 16
    const uint8_t *tables;           // table blob (global)
    uint32_t *entry_cur, *entry_nxt; // n_segs + 1 entry states each; swapped between rounds
    uint32_t *seg_sum;               // COUNT out, 6 words per segment: block starts, DC sums [3], phase map, bad
    const uint32_t *seg_start;       // WRITE in, 5 words per segment: first block ordinal, DC predictors [3], byte lag j of the reference window
    uint32_t *blk_index;             // WRITE out: n_blocks_total + 1
    int16_t *blk_dc;                 // WRITE out: n_blocks_total
    uint32_t *stats;                 // [0] bad, [1] terminal entry written, [2] max AC category, [3] max |DC|, [4] truncated reads, [5] a restart marker is not where the MCU count puts it
    uint32_t scan_len, n_segs, n_blocks_total, first_round;
    uint8_t nluma, nblocks, dc_id[3], ac_id[3];
    const uint32_t *filter_result;   // device filter ran (jda_pipeline): [0] = the filtered length; scan_len / n_segs above are upper bounds then
    uint32_t *worklist;              // jda_segscan_fused: two lists of n_segs (upper bound) segment numbers -- who walks in the next round
    uint32_t worklist_cap;
    // streams with restart intervals (0 / NULL otherwise): where in the FILTERED scan every interval starts -- restart_pos[0] = 0,
    // restart_pos[1 .. n_intervals - 1] = the positions the RSTn markers stood at, restart_pos[n_intervals] = JDA_RST_SENTINEL --,
    // blocks per interval (DRI x blocks per MCU), and whether the last interval is a whole one (the closing index entry is then
    // rounded up to a byte like every interval end, jpeg.inl:5339-5346)
    const uint32_t *restart_pos;
    uint32_t n_intervals, interval_blocks, round_last;
    // the walk's four tables (JDA_WT_BYTES) as jda_walk_tables_build makes them from the blob, once per image: a walker's workgroup
    // copies them (one wait) instead of converting the blob itself
    uint8_t *walk_tables;
    uint32_t walk_tables_shared;      // 1: another image of the batch (same tables, same table ids) builds them
    // RECORD mode (NULL / 0: the counting walk + WRITE walk of round 2): the counting walk leaves one record per block start --
    // rec_cap slots per segment (more than its 2,048 bits can start blocks: jda_record_cap), 16-byte groups -- and the rare
    // magnitude read that SOME entry lag would truncate as a candidate (16 bytes: segment, ordinal + 1 | round << 16, lag word at
    // the block's start, the lags that truncate); jda_segscan_finalize turns records into index entries and predictors,
    // jda_segscan_resolve the candidates of the true lag into flagged entries
    uint32_t *records;
    uint32_t rec_cap;
    uint32_t *cands;
    uint32_t cand_cap;
    // RECORD mode with restart intervals: two words per interval start nr = 1 .. n_intervals - 1, zeroed -- the walk that ends the
    // interval in front of it leaves (its segment << 11 | blocks it had started by then + 1, its round): jda_segscan_resolve_cands
    // checks that every marker ended an interval of a settled walk exactly where the MCU count puts it (what WRITE checks as it goes)
    uint32_t *rst_events;
};
#define JDA_RST_SENTINEL 0x1fffffffu      // (a byte position no scan reaches: the index packs positions in 25 bits)
#define JDA_SEG_HAS_RESTART 2u           // seg_sum word 5, bit 1: an interval ends inside the segment (its DC sums count from there)
// the parameters with what only the device knows filled in
JDA_HD jda_segscan_params jda_segscan_resolve(const jda_segscan_params &in)
{
    jda_segscan_params P = in;
    if (P.filter_result) {
        P.scan_len = JDA_G(const uint32_t, P.filter_result)[0]; P.n_segs = P.scan_len / JDA_SEG_BYTES + 1u;
        // as many RSTn markers as the MCU count asks for?  If not, nothing walks (restart_pos is not what the walk takes it for) and
        // the result words stay "no index": the serial pre-scan does what the reference does with such a file
        if (P.restart_pos && JDA_G(const uint32_t, P.filter_result)[1] + 1u != P.n_intervals) P.n_segs = 0;
    }
    return P;
}
struct jda_seg_sum { uint32_t nblk; int32_t dcsum[3]; uint32_t phase_map, bad, lag_last, max_ac; };
struct jda_seg_stats { uint32_t bad, terminal, max_ac_bits, max_abs_dc, trunc_events, mismatch; };

JDA_HD uint32_t jda_atomic_inc_u32(uint32_t *p)                   // the value before
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return (*p)++;                                   // (the emulator steps the lanes one after another)
#endif
}
JDA_HD void jda_atomic_or_u32(uint32_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p |= v;                                         // (the emulator steps the lanes one after another)
#endif
}
// The walk's view of its segment: 64 stream bits in two registers and the three dwords after them (the segment is read where it
// lies -- global memory, L2 -- and nothing of the scan is staged in LDS: the tables are all a workgroup keeps there, and the
// wavefronts a CU holds are bounded by registers, not by 17 KB of slots each).  A symbol is at most 31 bits (code <= 16, magnitude
// <= 15), so a step advances the window by at most one dword -- a register shuffle, no load: the five dwords are (re)loaded for
// ALL lanes of the wavefront at once, from wherever each lane stands, every JDA_SEG_REFILL_STEPS steps (jda_seg_walk).  A load
// inside the step ("the lane that crosses a dword boundary fetches the next one") was a load + its wait in EVERY step of the
// wavefront: one of 64 unsynchronised lanes crosses every time.  Five dwords are at least 97 bits ahead of the lane: 12 bits per
// symbol over eight steps; the lane that runs out anyway (or, at an interval end, steps over a dword) reloads on its own (rare).
struct jda_seg_reader {
    const uint32_t JDA_GLOBAL *d;   // the segment's first dword (readable: JDA_SEG_SLOT bytes)
    uint32_t idx, base;             // dword index of hi now / when the registers were loaded
    uint32_t hi, lo;                // stream bits, first byte on top
    uint32_t n1, n2, n3;            // the dwords behind lo (first byte on top, like hi and lo); valid while idx - base <= 3
};
#ifndef JDA_SEG_REFILL_STEPS
#define JDA_SEG_REFILL_STEPS 8u
#endif
JDA_HD void jda_seg_reader_init(jda_seg_reader &R, const uint32_t JDA_GLOBAL *d, uint32_t p)
{
    R.d = d; R.idx = R.base = p >> 5;
    R.hi = __builtin_bswap32(d[R.idx]); R.lo = __builtin_bswap32(d[R.idx + 1u]);
    R.n1 = __builtin_bswap32(d[R.idx + 2u]); R.n2 = __builtin_bswap32(d[R.idx + 3u]); R.n3 = __builtin_bswap32(d[R.idx + 4u]);      // (swapped here, once a reload: the slide is in every step)
}
// the next 32 bits of the stream at bit p of the segment
// (the window slides by selects, no branch: the walk's loop sees to it that p stays within the loaded dwords -- jda_seg_reader_holds --
// and reloads for the whole wavefront otherwise; a peek is at most one dword ahead of the last one)
JDA_HD uint32_t jda_seg_reader_peek(jda_seg_reader &R, uint32_t p)
{
    const uint32_t wi = p >> 5;
    const bool adv = wi != R.idx;
    R.hi = adv ? R.lo : R.hi;
    R.lo = adv ? R.n1 : R.lo;
    R.n1 = adv ? R.n2 : R.n1;
    R.n2 = adv ? R.n3 : R.n2;
    R.idx = wi;
    const uint64_t v = ((uint64_t)R.hi << 32) | R.lo;
    return (uint32_t)((v << (p & 31u)) >> 32);
}
JDA_HD bool jda_seg_reader_holds(const jda_seg_reader &R, uint32_t p) { return (p >> 5) - R.base <= 3u; }
#define JDA_SEG_READ_DWORDS (JDA_SEG_BYTES / 4u + 4u)     // dwords of the scan a walk may touch from its segment's start (64 + the window's reach)

// The walks' DC entries: the reference's DC LUT (jpeg.inl:1098-1152) re-laid out like the AC entries -- (code length - 1) << 12 |
// SSSS << 8 | folded -- so that one lookup serves a DC and an AC symbol alike.  "folded" (bit 0): the reference takes code
// and magnitude from the LUT in one step (:1132-1152) and does not refill between them.
//   JDA_WT_* (the segment walk's own LDS layout): FOUR tables of 2048 entries -- AC 0, AC 1, DC 0, DC 1 -- under the AC tables'
//   11-bit key (the stream's top 10 bits, or 1024 + the 10 bits behind six leading ones), so that a step computes ONE address:
//   table number from the block's place in the MCU, key from the stream.  A DC entry under that key is the reference's for every
//   stream that has those bits, provided no DC code 111110.. is longer than 10 bits (jda_dc_lut_walkable: the front end keeps
//   such a file -- none seen -- on the serial pre-scan).
//   An entry is 32 bits: the low half the symbol (the decode kernel's jda_ac_entry layout / jda_dc16_entry), the high half -- AC
//   tables, short keys -- the symbol BEHIND it where the key's ten bits hold that one's code too: two symbols a step (JDA_WT_PAIR_*;
//   the second is described by what a walk needs of it: bits it takes, coefficients it moves on by, its magnitude's size).
#define JDA_WT_TABLE_BYTES 8192u
#define JDA_WT_BYTES (4u * JDA_WT_TABLE_BYTES)
#define JDA_WT_PAIR_VALID 0x8000u                     // | size << 10 | (run + 1, 0 = EOB) << 5 | code + magnitude bits (<= 16)
// the pair half of the entry ea under `key` (< 1024): ac = the 2048 raw entries (length << 8 | RS, 0 = no code) of the AC table that
// decodes the symbol behind it -- the table itself for an AC entry, the table of the component for a DC entry (jda_wt_dc_follow)
JDA_HD uint32_t jda_wt_pair(uint32_t ea, const uint16_t JDA_GLOBAL *ac, uint32_t key)
{
    if (JDA_AC_STOPS(ea)) return 0u;                                // EOB ends the block, no code ends the walk's luck: no second symbol
    const uint32_t bits_a = (ea >> 12) + 1u + ((ea >> 8) & 15u);
    if (bits_a > 9u) return 0u;
    const uint32_t left = 10u - bits_a, key_b = (key << bits_a) & 1023u;      // the key's bits behind A, zeros behind them
    if ((key_b >> 4) == 63u) return 0u;                             // six ones: a code of the table's long half
    const uint32_t eb = jda_ac_entry(ac[key_b]);
    if ((eb & 0xffu) == JDA_AC_NONE) return 0u;
    const uint32_t len_b = (eb >> 12) + 1u;
    if (len_b > left) return 0u;                                    // B's code is not all there (a longer code matched the zeros)
    const bool eob = (eb & 0xffu) == JDA_AC_EOB;
    const uint32_t sz_b = eob ? 0u : (eb >> 8) & 15u, run_b = (eb >> 1) & 15u;
    if (len_b + sz_b > 16u) return 0u;                              // (a magnitude read the reference may truncate is a step of its own)
    return JDA_WT_PAIR_VALID | (sz_b << 10) | ((eob ? 0u : run_b + 1u) << 5) | (len_b + sz_b);
}
JDA_HD uint32_t jda_dc16_entry(uint32_t e8, int32_t folded)
{
    if (e8 == 0u) return JDA_AC_NONE;
    const uint32_t s = e8 & 15u, tot = e8 >> 4;
    const bool fold = s != 0u && folded != 0;
    const uint32_t len = fold ? tot - s : tot;                   // a folded entry holds code + magnitude length
    return ((len - 1u) << 12) | (s << 8) | (fold ? 1u : 0u);
}

// which AC table decodes the symbol behind a DC symbol of DC table t (2 bits per t; 2: none -- the components that share the DC
// table do not share an AC table): the DC entries take their pair halves from it
JDA_HD uint32_t jda_wt_dc_follow(const jda_segscan_params &P)
{
    const uint32_t ncomp = P.nblocks > P.nluma ? 3u : 1u;
    uint32_t f = 0;
    for (uint32_t t = 0; t < 2u; t++) {
        uint32_t seen = 3u;                                         // 3: no component uses DC table t
        for (uint32_t c = 0; c < ncomp; c++) {
            if ((P.dc_id[c] & 1u) != t) continue;
            const uint32_t a = P.ac_id[c] & 1u;
            seen = seen == 3u ? a : (seen == a ? a : 2u);
        }
        f |= (seen == 3u ? 2u : seen) << (2u * t);
    }
    return f;
}
// the segment walk's tables, staged by its workgroup (tid = thread in workgroup): wt holds JDA_WT_BYTES
JDA_HD void jda_walk_tables_from(const uint8_t *tables, uint32_t follow, uint32_t tid, uint32_t nthreads, uint8_t *wt)
{
    const uint16_t JDA_GLOBAL *ac = JDA_G(const uint16_t, tables + JDA_TB_AC);
    uint32_t *out = (uint32_t *)wt;
    for (uint32_t j = tid; j < 4096u; j += nthreads) {                              // both halves of both AC tables -> the kernels' entries
        const uint32_t t = j >> 11, key = j & 2047u;
        const uint32_t ea = jda_ac_entry(ac[j]);
        out[j] = ea | (key < 1024u ? jda_wt_pair(ea, ac + t * 2048u, key) << 16 : 0u);
    }
    for (uint32_t j = tid; j < 4096u; j += nthreads) {                              // the DC tables under the same key
        const uint32_t t = j >> 11, idx = jda_dc_lut_index(jda_walk_key_code12(j & 2047u, 0u));
        const uint8_t JDA_GLOBAL *dc = JDA_G(const uint8_t, tables) + JDA_TB_DC + t * 1024u;
        const uint32_t ea = jda_dc16_entry(dc[idx], (int8_t)dc[idx + 512u]), key = j & 2047u, fa = (follow >> (2u * t)) & 3u;
        out[4096u + j] = ea | (key < 1024u && fa < 2u ? jda_wt_pair(ea, ac + fa * 2048u, key) << 16 : 0u);
    }
}
// what a walker's workgroup does: a copy of the image's prepared tables (jda_walk_tables_build; 32 KB, one wait)
JDA_HD void jda_walk_tables_stage(const uint8_t *prepared, uint32_t tid, uint32_t nthreads, uint8_t *wt)
{
    const jda_chunk16_alias JDA_GLOBAL *src = JDA_G(const jda_chunk16_alias, prepared);
    jda_chunk16_alias *out = (jda_chunk16_alias *)wt;
    for (uint32_t i = tid; i < JDA_WT_BYTES / 16u; i += nthreads) out[i] = src[i];
}
// One walk of a segment.  wt: the walk's tables (jda_walk_tables_from); segw: the segment's first dword in the (zero-padded)
// filtered scan.
//
// A step decodes a symbol (and the one behind it where the table entry describes it: JDA_WT_PAIR_*), DC or AC alike, without a branch on which it is: the lanes of a wavefront sit at unrelated places
// of their blocks, so "if DC .. else AC .." ran both sides every step -- and every small `if` in the body costs an exec-mask
// region (s_and_saveexec / s_cbranch / s_or), so the body is written with selects; what stays a branch is what is rare (a
// truncated magnitude, an invalid code) or has to store (a block's start and end in the WRITE pass).  The reference's refills
// (jpeg.inl:2110-2114) happen at fixed places -- before a block's first symbol, before a DC magnitude that is not folded into
// its LUT entry, at the top and the bottom of the AC loop -- and two of them in a row at the same bit are one; so a step is:
// [refill before an unfolded DC magnitude] .. refill at its end (bottom of the AC loop / top of it after the DC symbol / opening
// refill of the next block after EOB: same position, same result).  State at a step boundary, hence at a segment boundary:
// after that refill.
//   SPEC   exit state only.
//   COUNT  + block starts, DC sums per component, and how the reference window's BYTE LAG u = (p >> 3) - pBuf propagates: the
//          six lags a boundary state can have (a refill leaves off <= 47) ride in 5-bit fields of one word (4 value bits + a
//          guard bit for the compare); bits consumed advance all of them by the same number of bytes, a refill resets those
//          that reached 6 bytes.  phase_map: field j (3 bits) = exit lag for entry lag j.
//   WRITE  the reader itself (pBuf, ulBitOff), block ordinals and DC predictors from seg_start: index entries (held back to the
//          block's end so that a truncation flag joins its entry in the register; only a block that crosses into the next segment
//          is ORed in atomically), blk_dc -- both collected in registers and stored as aligned 16-byte groups (below) --,
//          maxima, truncation count.
// RST: the stream has restart intervals.  The filter recorded where every interval starts (byte aligned: the rest of the byte in
// front is padding); a walk that finishes an MCU within 7 bits of the next start has finished the interval: it steps over the
// padding, the reference rounds ulBitOff up WITHOUT a refill (jpeg.inl:5339-5346; so the refill after an interval's closing EOB
// waits for the next block's opening one), the DC predictors restart at zero.  The reference itself counts MCUs and never
// looks at marker positions: the WRITE pass checks that the two agree (every interval end at a multiple of interval_blocks,
// every such multiple an interval end) and sends the image to the serial pre-scan when they do not.
JDA_HD void jda_store_u32x4(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)      // one 16-byte store (p: global memory, 16-byte aligned)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *JDA_G(uint4, p) = make_uint4(a, b, c, d);
#else
    uint32_t *o = (uint32_t *)p; o[0] = a; o[1] = b; o[2] = c; o[3] = d;
#endif
}
// the six 5-bit byte lags of a counting walk + n each (n <= 3): U + n * 0x02108421 on the 24-bit multiplier and a shift-add
// (the 26-bit constant makes the product a quarter-rate v_mad_u64_u32, twice per symbol)
JDA_HD uint32_t jda_lag_add(uint32_t U, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;                                                     // (written out: the compiler puts any product + sum it recognises back on v_mad_u64_u32)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(n), "s"(0x00108421u), "v"(U));
    return r + (n << 25);
#else
    return U + n * 0x02108421u;
#endif
}
template <int OP, bool RST = false>
JDA_HD uint32_t jda_seg_walk(const jda_segscan_params &P, uint32_t seg, uint32_t entry, const uint32_t JDA_GLOBAL *segw, const uint8_t *wt,
                             jda_seg_sum &S, jda_seg_stats &ST, uint32_t round = 0)
{
    const bool REC = OP == JDA_SEG_RECORD;                            // sums + one record per block start + truncation candidates (else: the exit state only)
#ifndef JDA_SEG_PAIR_OFF
#define JDA_SEG_PAIR_OFF() false                                             // (host simulator: a switch, to count what the pairs save)
#endif
    const bool PAIR = !JDA_SEG_PAIR_OFF();                           // two AC symbols a step where the table holds the second (JDA_WT_PAIR_*)
    (void)ST;
    S.nblk = 0; S.dcsum[0] = S.dcsum[1] = S.dcsum[2] = 0; S.phase_map = 0; S.bad = 0; S.lag_last = 0; S.max_ac = 0;
    if (entry == JDA_SEG_DEAD) { S.bad = 1; return JDA_SEG_DEAD; }
    uint32_t p = entry & 63u, b2 = ((entry >> 6) & 7u) * 2u, k = (entry >> 9) & 63u;      // b2: twice the block's place in the MCU
    // per place in the MCU, in 2-bit fields (uniform): the block's component, and which of the walk's four tables decodes its
    // AC symbols (0 / 1) and its DC symbol (2 / 3)
    uint32_t csel = 0, acsel = 0, dcsel = 0;
    const uint32_t nluma = P.nluma, nblocks2 = 2u * P.nblocks;
    const uint32_t aid0 = P.ac_id[0] & 1u, aid1 = P.ac_id[1] & 1u, aid2 = P.ac_id[2] & 1u, did0 = P.dc_id[0] & 1u, did1 = P.dc_id[1] & 1u, did2 = P.dc_id[2] & 1u;
#pragma unroll
    for (uint32_t i = 0; i < 6u; i++) {                             // (constant indices only: a dynamic one would move P to scratch memory)
        const uint32_t ci = i < nluma ? 0u : i - nluma + 1u;
        const uint32_t a = ci == 0u ? aid0 : (ci == 1u ? aid1 : aid2), d = ci == 0u ? did0 : (ci == 1u ? did1 : did2);
        if (i < P.nblocks && ci < 3u) { csel |= ci << (2u * i); acsel |= a << (2u * i); dcsel |= (2u + d) << (2u * i); }
    }
    uint32_t U = 0;                                                 // the six byte lags
    const uint32_t kOnes = 0x02108421u, kGuard = 0x21084210u;       // 1 / 16 in each 5-bit field
    uint32_t nblk = 0;
    bool sbad = false;                                              // (a flag per lane: the compiler keeps it as a lane mask on the scalar unit)
    int32_t ds0 = 0, ds1 = 0, ds2 = 0;
    uint32_t max_ac = 0;
    if (REC) U = 0u | (1u << 5) | (2u << 10) | (3u << 15) | (4u << 20) | (5u << 25);
    // RECORD: the last four records (newest last; a group of four is stored when it is complete: the segment's slots are its own),
    // the lag word at the first AC symbol of the block in progress
    uint32_t rb0 = 0, rb1 = 0, rb2 = 0, rb3 = 0, Ublk = 0;
    uint32_t JDA_GLOBAL *recs = REC ? JDA_G(uint32_t, P.records) + (size_t)seg * P.rec_cap : (uint32_t JDA_GLOBAL *)0;
    // RST: the next interval start ahead of the walk (as a bit position relative to the segment); the ordinal of the segment's first
    // block behind an interval end (its DC sums count from there)
    const uint32_t JDA_GLOBAL *rpos = JDA_G(const uint32_t, P.restart_pos);
    const uint32_t seg_bit0 = seg * JDA_SEG_BITS;
    uint32_t nr = 0, next_bit = 0xffffffffu, has_rst = 0, first_rst = 0;
    if (RST) {
        const uint32_t byte0 = (seg_bit0 + p) >> 3;                 // smallest nr >= 1 with restart_pos[nr] * 8 > the entry position
        uint32_t lo = 1, hi = P.n_intervals;                        // (restart_pos[n_intervals] is the sentinel)
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rpos[mid] > byte0) hi = mid; else lo = mid + 1; }
        nr = lo;
        next_bit = (rpos[nr] << 3) - seg_bit0;
    }
    jda_seg_reader R;
    bool go = p < JDA_SEG_BITS;
    while (go) {
    jda_seg_reader_init(R, segw, p);                                // every lane of the wavefront from where it stands: one wait for all of them
    uint32_t since = 0;
    bool jumped = false;                                            // RST: this step went over an interval's padding
    do {
#ifdef JDA_SEG_STEP_HOOK
        JDA_SEG_STEP_HOOK();                                        // (host simulator: counts the steps of a walk)
#endif
        if (RST && p >= next_bit) {                                 // (a walk off the decoder's path ran over an interval start: catch up)
            nr++; next_bit = (rpos[nr] << 3) - seg_bit0;
            if (nr > P.n_intervals) { nr = P.n_intervals; next_bit = 0xffffffffu; }
        }
        const bool isdc0 = k == 0;
        const uint32_t c = jda_bfe(csel, b2, 2u);
        const uint32_t w = jda_seg_reader_peek(R, p);
        // one lookup, one address: the table by the block's place in the MCU and DC / AC, the entry by the 11-bit key (the
        // stream's top 10 bits, or 1024 + the 10 bits behind six leading ones)
        const uint32_t tsel = jda_bfe(isdc0 ? dcsel : acsel, b2, 2u);
        const uint32_t key = jda_bfe(w, w >= 0xfc000000u ? 16u : 22u, 11u);
        const uint32_t e32 = *(const jda_u32_alias *)(wt + (tsel << 13) + (key << 2));      // (sums: two shift-adds onto the tables' address)
        const uint32_t e = e32 & 0xffffu;
        const uint32_t elow = e & 0xffu;
        // no such code (:2137-2138, :2237-2238).  A walk that is not on the decoder's path yet may meet anything: it steps
        // on one bit and keeps looking (a walk that gave up would hand "dead" down the chain of segments, one per round) -- a
        // one-bit symbol without effects, by selects; its sums are void from there on (sbad).
        const bool inval = elow == JDA_AC_NONE;
        sbad |= inval;
        const bool live = !inval;                                   // the step's effects count
        const bool isdc = isdc0 & live;
        const bool eob = (elow == JDA_AC_EOB) & !inval;
        const uint32_t len = inval ? 1u : (e >> 12) + 1u, sz = (eob | inval) ? 0u : (e >> 8) & 15u;
        const uint32_t kk = k + ((e >> 1) & 15u);                   // (DC: + 0)
        const bool dcmag = isdc & (sz != 0u) & ((e & 1u) == 0u);    // a DC magnitude the reference refills for (not folded into the LUT entry)
        const bool acmag = !isdc0 & live & (sz != 0u) & (kk < 64u); // an AC magnitude that is stored
        const uint32_t p1 = p + len;
        bool counted = false;
        if (REC) {
            // a counted block starts here: its record is its DC VALUE so far as the segment knows it -- the running sum of its
            // component's differences (:2155-2165; a folded entry holds the same value), its own included -- and the place of its first AC symbol
            // a DC category no 8-bit baseline stream has does not fit the record's 20-bit sum: such a file keeps to the serial pre-scan
            sbad |= isdc & (sz > 11u);
            counted = isdc & !sbad;
            const int32_t diff = (counted & (sz != 0u)) ? jda_extend_top(w << len, sz) : 0;
            ds0 += c == 0 ? diff : 0; ds1 += c == 1 ? diff : 0; ds2 += c >= 2 ? diff : 0;
            const int32_t run = c == 0 ? ds0 : (c == 1 ? ds1 : ds2);
            const uint32_t rec = (p1 + sz) | ((uint32_t)run << JDA_REC_POS_BITS);          // (<= 2047 + 27: twelve bits)
            rb0 = counted ? rb1 : rb0; rb1 = counted ? rb2 : rb1; rb2 = counted ? rb3 : rb2; rb3 = counted ? rec : rb3;
            const bool group = counted & ((nblk & 3u) == 3u), fits = nblk < P.rec_cap;      // (one branch region, not two nested ones)
            if (group & fits) jda_store_u32x4(recs + (nblk - 3u), rb0, rb1, rb2, rb3);
            sbad |= group & !fits;                                  // (jda_record_cap leaves no room for this; memory stays ours anyway)
            nblk += counted ? 1u : 0u;
            U = jda_lag_add(U, ((p & 7u) + len) >> 3);                   // whole bytes the code bits advance the stream position by
            const uint32_t f1 = ((U | kGuard) - 6u * kOnes) & kGuard;
            // SURVEY fact 6 without knowing the entry lag: the reference reads the magnitude at ulBitOff = 8 u + (p1 & 7) and loses
            // bits when that + sz > 64 -- u >= 7 for (p1 & 7) + sz in 9..16, u >= 6 above 16 (u <= 7 here) -- so the six candidate
            // lags are tested at once; the rare hit is kept with the lag word at the block's first AC symbol (jda_segscan_resolve)
            // (ulBitOff <= 47 in front of the code for every candidate lag: code + magnitude must be 17 bits and more -- rare symbols)
            const uint32_t m = acmag ? sz : 0u;
            max_ac = m > max_ac ? m : max_ac;
            if (__builtin_expect(acmag && len + sz > 16u && !sbad, 0)) {
                const uint32_t t = (p1 & 7u) + sz;
                const uint32_t f7 = ((U | kGuard) - 7u * kOnes) & kGuard;
                const uint32_t hit = t > 16u ? f1 : (t > 8u ? f7 : 0u);
                if (hit != 0u) {
                    const uint32_t at = jda_atomic_inc_u32(P.stats + JDA_ST_NCAND);
                    if (at < P.cand_cap) jda_store_u32x4(P.cands + (size_t)at * 4u, seg, nblk | (round << 16), Ublk, hit);
                }
            }
            U &= dcmag ? ~(f1 - (f1 >> 4)) : 0xffffffffu;           // the refill before an unfolded DC magnitude
            U = jda_lag_add(U, ((p1 & 7u) + sz) >> 3);
        }
        p = p1 + sz;
        const bool ends = (eob | (kk + 1u >= 64u)) & !inval;
        const uint32_t bn = b2 + 2u == nblocks2 ? 0u : b2 + 2u;
        // RST: this symbol completes an MCU within 7 bits of the next interval's start = it completes the interval
        const bool iend = RST & ends & live & (bn == 0u) & (next_bit - p < 8u);
#ifdef JDA_SEG_TRACE_HOOK
        JDA_SEG_TRACE_HOOK(OP, seg, p, k, kk, b2, bn, ends, iend, next_bit, nr, nblk, e, inval);
#endif
        const bool hold = iend & eob;                               // the refill after an interval's closing EOB waits for the rounding
        // ---- the refill at the end of the step (not after EOB in the reference -- there it is the next block's opening one)
        if (REC) {
            const uint32_t f = ((U | kGuard) - 6u * kOnes) & kGuard;
            U &= hold ? 0xffffffffu : ~(f - (f >> 4));
            Ublk = counted ? U : Ublk;                              // the lags at the block's first AC symbol (behind the refill at the top of the AC loop)
        }
        k = (ends | inval) ? 0u : kk + 1u;
        b2 = ends ? bn : b2;
        if (PAIR) {
            // the symbol behind an AC symbol that leaves its block open, where the table knows it (jda_wt_pair) and the segment goes
            // on: what a step of its own would do to the walk's state -- an AC symbol moves p, k, the lags and the largest size
            const uint32_t pd = e32 >> 16;
            const uint32_t dk = (pd >> 5) & 31u, kend = kk + 1u + dk;
            const bool last_b = (dk == 0u) | (kend >= 64u);         // B would end its block
            // (RST: a symbol that completes an MCU may complete the interval -- a step of its own, for the test above)
            const bool pair = (pd != 0u) & live & !ends & (p < JDA_SEG_BITS) & !(RST & last_b & (bn == 0u));
            const uint32_t bits_b = pair ? pd & 31u : 0u;
            const bool ends_b = pair & last_b;
            if (REC) {
                const uint32_t m = (pair & (dk != 0u) & (kend <= 64u)) ? (pd >> 10) & 15u : 0u;      // (a stored magnitude: the coefficient's place is in the block)
                max_ac = m > max_ac ? m : max_ac;
                U = jda_lag_add(U, ((p & 7u) + bits_b) >> 3);       // (no pair: no bytes, and no lag is at 6 behind the refill above ..
                const uint32_t f = ((U | kGuard) - 6u * kOnes) & kGuard;
                U &= (RST && !pair) ? 0xffffffffu : ~(f - (f >> 4)); // .. but behind an interval's closing EOB, whose refill waits: RST)
            }
            p += bits_b;
            k = pair ? (ends_b ? 0u : kend) : k;
            b2 = ends_b ? bn : b2;
        }
        if (RST) {
            if (iend) {                                             // over the padding to the next interval's first byte
                const uint32_t frac = (p & 7u) ? 1u : 0u;           // (interval starts are byte aligned, so are segment starts)
                if (REC) {
                    U = jda_lag_add(U, frac);
                    const uint32_t f = ((U | kGuard) - 6u * kOnes) & kGuard;
                    U &= ~(f - (f >> 4));
                    ds0 = ds1 = ds2 = 0;
                    first_rst = has_rst ? first_rst : nblk; has_rst = JDA_SEG_HAS_RESTART;
                }
                if (REC && !sbad && nr < P.n_intervals) {      // who ended the interval in front of start nr, and after how many of its blocks
                    uint32_t JDA_GLOBAL *ev = JDA_G(uint32_t, P.rst_events) + 2u * (size_t)nr;
                    ev[0] = (seg << 11) | (nblk + 1u); ev[1] = round;
                }
                p = next_bit;
                jumped = true;
                nr++; next_bit = nr <= P.n_intervals ? (rpos[nr] << 3) - seg_bit0 : 0xffffffffu;
            }
        }
        go = p < JDA_SEG_BITS;
        since++;
        // (the next peek must find p within the loaded dwords and at most one dword on: a lane that has used them up, or jumped over an
        // interval's padding, sits out the rest of the wavefront's steps until the reload)
    } while (go & (since < JDA_SEG_REFILL_STEPS) & jda_seg_reader_holds(R, p) & !(RST && jumped));
    }
    if (REC) {                                                      // what is left of the last group, slot by slot
        const uint32_t r = nblk & 3u, n4 = nblk & ~3u;
        if (nblk <= P.rec_cap) {
            if (r > 0u) recs[n4 + r - 1u] = rb3;
            if (r > 1u) recs[n4 + r - 2u] = rb2;
            if (r > 2u) recs[n4 + r - 3u] = rb1;
        }
        S.lag_last = Ublk; S.max_ac = max_ac;
        uint32_t map = 0;
        for (int j = 0; j < 6; j++) map |= ((U >> (5 * j)) & 7u) << (3 * j);
        S.phase_map = map;
    }
    S.nblk = nblk; S.dcsum[0] = ds0; S.dcsum[1] = ds1; S.dcsum[2] = ds2;
    S.bad = (sbad ? 1u : 0u) | has_rst | (first_rst << JDA_SEG_FIRST_RST_SHIFT);
    return (p - JDA_SEG_BITS) | (b2 << 5) | (k << 9);
}

// ---- the marker filter's state machine on sixteen bytes at once (jda_filter_*, jda_kernels.hip) ------------------------------
// JPEGFilter (jpeg.inl:1431-1540) is a two-state machine: an FF in state 0 waits for its partner (state 1); the partner is dropped
// with it unless it is 00 (then FF is emitted); state 0 again.  After any byte that is not FF the state is 0, so the state in front
// of byte i is the parity of its distance from the start of the run of FFs that ends at i - 1: runs that start at an even position
// put state 1 in front of odd positions and vice versa, and adding the start bits of the even-started runs to the FF mask clears
// exactly those runs (the carry runs through a run and stops behind it).  cin: the state in front of byte 0 -- then byte 0 is a
// partner, and a run of FFs from byte 0 on counts as started at -1.
struct jda_filter_bits { uint32_t S, E, R; };       // state in front of byte i (bits 0..16), byte i is emitted / a restart marker's second byte (bits 0..15)
JDA_HD uint32_t jda_zero_byte_bits(uint32_t x)      // bit 7 of every byte that is zero
{
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}
// bits 7, 15, 23, 31 -> 0..3 (one dot product of bytes; the multiply-and-shift of the textbook is a quarter-rate v_mul_lo_u32)
JDA_HD uint32_t jda_byte_flags_to_nibble(uint32_t f) { return jda_udot4(f >> 7, 0x08040201u); }
struct jda_filter_masks { uint32_t ff, zero, rst; };  // per byte of the sixteen: is FF / is 00 / is D0..D7
JDA_HD jda_filter_masks jda_filter_classify(const uint32_t b[4])
{
    jda_filter_masks M;
    M.ff = M.zero = M.rst = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        M.ff |= jda_byte_flags_to_nibble(jda_zero_byte_bits(~b[d])) << (4 * d);
        M.zero |= jda_byte_flags_to_nibble(jda_zero_byte_bits(b[d])) << (4 * d);
        M.rst |= jda_byte_flags_to_nibble(jda_zero_byte_bits((b[d] & 0xf8f8f8f8u) ^ 0xd0d0d0d0u)) << (4 * d);
    }
    return M;
}
JDA_HD jda_filter_bits jda_filter_run(const jda_filter_masks &M, uint32_t valid, uint32_t cin)
{
    const uint32_t V = (1u << valid) - 1u;                           // valid <= 16
    const uint32_t starts = M.ff & ~(M.ff << 1);
    const uint32_t es = starts & 0x5555u & ~cin;                    // (cin: a run from byte 0 on is the tail of one that started at -1)
    const uint32_t Re = M.ff & ~(M.ff + es), Ro = M.ff & ~Re;
    jda_filter_bits F;
    F.S = ((Re << 1) & 0x0aaaau) | ((Ro << 1) & 0x15555u) | cin;
    F.E = V & ((~F.S & ~M.ff) | (F.S & M.zero));
    F.R = V & F.S & M.rst;
    return F;
}

// ---- from records to the index (block-parallel) ----------------------------------------------------------
// Index format 2: entry = the reader at the block's FIRST AC SYMBOL, blk_dc = the block's own DC value.  The entry of a block the
// reference reads without truncation is CANONICAL here: (p >> 3) << 7 | (p & 7) for the bit position p -- P1 needs p alone
// (pos * 8 + off, whatever the split); only a block flagged JDA_INDEX_TRUNC carries the reference reader's true (pBuf, ulBitOff)
// there, which its emulation starts from.  The serial pre-scan writes the true phase everywhere: the two indexes agree on p and
// on the flag of every block and on the whole entry of a flagged one.  The closing entry (behind the last block) bounds the scan
// from above: the serial pre-scan's is the reader as the last block left it, this one's lies behind the DC symbol that the stream's
// padding decodes to -- at most 34 bits further on (nothing reads it as a position: the last tile's window ends there).
JDA_HD uint32_t jda_index_canonical(uint32_t p_abs) { return ((p_abs >> 3) << JDA_INDEX_OFF_BITS) | (p_abs & 7u); }
struct jda_fin_acc { uint32_t bad, terminal, max_abs_dc; };
// record i of segment seg (first block ordinal g0, DC values pr0..2 of the components at its entry: jda_segscan_sums; rst_from: the
// segment's first record behind an interval end, 0xffffffff without one) -> index entry, DC value
// b0 = g0 % P.nblocks, inv = jda_fin_recip(P.nblocks): the block's place in the MCU without a division per record
JDA_HD uint32_t jda_fin_recip(uint32_t nblocks) { return (65536u + nblocks - 1u) / nblocks; }      // x / n = x * inv >> 16 while x * n < 65536
// (the record's load apart from its use: the kernel asks for the records of several segments before it waits for the first)
JDA_HD uint32_t jda_finalize_load(const jda_segscan_params &P, uint32_t seg, uint32_t i) { return JDA_G(const uint32_t, P.records)[(size_t)seg * P.rec_cap + i]; }
JDA_HD uint32_t jda_fin_rst_from(uint32_t sum5) { return (sum5 & JDA_SEG_HAS_RESTART) ? sum5 >> JDA_SEG_FIRST_RST_SHIFT : 0xffffffffu; }
JDA_HD void jda_finalize_apply(const jda_segscan_params &P, uint32_t seg, uint32_t i, uint32_t rec, uint32_t g0, uint32_t b0, uint32_t inv, int32_t pr0, int32_t pr1, int32_t pr2,
                               uint32_t rst_from, jda_fin_acc &A)
{
    const uint32_t g = g0 + i;
    if (g > P.n_blocks_total) return;                               // behind the image: padding decoded as blocks
    const uint32_t p_abs = seg * JDA_SEG_BITS + (rec & ((1u << JDA_REC_POS_BITS) - 1u));
    // a stream that ends early has been read on into its zero padding: the serial pre-scan knows what the reference does with it
    // (its test is on the reference's pBuf, at most five bytes behind: the margin makes this one the stricter)
    if ((p_abs >> 3) + 8u > P.scan_len + JDA_SCAN_PAD - 8u) A.bad = 1;
    if (g == P.n_blocks_total) {                                    // behind the last block: the closing entry (an upper bound, see above)
        JDA_G(uint32_t, P.blk_index)[g] = jda_index_canonical(P.restart_pos ? ((p_abs + 7u) & ~7u) : p_abs);
        A.terminal++;
        return;
    }
    const uint32_t x = b0 + i, b = x - jda_umul24(jda_umul24(x, inv) >> 16, P.nblocks), c = b < P.nluma ? 0u : b - P.nluma + 1u;      // (x < 6 + rec_cap)
    const int32_t base = i >= rst_from ? 0 : (c == 0u ? pr0 : (c == 1u ? pr1 : pr2));      // (DC values restart at zero with an interval)
    const int32_t dc = base + ((int32_t)rec >> JDA_REC_POS_BITS);
    if (dc < -32768 || dc > 32767) A.bad = 1;
    const uint32_t a = (uint32_t)(dc < 0 ? -dc : dc);
    A.max_abs_dc = a > A.max_abs_dc ? a : A.max_abs_dc;
    JDA_G(uint32_t, P.blk_index)[g] = jda_index_canonical(p_abs);
    JDA_G(int16_t, P.blk_dc)[g] = (int16_t)dc;
}
JDA_HD void jda_finalize_item(const jda_segscan_params &P, uint32_t seg, uint32_t i, uint32_t g0, uint32_t b0, uint32_t inv, int32_t pr0, int32_t pr1, int32_t pr2, uint32_t rst_from, jda_fin_acc &A)
{
    if (g0 + i > P.n_blocks_total) return;
    jda_finalize_apply(P, seg, i, jda_finalize_load(P, seg, i), g0, b0, inv, pr0, pr1, pr2, rst_from, A);
}
// candidate ci: a magnitude read that some entry lag of its segment truncates.  With the segment's true lag known: does it?  Then
// the block's entry becomes the reference reader's true phase (at the block's first AC symbol) + the flag.  Returns 1 for a
// truncated read (the serial pre-scan's count).
JDA_HD uint32_t jda_resolve_item(const jda_segscan_params &P, uint32_t ci)
{
    const uint32_t JDA_GLOBAL *cd = JDA_G(const uint32_t, P.cands) + (size_t)ci * 4u;
    const uint32_t seg = cd[0], ord = cd[1] & 0xffffu, round = cd[1] >> 16, hit = cd[3];
    const uint32_t JDA_GLOBAL *sum = JDA_G(const uint32_t, P.seg_sum);
    const uint32_t JDA_GLOBAL *st = JDA_G(const uint32_t, P.seg_start);
    if (seg >= P.n_segs || sum[(size_t)seg * JDA_SEG_SUM_WORDS + 7u] != round) return 0;      // the segment was walked again: not its last walk's candidate
    const uint32_t g0 = st[(size_t)seg * 5u], j = st[(size_t)seg * 5u + 4u];
    if (g0 >= 0xfffffff0u || !((hit >> (4u + 5u * j)) & 1u)) return 0;
    uint32_t t = seg, i = ord - 1u, lagw = cd[2], jt = j, g = g0 + i;
    if (ord == 0u) {                                                // the block was open at the segment's entry: it started in ..
        if (seg == 0u) return 0;
        do { t--; } while (t > 0u && sum[(size_t)t * JDA_SEG_SUM_WORDS] == 0u);              // .. the last segment before that starts a block
        const uint32_t nb = sum[(size_t)t * JDA_SEG_SUM_WORDS];
        if (nb == 0u) return 0;
        i = nb - 1u; lagw = sum[(size_t)t * JDA_SEG_SUM_WORDS + 6u]; jt = st[(size_t)t * 5u + 4u]; g = g0 - 1u;
    }
    if (g >= P.n_blocks_total || i >= P.rec_cap) return 0;
    const uint32_t rec = JDA_G(const uint32_t, P.records)[(size_t)t * P.rec_cap + i];
    const uint32_t p_abs = t * JDA_SEG_BITS + (rec & ((1u << JDA_REC_POS_BITS) - 1u));
    const uint32_t u0 = (lagw >> (5u * jt)) & 15u;                  // the window's byte lag at the block's first AC symbol (behind the AC loop's opening refill)
    JDA_G(uint32_t, P.blk_index)[g] = (((p_abs >> 3) - u0) << JDA_INDEX_OFF_BITS) | JDA_INDEX_TRUNC | (8u * u0 + (p_abs & 7u));
    return 1;
}
// interval start nr (1 .. n_intervals - 1): did a settled walk end the interval in front of it, after exactly nr x interval_blocks
// blocks?  (The reference counts MCUs and never looks where the markers were: the two must agree.)  Returns 1 for a mismatch.
JDA_HD uint32_t jda_rst_event_item(const jda_segscan_params &P, uint32_t nr)
{
    const uint32_t JDA_GLOBAL *ev = JDA_G(const uint32_t, P.rst_events) + 2u * (size_t)nr;
    const uint32_t e = ev[0];
    // The reference restarts by MCU count and never looks for the markers.  A marker behind the image's last block by that count
    // is one it never gets to; every other one must stand exactly where the count puts it -- also when a damaged interval has the
    // walk reach the marker only behind the image's last block (found by the pipeline's fuzz: the reference had restarted 2,000 bits
    // earlier, in the middle of what the walk took for one interval).
    const uint64_t want = (uint64_t)nr * P.interval_blocks;
    if (want >= P.n_blocks_total) return 0u;
    if (e == 0u) return 1u;                                         // nobody ended an interval at this marker
    const uint32_t seg = e >> 11, nb = (e & 2047u) - 1u;
    if (seg >= P.n_segs || JDA_G(const uint32_t, P.seg_sum)[(size_t)seg * JDA_SEG_SUM_WORDS + 7u] != ev[1]) return 1u;      // .. not in its segment's last walk
    const uint32_t g0 = JDA_G(const uint32_t, P.seg_start)[(size_t)seg * 5u];
    if (g0 >= 0xfffffff0u) return 0u;                               // behind a bad code: the image is rejected for that
    return (uint64_t)g0 + nb == want ? 0u : 1u;
}

// ================================================================================================
// Tile phases.  Every lane of the tile's wavefront runs each phase; a wave-local fence separates
// consecutive phases (the host emulator runs all 64 lanes of a phase, then the next).
//
//   P0  tables -> LDS, the tile's slice of the scan -> LDS window, zero the list counters
//   P1  thread = block: Huffman/RLE expand into the block's int16[64] in LDS (JPEGDecodeMCU,
//       jpeg.inl:2090-2274); classify the block and append its non-empty columns / its row class to
//       work lists.  Scaled 1/4 and 1/8 outputs are finished here (2x2 IDCT or DC fill).
//   P2  thread = (block, non-empty column): dequant + column stage of the IDCT (jpeg.inl:2553-2679).
//       Empty columns cost nothing -- the reference's own shortcut (:2555-2560), made data-parallel
//       by compacting the work items (wave prefix sum / ballots, jda_p1_lists).
//   P3  thread = (block, row): row stage + range limit (jpeg.inl:2680-2797), blocks grouped by the
//       reference's row variant so that a pass never mixes variants.  8 samples -> the MCU's plane.
//   P4  threads tile the output rows: colour conversion and coalesced stores (JPEGPutMCU*).
// ================================================================================================

// component table ids: the descriptor bytes are workgroup-uniform (scalar loads); pick by the
// lane's component with selects instead of a per-lane indexed global load
JDA_HD uint32_t jda_pick3(const uint8_t ids[3], uint32_t c)
{
    const uint32_t a = ids[0], b = ids[1], d = ids[2];
    return c == 0 ? a : (c == 1 ? b : d);
}

struct jda_tile_ctx {                 // wave-uniform facts about the tile, computed once per thread
    uint32_t first_mcu;               // linear MCU index of the tile's first MCU
    uint32_t count;                   // MCUs of the tile that are decoded (<= MCUS, clipped by n_mcus_ok; 0 = padding tile)
    uint32_t first_block;             // linear block index of the tile's first block
    uint32_t win_lo, win_len;         // bytes of the scan staged in LDS
    uint32_t win_need;                // bytes the tile's lanes can touch (> win_len: some reads go to HBM)
};

// same, from index entries already in registers: ix_first = index[first block of the tile],
// ix_end = index[first block after the tile]
template <int MODE>
JDA_HD jda_tile_ctx jda_tile_setup_from(const jda_dev_desc &D, const jda_strip &S, uint32_t ix_first, uint32_t ix_end, uint32_t win_cap = (uint32_t)jda_lds_layout<MODE>::WIN_BYTES)
{
    typedef jda_mode_traits<MODE> T;
    jda_tile_ctx C;
    C.first_mcu = S.mcu_y * D.mcus_x + S.mcu_x0;
    C.count = S.count;
    if (C.first_mcu >= D.n_mcus_ok) C.count = 0;
    else if (C.first_mcu + C.count > D.n_mcus_ok) C.count = D.n_mcus_ok - C.first_mcu;
    C.first_block = C.first_mcu * T::NBLK;
    C.win_lo = 0; C.win_len = 0; C.win_need = 0;
    if (C.count && D.scale_shift != 3 && !(D.pad_[0] & JDA_DESC_DC_ONLY)) {      // (1/8, a progressive file's DC scan: the DC values are all there is, the scan is not read)
        C.win_lo = (ix_first >> JDA_INDEX_OFF_BITS) & ~15u;
        uint32_t hi = ((ix_end >> JDA_INDEX_OFF_BITS) + 8u + 12u + 15u) & ~15u;
        const uint32_t cap = (D.scan_len + JDA_SCAN_PAD) & ~15u;
        if (hi > cap) hi = cap;
        C.win_len = hi > C.win_lo ? hi - C.win_lo : 0;
        // (the slice is asked for by every lane, whatever its length -- jda_window_load --, so an empty one must still start inside the
        // scan: the streamed pipeline launches a decode before the pre-scan's verdict, the entries of a damaged stream may point anywhere)
        if (C.win_len == 0) C.win_lo = 0;
        C.win_need = C.win_len;
        if (C.win_len > win_cap) C.win_len = win_cap;
    }
    return C;
}

template <int MODE>
JDA_HD jda_tile_ctx jda_tile_setup(const jda_dev_desc &D, const jda_strip &S)
{
    typedef jda_mode_traits<MODE> T;
    jda_tile_ctx C;
    C.first_mcu = S.mcu_y * D.mcus_x + S.mcu_x0;
    C.count = S.count;
    if (C.first_mcu >= D.n_mcus_ok) C.count = 0;
    else if (C.first_mcu + C.count > D.n_mcus_ok) C.count = D.n_mcus_ok - C.first_mcu;
    C.first_block = C.first_mcu * T::NBLK;
    C.win_lo = 0; C.win_len = 0; C.win_need = 0;
    if (C.count && D.scale_shift != 3 && !(D.pad_[0] & JDA_DESC_DC_ONLY)) {
        // the tile's blocks are consecutive in the scan: stage one contiguous run of bytes
        C.win_lo = (JDA_G(const uint32_t, D.blk_index)[C.first_block] >> JDA_INDEX_OFF_BITS) & ~15u;
        // a thread may read 12 bytes past (start of the block after the tile) + 8
        uint32_t hi = ((JDA_G(const uint32_t, D.blk_index)[C.first_block + C.count * T::NBLK] >> JDA_INDEX_OFF_BITS) + 8u + 12u + 15u) & ~15u;
        const uint32_t cap = (D.scan_len + JDA_SCAN_PAD) & ~15u;      // never past the padded allocation
        if (hi > cap) hi = cap;
        C.win_len = hi > C.win_lo ? hi - C.win_lo : 0;
        if (C.win_len == 0) C.win_lo = 0;
        C.win_need = C.win_len;
        if (C.win_len > (uint32_t)jda_lds_layout<MODE>::WIN_BYTES) C.win_len = (uint32_t)jda_lds_layout<MODE>::WIN_BYTES;
    }
    return C;
}

// ---- P0 ---------------------------------------------------------------------------------------
// the set bits of the nibble m, lowest first, in 3-bit fields (unused fields 0)
JDA_HD uint32_t jda_nibble_list(uint32_t m)
{
    uint32_t r = 0, n = 0;
    for (uint32_t b = 0; b < 4; b++) if ((m >> b) & 1u) { r |= b << (3u * n); n++; }
    return r;
}
// tables: once per workgroup (tid = thread in workgroup, nthreads = workgroup size)
// with_long: also the long halves of the AC LUTs (JDA_LT_LONG; tab_lds then holds JDA_LT_BYTES + JDA_LT_LONG_BYTES)
JDA_HD void jda_p0_tables_from(const uint8_t *tables, uint32_t tid, uint32_t nthreads, uint8_t *tab_lds, bool with_long = false);
JDA_HD void jda_p0_tables(const jda_dev_desc &D, uint32_t tid, uint32_t nthreads, uint8_t *tab_lds, bool with_long = false) { jda_p0_tables_from(D.tables, tid, nthreads, tab_lds, with_long); }
JDA_HD void jda_p0_tables_from(const uint8_t *tables, uint32_t tid, uint32_t nthreads, uint8_t *tab_lds, bool with_long)
{
    const jda_chunk16_alias JDA_GLOBAL *blob = JDA_G(const jda_chunk16_alias, tables);
    jda_chunk16_alias *tab = (jda_chunk16_alias *)tab_lds;
    // DC LUTs: blob[0, 2048) -> LT_DC ; AC short halves: blob[2048 + k*4096, +2048) -> LT_AC + k*2048 ;
    // quant: blob[10240, 10752) -> the DC LUTs' unused bytes; zigzag: built below
    for (uint32_t j = tid; j < JDA_ZZ_ENTRIES; j += nthreads) {  // zigzag + flag bits of A.2 in one lookup
        uint32_t v = JDA_ZZ_DUMP;
        if (j < 64) {
            const uint32_t n = JDA_G(const uint8_t, tables)[JDA_TB_ZIGZAG + j];
            v = (n << 1) | ((1u << (n & 7u)) << 8);
        }
        ((uint16_t *)(tab_lds + JDA_LT_ZZ))[j] = (uint16_t)v;
    }
    (void)with_long;
    for (uint32_t i = tid; i < JDA_LT_ZZ / 16; i += nthreads) {
        uint32_t src;
        if (i < 128) {                                          // DC LUTs; chunks 16..31 of each (bytes 256..511) take two quantiser tables
            const uint32_t within = i & 63u;
            src = (within >= 16u && within < 32u) ? (JDA_TB_QUANT >> 4) + (i >> 6) * 16u + (within - 16u) : i;
            if (i == (JDA_LT_EOB >> 4)) src = JDA_TB_EOB >> 4;      // (bytes 64..127 of a DC LUT are never read either)
            if (i == (JDA_LT_NIB >> 4) || i == (JDA_LT_NIB >> 4) + 1u) {
                jda_chunk16_alias nb;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) nb.w[k] = jda_nibble_list((i - (JDA_LT_NIB >> 4)) * 8u + 2u * k) | (jda_nibble_list((i - (JDA_LT_NIB >> 4)) * 8u + 2u * k + 1u) << 16);
                tab[i] = nb;
                continue;
            }
        }
        else src = (JDA_TB_AC >> 4) + (i - 128);                // the AC LUTs, both halves of both tables, in the blob's order
        jda_chunk16_alias c = blob[src];
        if (i >= 128) {                                         // AC entries -> the kernels' layout
#pragma unroll
            for (int k = 0; k < 4; k++) c.w[k] = jda_ac_entry(c.w[k] & 0xffffu) | (jda_ac_entry(c.w[k] >> 16) << 16);
        }
        tab[i] = c;
    }
}

// per wave: zero the list counters, stage the tile's slice of the scan
template <int MODE>
JDA_HD void jda_p0_stage(const jda_dev_desc &D, const jda_tile_ctx &C, uint32_t t, uint8_t *wl, uint32_t win_cap)
{
    typedef jda_lds_layout<MODE> L;
    if (t < 8) ((uint32_t *)(wl + L::CNT_OFF))[t] = 0;
    const uint32_t len = C.win_len < win_cap ? C.win_len : win_cap;
    jda_window_fill(JDA_G(const uint8_t, D.scan), C.win_lo, len, wl + L::WIN_OFF, t);
}

// What a lane of P1 needs that depends only on the image and on the lane -- its block's place in the MCU decides the
// component, hence the Huffman LUTs and the quantiser table -- worked out when a wavefront meets a new image (LDS
// addresses are 32 bits: five registers), not once per tile.
struct jda_lane_pre {          // (offsets, not pointers: a pointer carried around the tile loop loses its address space)
    uint32_t dc_off;          // byte offset in the LDS table copy: DC LUT of the lane's component
    uint32_t ac_off;          // its AC LUT (short half)
    uint32_t ac_long_off;     // byte offset of the AC LUT's long half in the table blob (global)
    uint32_t quant_off;       // byte offset of its quantiser table in the LDS table copy
    uint32_t qsel;            // (offset of the quantiser table / 128) << 10: the column work items carry it
    uint32_t chroma;          // the lane's block is a chroma block
    uint32_t eob_sh, eob_code;   // the next symbol is EOB when (next 32 stream bits >> eob_sh) == eob_code (JDA_TB_EOB)
    uint32_t item_pre;        // qsel | lane << 4: a column work item of the lane's block, less its column (which rides in bits 3:1: a byte offset)
};
template <int MODE>
JDA_HD void jda_lane_prepare(jda_lane_pre &LP, const jda_dev_desc &D, uint32_t lane, const uint8_t *tab)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t b = lane % (uint32_t)T::NBLK;                  // block within the MCU
    const uint32_t c = b < (uint32_t)T::NLUMA ? 0u : b - T::NLUMA + 1u;
    // (jda_pick3 reads the three ids into values first: a select between the array's elements becomes a dynamically
    // indexed load, and that sends the whole descriptor from SGPRs to scratch memory)
    const uint32_t dc_id = jda_pick3(D.dc_id, c), ac_id = jda_pick3(D.ac_id, c), q_id = jda_pick3(D.q_id, c);
    const uint32_t eob = *(const jda_u32_alias *)(tab + JDA_LT_EOB + 4u * ac_id);   // (the image's tables must be staged)
    LP.eob_sh = eob >> 16; LP.eob_code = eob & 0xffffu;
    LP.dc_off = JDA_LT_DC + dc_id * 1024;
    LP.ac_off = JDA_LT_AC + ac_id * 4096;
    LP.ac_long_off = JDA_TB_AC + (ac_id * 2048 + 1024) * 2;
    LP.quant_off = JDA_LT_QUANT_OFF(q_id);
    LP.qsel = (JDA_LT_QUANT_OFF(q_id) >> 7) << 10;                // (the table's offset in units of 128 bytes)
    LP.item_pre = LP.qsel | (lane << 4);
    LP.chroma = b >= (uint32_t)T::NLUMA ? 1u : 0u;
}

// ---- P1 ---------------------------------------------------------------------------------------
// What a thread needs from HBM before it can start decoding its block: issued at kernel entry so the
// two dependent loads (lane schedule -> index entry) overlap P0's table / window staging.
struct jda_p1_inputs { uint32_t lb, ix; int32_t pred; bool active; };

template <int MODE>
JDA_HD jda_p1_inputs jda_p1_prefetch(const jda_dev_desc &D, const jda_tile_ctx &C, uint32_t t)
{
    typedef jda_mode_traits<MODE> T;
    jda_p1_inputs in;
    in.lb = 0; in.ix = 0; in.pred = 0;
    in.active = t < C.count * T::NBLK;
    if (in.active) {
        in.lb = t;
        const uint32_t gb = C.first_block + in.lb;
        in.ix = JDA_G(const uint32_t, D.blk_index)[gb];
        in.pred = JDA_G(const int16_t, D.blk_dc)[gb];
    }
    return in;
}

// Result of a lane's P1: the block's occupancy flags (A.2; 0 = DC-only), or JDA_NO_LIST when the lane has
// nothing for the IDCT work lists (no block, chroma of a luma-only decode, 1/4 and 1/8 scale).
#define JDA_NO_LIST 0xffffffffu

template <int MODE>
JDA_HD uint32_t jda_p1_entropy(const jda_dev_desc &D, const jda_tile_ctx &C, const jda_p1_inputs &in, const jda_lane_pre &LP, const uint8_t *tab, uint8_t *wl,
                               const uint8_t *win, uint32_t win_cap)
{
    typedef jda_lds_layout<MODE> L;
    if (!in.active) return JDA_NO_LIST;
    JDA_P1_TRACE(10);
    const uint32_t lb = in.lb;                                   // (== the lane: LP was made for it)
    if (MODE != JDA_MODE_GRAY && D.gray_from_color && LP.chroma) return JDA_NO_LIST;   // :5225-5233 chroma never decoded
    jda_tables TB;
    TB.dc = tab + LP.dc_off;
    TB.ac_short = (const uint16_t *)(tab + LP.ac_off);
    TB.ac_long = (const uint16_t JDA_GLOBAL *)(JDA_G(const uint8_t, D.tables) + LP.ac_long_off);
    TB.ac_long_lds = TB.ac_short + 1024;
    TB.zz = (const uint16_t *)(tab + JDA_LT_ZZ);
    TB.eob_sh = LP.eob_sh; TB.eob_code = LP.eob_code;
    const int16_t *quant = (const int16_t *)(tab + LP.quant_off);
    int16_t *coef = (int16_t *)(wl + L::COEF_OFF + lb * JDA_COEF_STRIDE);
    uint8_t *plane = (uint8_t *)coef;                            // samples overwrite the block's own slot

    jda_bitreader br;
    br.base = JDA_G(const uint8_t, D.scan);
    br.win = win;
    br.win_lo = C.win_lo;
    br.win_len = C.win_len < win_cap ? C.win_len : win_cap;
    const uint32_t ix = in.ix;
    const bool trunc = (ix & JDA_INDEX_TRUNC) != 0u;             // the entry holds the reference reader's true phase (else maybe a canonical one)
    br.pos = ix >> JDA_INDEX_OFF_BITS;
    br.off = ix & (JDA_INDEX_TRUNC - 1u);
    const uint8_t *wbase = br.win - br.win_lo;
    const int32_t dc = in.pred;                                  // the block's own DC value (index format 2)

    JDA_P1_TRACE(8);
    const int shift = D.scale_shift;
    const bool dc_only = (D.pad_[0] & JDA_DESC_DC_ONLY) != 0;     // wave-uniform: the DC scan of a progressive file -- no AC symbol exists (JPEGDecodeMCU_P with Se = 0)
    if (shift == 3 || (dc_only && shift == 2)) {                 // 1/8: the DC term is the pixel (:5146-5154, bThumbnail) -- the scan is not read at all
        *(jda_u32_alias *)plane = jda_dup8(jda_range_limit5(dc * (int32_t)quant[0]));
        return JDA_NO_LIST;
    }
    if (dc_only) {                                               // (a progressive file asked for at full or half size: DC-only blocks through the IDCT's bypass)
        jda_u64_alias *z = (jda_u64_alias *)coef;
        z[0] = (uint64_t)(uint16_t)dc;
#pragma unroll
        for (int i = 1; i < 16; i++) z[i] = 0;
        return 0;
    }
    // wave-uniform: the whole slice is in LDS (and the tables allow the window-only reader's EOB test)
    const bool win_only = C.win_need <= br.win_len && !(D.pad_[0] & JDA_DESC_GENERAL_P1);
    if (shift == 2) {                                            // 1/4: 2x2 from coefficients 0,1,8,9
        uint32_t flags;
        if (win_only) flags = jda_decode_block_win<5, true, L::LONG_LDS != 0>(br.pos, br.off, wbase, TB, coef, dc, true, trunc);
        else { br.bits = jda_load_be64(br, br.pos); flags = jda_decode_block<5>(br, TB, coef, dc, trunc); }
        const uint32_t px = flags == 0 ? jda_dup8(jda_range_limit5(dc * (int32_t)quant[0]))
                                       : jda_idct_2x2(coef, quant);
        *(jda_u32_alias *)plane = px;
        return JDA_NO_LIST;
    }
    uint32_t flags;
    if (win_only) {
        // the reference's ulBitOff is followed only in a tile that holds a block with a truncated magnitude read (the
        // pre-scan flags those: a fraction of a percent of the blocks of a photograph, none of most synthetic images)
        if (jda_wave_any(trunc)) flags = jda_decode_block_win<64, true, L::LONG_LDS != 0>(br.pos, br.off, wbase, TB, coef, dc, true, trunc);
        else flags = jda_decode_block_win<64, false, L::LONG_LDS != 0>(br.pos, br.off, wbase, TB, coef, dc, true);
    }
    else { br.bits = jda_load_be64(br, br.pos); flags = jda_decode_block<64>(br, TB, coef, dc, trunc); }
    JDA_P1_TRACE(9);
    return flags;
}

// ---- P1 in chunks (images whose index carries continuation entries: photographs, high qualities) -------------------------------
// A wavefront runs P1's symbol loop as long as its longest block, and the blocks of a photograph differ by a factor of ten -- a luma
// block of forty symbols beside chroma blocks of four: 23-41 % of the lanes' trips do work (profiles/r04_p1_lane_balance_estimate.txt).
// With an entry every JDA_CONT_SYMS symbols of a long block (JDA_CONT_*: both pre-scans write them as they pass) the tile's work is
//   pass A   lane = block: the block's first chunk (<= 8 symbols) -- clears the block, stores its DC value;
//   passes B lane = continuation entry e of the tile (they are contiguous in memory, blocks in order: entry C0 + 64 p + lane): the
//            chunk behind it, into ITS block's coefficients; the block's bit position comes from the lane that owns the block
//            (ds_bpermute), the chunk's flag word is ORed into the block's slot (ds_or_b32 on JDA_SLOT_FLAGS);
//   finish   lane = block: own flag word | the slot's.
// A block flagged JDA_INDEX_TRUNC is decoded whole by its own lane in pass A (the reference's ulBitOff has to be followed from the
// block's first symbol); a tile whose slice of the scan does not fit the LDS window takes the general reader, whole blocks.
// cross-lane read (GPU: ds_bpermute, every lane of the wavefront active; host emulator: the array of every lane's value)
JDA_HD uint32_t jda_lane_pull(uint32_t mine, uint32_t src, const uint32_t *all)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)all;
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)mine);
#else
    (void)mine;
    return all[src & 63u];
#endif
}
JDA_HD void jda_lds_or_u32(uint8_t *p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)__hip_atomic_fetch_or((uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *(jda_u32_alias *)p |= v;
#endif
}
#define JDA_P1C_SKIP  0x80000000u    // own.bits: the lane has no block to share (no block, chroma of a luma-only decode, a flagged block, a tile on the general reader)
struct jda_p1c_own { uint32_t bits;  // bit position of the block's first AC symbol | JDA_P1C_SKIP
                     uint32_t fl;    // pass A's flag word (chunk mode: the OR of zigzag entries; whole mode: the block's folded flags)
                     uint32_t whole; // the block was decoded whole in pass A
                     uint32_t listed; };
template <int MODE>
JDA_HD void jda_p1c_tables(jda_tables &TB, const jda_dev_desc &D, uint32_t blk_lane, const uint8_t *tab)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t b = blk_lane % (uint32_t)T::NBLK, c = b < (uint32_t)T::NLUMA ? 0u : b - T::NLUMA + 1u;
    const uint32_t ac_id = jda_pick3(D.ac_id, c);
    const uint32_t eob = *(const jda_u32_alias *)(tab + JDA_LT_EOB + 4u * ac_id);
    TB.dc = tab; TB.ac_short = (const uint16_t *)(tab + JDA_LT_AC + ac_id * 4096u); TB.ac_long = nullptr; TB.ac_long_lds = TB.ac_short + 1024;
    TB.zz = (const uint16_t *)(tab + JDA_LT_ZZ);
    TB.eob_sh = eob >> 16; TB.eob_code = eob & 0xffffu;
}
// pass A.  chunked: the tile takes the chunked path (uniform: its slice is in the window, the tables allow the window reader)
template <int MODE>
JDA_HD jda_p1c_own jda_p1c_block(const jda_dev_desc &D, const jda_tile_ctx &C, const jda_p1_inputs &in, const jda_lane_pre &LP, const uint8_t *tab, uint8_t *wl,
                                 const uint8_t *win, uint32_t win_cap, bool chunked)
{
    typedef jda_lds_layout<MODE> L;
    jda_p1c_own O;
    O.bits = JDA_P1C_SKIP; O.fl = 0; O.whole = 1; O.listed = 0;
    if (!in.active) return O;
    if (MODE != JDA_MODE_GRAY && D.gray_from_color && LP.chroma) return O;
    O.listed = 1;
    jda_tables TB;
    TB.dc = tab + LP.dc_off;
    TB.ac_short = (const uint16_t *)(tab + LP.ac_off);
    TB.ac_long = nullptr;
    TB.ac_long_lds = TB.ac_short + 1024;
    TB.zz = (const uint16_t *)(tab + JDA_LT_ZZ);
    TB.eob_sh = LP.eob_sh; TB.eob_code = LP.eob_code;
    int16_t *coef = (int16_t *)(wl + L::COEF_OFF + in.lb * JDA_COEF_STRIDE);
    const uint32_t ix = in.ix;
    const bool trunc = (ix & JDA_INDEX_TRUNC) != 0u;
    const uint32_t pos = ix >> JDA_INDEX_OFF_BITS, off = ix & (JDA_INDEX_TRUNC - 1u);
    const uint32_t win_len = C.win_len < win_cap ? C.win_len : win_cap;
    const uint8_t *wbase = win - C.win_lo;
    if (!chunked) {                                              // the general reader, whole blocks (as jda_p1_entropy)
        jda_bitreader br;
        br.base = JDA_G(const uint8_t, D.scan); br.win = win; br.win_lo = C.win_lo; br.win_len = win_len; br.pos = pos; br.off = off;
        const bool win_only = C.win_need <= win_len && !(D.pad_[0] & JDA_DESC_GENERAL_P1);
        if (win_only) O.fl = jda_decode_block_win<64, true, true>(pos, off, wbase, TB, coef, in.pred, true, trunc);
        else { br.bits = jda_load_be64(br, br.pos); O.fl = jda_decode_block<64>(br, TB, coef, in.pred, trunc); }
        return O;
    }
    *(jda_u64_alias *)((uint8_t *)coef + JDA_SLOT_FLAGS) = 0;    // the chunks' shared flag word (and the dump bytes behind it)
    if (trunc) {                                                 // (rare) the reference truncates a read of this block: whole, from its exact phase
        O.fl = jda_decode_block_win<64, true, true>(pos, off, wbase, TB, coef, in.pred, true, true);
        return O;
    }
    jda_u64_alias *z = (jda_u64_alias *)coef;
    z[0] = (uint64_t)(uint16_t)in.pred;
#pragma unroll
    for (int i = 1; i < 16; i++) z[i] = 0;
    O.whole = 0;
    O.bits = pos * 8u + off;
    O.fl = jda_decode_chunk_win(O.bits, wbase, TB, coef, 1u, JDA_CONT_SYMS);
    return O;
}
// a pass B item: entry = the continuation entry (valid: this lane has one), owner_bits = the owning lane's jda_p1c_own::bits
template <int MODE>
JDA_HD void jda_p1c_item(const jda_dev_desc &D, const jda_tile_ctx &C, uint32_t entry, uint32_t blk_lane, uint32_t owner_bits, bool valid,
                         const uint8_t *tab, uint8_t *wl, const uint8_t *win)
{
    typedef jda_lds_layout<MODE> L;
    if (!valid || (owner_bits & JDA_P1C_SKIP)) return;
    jda_tables TB;
    jda_p1c_tables<MODE>(TB, D, blk_lane, tab);
    uint8_t *slot = wl + L::COEF_OFF + blk_lane * JDA_COEF_STRIDE;
    const uint32_t fl = jda_decode_chunk_win(owner_bits + JDA_CONT_REL(entry), win - C.win_lo, TB, (int16_t *)slot, JDA_CONT_K(entry), JDA_CONT_SYMS);
    jda_lds_or_u32(slot + JDA_SLOT_FLAGS, fl);
}
// finish: the block's flags for the IDCT work lists (what jda_p1_entropy returns)
template <int MODE>
JDA_HD uint32_t jda_p1c_finish(const jda_p1c_own &O, uint32_t lb, const uint8_t *wl)
{
    typedef jda_lds_layout<MODE> L;
    if (!O.listed) return JDA_NO_LIST;
    if (O.whole) return O.fl;
    return jda_p1c_fold(O.fl | *(const jda_u32_alias *)(wl + L::COEF_OFF + lb * JDA_COEF_STRIDE + JDA_SLOT_FLAGS));
}

// The work lists of the IDCT stages, built by the whole wavefront at once (every lane calls this, with
// its jda_p1_entropy result): the non-empty columns of every block -> two column lists (rows 4-7 empty
// or not, jpeg.inl:2561), every block -> one of the three row-variant lists (:2686-2688) or the DC-only
// list (:5146-5154).  Positions come from a wave prefix sum / ballots; order within a list is irrelevant.
// all_flags: host emulation only (the flags of all 64 lanes).
template <int MODE>
JDA_HD void jda_p1_lists(const jda_dev_desc &D, const jda_lane_pre &LP, uint32_t lane, uint32_t flags, const uint32_t *all_flags, const uint8_t *tab, uint8_t *wl)
{
    typedef jda_lds_layout<MODE> L;
    (void)D;
    uint32_t *cnt = (uint32_t *)(wl + L::CNT_OFF);
    uint8_t *rowlist = wl + L::ROWLIST_OFF;
    uint16_t *collist = (uint16_t *)(wl + L::COLLIST_OFF);
    const bool listed = flags != JDA_NO_LIST;
    const bool has_ac = listed && flags != 0;
    // columns that hold data (column 0 always, :2555); rows 4-7 empty selects the short column stage
    const uint32_t colmask = has_ac ? ((flags & 0xffu) | 1u) : 0u;
    const uint32_t ncols = jda_popcount8(colmask);
    const bool half = (flags & 0x2000u) == 0;
    // one scan for both lists: short-stage columns in the low half of the word, full-stage in the high half
    const uint32_t mine = half ? ncols : (ncols << 16);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t *all_mine = nullptr, *all_cls = nullptr;
#else
    uint32_t all_mine[JDA_TILE_THREADS], all_cls[JDA_TILE_THREADS];
    for (uint32_t l = 0; l < JDA_TILE_THREADS; l++) {
        const uint32_t f = all_flags[l];
        const bool ac = f != JDA_NO_LIST && f != 0;
        const uint32_t n = ac ? jda_popcount8((f & 0xffu) | 1u) : 0u;
        all_mine[l] = (f & 0x2000u) == 0 ? n : (n << 16);
        all_cls[l] = f == JDA_NO_LIST ? 4u : (f == 0 ? 3u : ((f & 0xf0u) ? 2u : ((f & 0xfcu) ? 1u : 0u)));
    }
#endif
    uint32_t total;
    const uint32_t below = jda_wave_excl_sum(mine, lane, all_mine, total);
    // One array: the short-stage items, then the full-stage items.  A block's columns, lowest first, come out of a
    // 16-entry table as 3-bit fields (low nibble, then high nibble + 4), and the block stores EIGHT items from its
    // place on, whatever it has: what it stores past its own items lands on the places of blocks after it -- stored
    // by a later instruction (the eight stores run from the last item to the first, and a later block's item at the
    // same place has a smaller ordinal) -- or past the end of the list (unused entries, then at most 14 bytes of the
    // row list, which is written below).  No condition per column.  A block without columns stores at the end.
    const uint32_t n_half = total & 0xffffu, n_all = n_half + (total >> 16);
    uint32_t base = half ? (below & 0xffffu) : n_half + (below >> 16);
    if (ncols == 0) base = n_all;
    const uint16_t *nib = (const uint16_t *)(tab + JDA_LT_NIB);
    const uint32_t lo = colmask & 15u, hi = colmask >> 4;
    const uint32_t plist = ((uint32_t)nib[lo] | (((uint32_t)nib[hi] + 0x924u) << (3u * jda_popcount8(lo)))) << 1;      // (<< 1: a column rides in an item as its byte offset)
    uint16_t *dst = collist + base;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
#if !defined(__HIP_DEVICE_COMPILE__)
        if ((uint32_t)k >= ncols) continue;          // (the emulator steps the lanes one after another: no overrun there)
#endif
        dst[k] = (uint16_t)(LP.item_pre | ((plist >> (3 * k)) & 0xeu));
        JDA_STORE_ORDER();                           // the ORDER of the eight stores is what makes the overruns harmless
    }
    const uint32_t cls = !listed ? 4u : (flags == 0 ? 3u : ((flags & 0xf0u) ? 2u : ((flags & 0xfcu) ? 1u : 0u)));
    uint32_t n0, n1, n2, n3;
    const uint32_t r0 = jda_wave_class_rank(cls, 0, lane, all_cls, n0);
    const uint32_t r1 = jda_wave_class_rank(cls, 1, lane, all_cls, n1);
    const uint32_t r2 = jda_wave_class_rank(cls, 2, lane, all_cls, n2);
    const uint32_t r3 = jda_wave_class_rank(cls, 3, lane, all_cls, n3);
    const uint32_t rank = cls == 0 ? r0 : (cls == 1 ? r1 : (cls == 2 ? r2 : r3));
    // one array, classes back to back: 0 | 1 | 2 | DC-only
    const uint32_t cbase = cls == 0 ? 0u : (cls == 1 ? n0 : (cls == 2 ? n0 + n1 : n0 + n1 + n2));
    uint8_t *rdst = listed ? rowlist + cbase + rank : (uint8_t *)&cnt[7];
    *rdst = (uint8_t)lane;
    if (lane == 0) {
        cnt[0] = n_half; cnt[1] = total >> 16;
        // The row variant for columns 0-3 gives what the general variant gives on rows whose columns 4-7 are zero (jpeg.inl:2698-2743:
        // 362 - 256 = 106, and z10 = -z13, z12 = z11 make the odd parts the same products) -- so when the blocks of class 1 that do
        // not fill a pass (eight rows each, eight blocks a pass) fit into the idle lanes of class 2's last pass, they move over: the
        // list is class 0 | 1 | 2 | DC-only, the border between 1 and 2 shifts, and class 1 runs one pass less.  (The other
        // variants round differently: nothing else may move.)
        const uint32_t rem1 = n1 & 7u, slack2 = (8u - (n2 & 7u)) & 7u;
        const uint32_t k = (rem1 != 0u && rem1 <= slack2) ? rem1 : 0u;
        cnt[2] = n0; cnt[3] = n1 - k; cnt[4] = n2 + k; cnt[5] = n3;
    }
}

// ---- P2 ---------------------------------------------------------------------------------------
template <int MODE, bool FAST, bool HALF>
JDA_HD void jda_p2_column_item(const jda_dev_desc &D, uint32_t item, const uint8_t *tab, uint8_t *cbase)
{
    typedef jda_mode_traits<MODE> T;
    // item = (quantiser table's offset / 128) << 10 | block << 4 | column << 1: the column as the byte offset it is in both arrays
    // (six instructions from the item to the two addresses; eight with the column as a number)
    const uint32_t blk = (item >> 4) & 63u, col2 = item & 0xeu;
    (void)D; (void)sizeof(T);
    const int16_t *quant = (const int16_t *)(tab + (((item >> 3) & 0x3f80u) | col2));
    int16_t *coef = (int16_t *)(cbase + col2 + blk * JDA_COEF_STRIDE);
    int32_t cv[8], qv[8], r[8];
#pragma unroll
    for (int row = 0; row < 8; row++) {
        if (HALF && row >= 4) { cv[row] = 0; qv[row] = 0; }
        else { cv[row] = coef[row * 8]; qv[row] = quant[row * 8]; }
    }
    jda_idct_col<FAST, HALF>(cv, qv, r);
#pragma unroll
    for (int row = 0; row < 8; row++) coef[row * 8] = (int16_t)r[row];
}

template <int MODE, bool FAST>
JDA_HD void jda_p2_columns(const jda_dev_desc &D, uint32_t t, const uint8_t *tab, uint8_t *wl)
{
    typedef jda_lds_layout<MODE> L;
    const uint32_t *cnt = (const uint32_t *)(wl + L::CNT_OFF);
    const uint16_t *collist = (const uint16_t *)(wl + L::COLLIST_OFF);
    const uint32_t n_half = cnt[0], n_full = cnt[1];
    uint8_t *const cbase = wl + L::COEF_OFF;
    // (one number walks the list and ends the loop)
    const uint16_t *lp = collist + t;
    for (const uint16_t *const lend = collist + n_half; lp < lend; lp += JDA_TILE_THREADS) jda_p2_column_item<MODE, FAST, true>(D, *lp, tab, cbase);
    lp = collist + n_half + t;
    for (const uint16_t *const lend = collist + n_half + n_full; lp < lend; lp += JDA_TILE_THREADS) jda_p2_column_item<MODE, FAST, false>(D, *lp, tab, cbase);
}

// ---- P3 ---------------------------------------------------------------------------------------
template <int MODE, int RC>
JDA_HD void jda_p3_row_class(uint32_t t, uint8_t *wl, uint32_t first, uint32_t n_blocks)
{
    typedef jda_lds_layout<MODE> L;
    // item i = t + 64 pass is row i & 7 = t & 7 of block list[i >> 3] = list[(t >> 3) + 8 pass]: the row and everything that
    // depends on it alone are the lane's for the whole tile loop, and ONE number walks the list and ends the loop (the loop used to
    // carry i, shift it for the list and add the row's offset to the block's address: seven instructions a pass, four now)
    const uint32_t row = t & 7u;
    const uint8_t *lp = wl + L::ROWLIST_OFF + first + (t >> 3);
    const uint8_t *const lend = wl + L::ROWLIST_OFF + first + n_blocks;
#if defined(__HIP_DEVICE_COMPILE__)
    // (one register the compiler cannot see through: it would keep the constant part of the address apart for the store's offset
    // field and add it back in front of every two-dword read, whose offset fields are too narrow for it)
    uint32_t rb32 = JDA_LDS_A32(wl + L::COEF_OFF + row * 16);
    JDA_OPAQUE(rb32);
    uint8_t *const rowbase = (uint8_t *)(uint8_t __attribute__((address_space(3))) *)rb32;
#else
    uint8_t *const rowbase = wl + L::COEF_OFF + row * 16;
#endif
    for (; lp < lend; lp += JDA_TILE_THREADS / 8) {
        const uint32_t blk = *lp;
        uint8_t *const rp = rowbase + blk * JDA_COEF_STRIDE;
        const jda_u64_alias *src = (const jda_u64_alias *)rp;
        int32_t sv[8];
        const uint64_t a = src[0];
        sv[0] = (int16_t)a; sv[1] = (int16_t)(a >> 16); sv[2] = (int16_t)(a >> 32); sv[3] = (int16_t)(a >> 48);
        if (RC == 2) {
            const uint64_t bq = src[1];
            sv[4] = (int16_t)bq; sv[5] = (int16_t)(bq >> 16); sv[6] = (int16_t)(bq >> 32); sv[7] = (int16_t)(bq >> 48);
        } else { sv[4] = sv[5] = sv[6] = sv[7] = 0; }
        const jda_row8 p = jda_idct_row<RC>(sv);
        // in place (jpeg.inl:2682): row r's 8 bytes land on bytes [8r, 8r+8) of the block, i.e. on row
        // r/2's int16 data -- safe because the 8 rows of a block are handled by 8 adjacent lanes of one
        // wavefront in the same instruction (all reads precede all writes), and on the sequential host
        // emulator rows are visited in ascending order
        jda_u32_alias *dst = (jda_u32_alias *)(rp - row * 8);
        dst[0] = p.lo; dst[1] = p.hi;
    }
}

template <int MODE>
JDA_HD void jda_p3_rows(const jda_dev_desc &D, uint32_t t, const uint8_t *tab, uint8_t *wl)
{
    typedef jda_mode_traits<MODE> T;
    typedef jda_lds_layout<MODE> L;
    const uint32_t *cnt = (const uint32_t *)(wl + L::CNT_OFF);
    const uint32_t n0 = cnt[2], n1 = cnt[3], n2 = cnt[4];
    jda_p3_row_class<MODE, 0>(t, wl, 0, n0);
    jda_p3_row_class<MODE, 1>(t, wl, n0, n1);
    jda_p3_row_class<MODE, 2>(t, wl, n0 + n1, n2);
    // DC-only blocks: all 64 samples = RT((pred * q0) >> 5)  (:5146-5154); one thread per block (rare)
    const uint8_t *list = wl + L::ROWLIST_OFF + n0 + n1 + n2;
    const uint32_t n_dc = cnt[5];
    for (uint32_t i = t; i < n_dc; i += JDA_TILE_THREADS) {
        const uint32_t blk = list[i];
        const uint32_t b = blk % T::NBLK;
        const uint32_t c = b < (uint32_t)T::NLUMA ? 0u : b - T::NLUMA + 1u;
        const int32_t q0 = *(const int16_t *)(tab + JDA_LT_QUANT_OFF(jda_pick3(D.q_id, c)));
        const int32_t dc = *(const int16_t *)(wl + L::COEF_OFF + blk * JDA_COEF_STRIDE);
        const uint32_t v = jda_dup8(jda_range_limit5(dc * q0));
        jda_u32_alias *dst = (jda_u32_alias *)(wl + L::COEF_OFF + blk * JDA_COEF_STRIDE);
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = v;
    }
}

// i / d without an integer division per work item: q = (i * ceil(2^22 / d)) >> 22, exact while
// i * d < 2^22 (here i < 16 * d and d <= 384, so i * d < 2.4M) and i * ceil(2^22 / d) < 2^32
JDA_HD uint32_t jda_recip22(uint32_t d) { return ((1u << 22) + d - 1u) / d; }


// ---- P4: colour conversion + coalesced stores ------------------------------------------------------
// four converted pixels -> memory in the requested format.  CLIP: the group may cross the right edge.
// nvalid: how many of the four belong to this tile (a tile's width need not be a multiple of 4, and
// its neighbour -- another wavefront -- owns the pixels right of it).
template <int PT, bool CLIP>
JDA_HD void jda_store4(uint8_t JDA_GLOBAL *row, uint32_t X, uint32_t out_w, const uint32_t v[4], uint32_t nvalid = 4)
{
    uint32_t n = (!CLIP || X + 4 <= out_w) ? 4u : out_w - X;
    if (CLIP && nvalid < n) n = nvalid;
    if (CLIP && (X & 3u)) {                              // tile starts off a 4-pixel boundary: element stores
        for (uint32_t j = 0; j < n; j++) {
            if (PT == JDA_RGB8888) ((jda_u32_alias JDA_GLOBAL *)(row + (size_t)X * 4))[j] = v[j];
            else if (PT == JDA_EIGHT_BIT_GRAYSCALE) row[X + j] = (uint8_t)v[j];
            else ((uint16_t JDA_GLOBAL *)(row + (size_t)X * 2))[j] = (uint16_t)v[j];
        }
        return;
    }
    if (PT == JDA_RGB8888) {
        jda_u32_alias JDA_GLOBAL *d = (jda_u32_alias JDA_GLOBAL *)(row + (size_t)X * 4);
        if (n == 4) { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
        else for (uint32_t j = 0; j < n; j++) d[j] = v[j];
    } else if (PT == JDA_EIGHT_BIT_GRAYSCALE) {
        if (n == 4) *(jda_u32_alias JDA_GLOBAL *)(row + X) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        else for (uint32_t j = 0; j < n; j++) row[X + j] = (uint8_t)v[j];
    } else {
        if (n == 4) { jda_u32_alias JDA_GLOBAL *d = (jda_u32_alias JDA_GLOBAL *)(row + (size_t)X * 2); d[0] = v[0] | (v[1] << 16); d[1] = v[2] | (v[3] << 16); }
        else for (uint32_t j = 0; j < n; j++) ((uint16_t JDA_GLOBAL *)(row + (size_t)X * 2))[j] = (uint16_t)v[j];
    }
}

// runtime pixel type -> the templated store (used by the generic path only)
template <bool CLIP>
JDA_HD void jda_store4_rt(uint8_t JDA_GLOBAL *row, uint32_t X, uint32_t out_w, int pt, const uint32_t v[4], uint32_t nvalid)
{
    if (pt == JDA_RGB8888) jda_store4<JDA_RGB8888, CLIP>(row, X, out_w, v, nvalid);
    else if (pt == JDA_EIGHT_BIT_GRAYSCALE) jda_store4<JDA_EIGHT_BIT_GRAYSCALE, CLIP>(row, X, out_w, v, nvalid);
    else jda_store4<JDA_RGB565_LITTLE_ENDIAN, CLIP>(row, X, out_w, v, nvalid);
}

// one chroma sample shared by a 2x2 (or 1x1) group: the products of jpeg.inl:3158-3161, already
// shifted: ((k*c) + (Y << 12)) >> 12 == Y + ((k*c) >> 12) exactly, because Y << 12 is a multiple of 4096
struct jda_chroma { int32_t r, g, b; };
JDA_HD jda_chroma jda_chroma_terms(uint32_t cb8, uint32_t cr8)
{
    // k * (c - 128) == k * c - 128 k : the level shift folds into the multiply-add's constant
    const int32_t cb = (int32_t)cb8, cr = (int32_t)cr8;
    jda_chroma t;
    t.r = (5742 * cr - 5742 * 128) >> 12;
    t.g = (-1409 * cb - 2925 * cr + (1409 + 2925) * 128) >> 12;
    t.b = (7258 * cb - 7258 * 128) >> 12;
    return t;
}
// The same three terms, each duplicated into both 16-bit halves of a word (the operand of the packed
// pixel-pair adds).  x >> 12 == (16 x) >> 16, and the upper half of a word is picked by the byte
// permute that duplicates it: multiply-add + permute per term, no shift.
struct jda_chroma2 { uint32_t r, g, b; };
// un-shifted terms (16 x the products); the wanted value sits in bits 31:16
JDA_HD jda_chroma2 jda_chroma_terms16(uint32_t cb8, uint32_t cr8)
{
    const int32_t cb = (int32_t)cb8, cr = (int32_t)cr8;
    jda_chroma2 t;                                                // (four multiply-adds: left to itself the compiler makes the green term two multiplies and a three-operand add)
    t.r = (uint32_t)jda_mad24(cr, 16 * 5742, -16 * 5742 * 128);
    int32_t g = jda_mad24(cr, -16 * 2925, 16 * (1409 + 2925) * 128);
    JDA_OPAQUE(g);                                                // (.. as it does when it sees both products at once)
    t.g = (uint32_t)jda_mad24(cb, -16 * 1409, g);
    t.b = (uint32_t)jda_mad24(cb, 16 * 7258, -16 * 7258 * 128);
    return t;
}
JDA_HD uint32_t jda_pack_hi16(uint32_t lo, uint32_t hi) { return jda_perm(hi, lo, 0x07060302u); }   // {hi[31:16], lo[31:16]}
// Two horizontally adjacent RGB8888 pixels, red and green of a pixel side by side in a word.  ypair = Y0 | Y1 << 16; trg0 / trg1 = pixel
// 0's / 1's red term in bits 15:0 and green term in bits 31:16 (the same word for both when the pixels share their chroma sample);
// tb: the blue terms where the caller has them -- BHI: one term in bits 31:16 for both pixels, else pixel 0's in 15:0, pixel 1's in 31:16.
// A pixel costs four instructions (add + saturate for R|G, half an add + saturate for B, one permute that also sets alpha) where
// three channel pairs cost four and a half.
template <bool BHI>
JDA_HD void jda_rgba_pair_rg(uint32_t ypair, uint32_t trg0, uint32_t trg1, uint32_t tb, uint32_t &px0, uint32_t &px1)
{
    const uint32_t rg0 = jda_sat_pk_u8(jda_pk_add16_alo(ypair, trg0));                          // [R0, G0, 0, 0]
    const uint32_t rg1 = jda_sat_pk_u8(jda_pk_add16_ahi(ypair, trg1));                          // [R1, G1, 0, 0]
    const uint32_t b2 = jda_sat_pk_u8(BHI ? jda_pk_add16_bhi(ypair, tb) : jda_pk_add16(ypair, tb));   // [B0, B1, 0, 0]
    px0 = jda_perm(b2, rg0, 0x0d040100u);                                                      // [R0, G0, B0, 0xff]
    px1 = jda_perm(b2, rg1, 0x0d050100u);                                                      // [R1, G1, B1, 0xff]
}
template <int PT>
JDA_HD uint32_t jda_rgb_pixel(uint32_t y8, const jda_chroma &t)
{
    const int32_t y = (int32_t)y8;
    if (PT == JDA_RGB8888) {
        const int32_t r = jda_clamp255(t.r + y);
        int32_t g = jda_clamp255(t.g + y);
        const int32_t b = jda_clamp255(t.b + y);
        JDA_OPAQUE(g);
        return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | 0xff000000u;
    }
    // RGB565 goes through the 10-bit wrapping tables
    const int32_t r = jda_clamp255(jda_sext10_at(t.r + y, 0));
    const int32_t g = jda_clamp255(jda_sext10_at(t.g + y, 0));
    const int32_t b = jda_clamp255(jda_sext10_at(t.b + y, 0));
    uint32_t v = (uint32_t)((r >> 3) << 11) | (uint32_t)((g >> 2) << 5) | (uint32_t)(b >> 3);
    if (PT == JDA_RGB565_BIG_ENDIAN) v = ((v & 0xffu) << 8) | (v >> 8);
    return v;
}

// per 16-bit lane: clamp a signed value to 0..255 (v_pk_max_i16, v_pk_min_i16)
JDA_HD uint32_t jda_pk_clamp255(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short jda_s2 __attribute__((ext_vector_type(2)));
    jda_s2 v = __builtin_bit_cast(jda_s2, a);
    const jda_s2 lo = { 0, 0 }, hi = { 255, 255 };
    v = __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi);
    return __builtin_bit_cast(uint32_t, v);
#else
    int32_t l = (int16_t)(a & 0xffffu), h = (int16_t)(a >> 16);
    l = l < 0 ? 0 : (l > 255 ? 255 : l); h = h < 0 ? 0 : (h > 255 ? 255 : h);
    return (uint32_t)l | ((uint32_t)h << 16);
#endif
}
// Two horizontally adjacent RGB565 pixels at once (JPEGPixelLE / BE, jpeg.inl:3101-3156): pixel 0 in bits 15:0,
// pixel 1 in bits 31:16.  The reference indexes 10-bit-wrapping tables with (Y + term) & 0x3ff; for 8-bit
// samples Y + term stays inside -227..480, where the wrap is the identity, so the tables reduce to a clamp
// and the field extraction (r >> 3) << 11 | (g >> 2) << 5 | b >> 3 works on both 16-bit lanes of a word.
template <int PT>
JDA_HD uint32_t jda_565_pair(uint32_t ypair, uint32_t tr, uint32_t tg, uint32_t tb)
{
    const uint32_t r2 = jda_pk_clamp255(jda_pk_add16(ypair, tr));
    const uint32_t g2 = jda_pk_clamp255(jda_pk_add16(ypair, tg));
    const uint32_t b2 = jda_pk_clamp255(jda_pk_add16(ypair, tb));
    uint32_t v = ((r2 & 0x00f800f8u) << 8) | ((g2 & 0x00fc00fcu) << 3) | ((b2 >> 3) & 0x001f001fu);
    if (PT == JDA_RGB565_BIG_ENDIAN) v = jda_perm(0, v, 0x02030001u);      // swap the bytes of each pixel (:3149)
    return v;
}

template <int PT>
JDA_HD uint32_t jda_565_pair_t16(uint32_t ypair, const jda_chroma2 &t)
{
    const uint32_t r2 = jda_pk_clamp255(jda_pk_add16_bhi(ypair, t.r));
    const uint32_t g2 = jda_pk_clamp255(jda_pk_add16_bhi(ypair, t.g));
    const uint32_t b2 = jda_pk_clamp255(jda_pk_add16_bhi(ypair, t.b));
    uint32_t v = ((r2 & 0x00f800f8u) << 8) | ((g2 & 0x00fc00fcu) << 3) | ((b2 >> 3) & 0x001f001fu);
    if (PT == JDA_RGB565_BIG_ENDIAN) v = jda_perm(0, v, 0x02030001u);
    return v;
}

// Addresses of the colour stage's work items for a FULL 4:2:0 tile (10 MCUs = 160 x 16 pixels = 320 items of 4x2
// pixels = exactly five passes of the wavefront): item i = lane + 64 * pass always lands on the same LDS offsets and
// on the same offset from the tile's first output pixel, so the index arithmetic (a third of the stage's VALU work)
// is done once per image and kept in registers.  Only the output offset depends on the image (pitch, pixel size).
// The same for a FULL 4:4:4 tile (20 MCUs = 160 x 8 pixels = 320 items of 4 pixels of one row = five passes):
// yo = the luma bytes (Cb, Cr one and two block slots further), rel as above; co is not used.
#define JDA_P4_PASSES 5
#define JDA_P4_PASSES_444 5
struct jda_p4_pre { uint32_t yo[JDA_P4_PASSES_444], co[JDA_P4_PASSES], rel[JDA_P4_PASSES_444]; };
JDA_HD void jda_p4_precompute(jda_p4_pre &P, uint32_t t, uint32_t plane_stride, uint32_t pitch, uint32_t bpp)
{
#pragma unroll
    for (int it = 0; it < JDA_P4_PASSES; it++) {
        const uint32_t i = t + 64u * (uint32_t)it, rp = i / 40u, g = i - rp * 40u;
        const uint32_t po = (g >> 2) * plane_stride;
        P.yo[it] = po + (rp >> 2) * (2 * JDA_COEF_STRIDE) + (rp & 3u) * 16 + ((g >> 1) & 1u) * JDA_COEF_STRIDE + (g & 1u) * 4;
        P.co[it] = po + 4 * JDA_COEF_STRIDE + rp * 8 + (g & 3u) * 2;
        P.rel[it] = rp * 2u * pitch + g * 4u * bpp;
    }
}

JDA_HD void jda_p4_precompute_444(jda_p4_pre &P, uint32_t t, uint32_t plane_stride, uint32_t pitch, uint32_t bpp)
{
#pragma unroll
    for (int it = 0; it < JDA_P4_PASSES_444; it++) {
        const uint32_t i = t + 64u * (uint32_t)it, r = i / 40u, x4 = (i - r * 40u) * 4u;    // 40 groups of 4 pixels per row
        P.yo[it] = (x4 >> 3) * plane_stride + r * 8u + (x4 & 7u);
        P.rel[it] = r * pitch + x4 * bpp;
    }
#pragma unroll
    for (int it = 0; it < JDA_P4_PASSES; it++) P.co[it] = 0;
}

// full-size 4:2:0 colour output (JPEGPutMCU22 scalar body, jpeg.inl:4333-4543): a work item is a 4x2
// pixel group (the two rows share their chroma samples); items are dealt to the threads in row-major
// order so that consecutive threads store consecutive 16-byte groups.
// the eight pixels of one item: ya / yb = four luma samples of the upper / lower row, cb2 / cr2 = two chroma samples each
template <int PT>
JDA_HD void jda_p4_420_item(uint32_t ya, uint32_t yb, uint32_t cb0, uint32_t cb1, uint32_t cr0, uint32_t cr1, uint32_t v0[4], uint32_t v1[4])
{
    // (a chroma sample's three terms serve the two pixels of a pair in both rows: they stay in the upper halves of their words, the
    // packed adds pick them up there; the four chroma samples come as bytes of their own -- ds_read_u8 -- so nothing is spent on
    // taking a 16-bit load apart)
    const jda_chroma2 d0 = jda_chroma_terms16(cb0, cr0);
    const jda_chroma2 d1 = jda_chroma_terms16(cb1, cr1);
    if (PT == JDA_RGB8888) {
        const uint32_t rg0 = jda_pack_hi16(d0.r, d0.g), rg1 = jda_pack_hi16(d1.r, d1.g);
        jda_rgba_pair_rg<true>(jda_perm(0, ya, 0x0c010c00u), rg0, rg0, d0.b, v0[0], v0[1]);
        jda_rgba_pair_rg<true>(jda_perm(0, ya, 0x0c030c02u), rg1, rg1, d1.b, v0[2], v0[3]);
        jda_rgba_pair_rg<true>(jda_perm(0, yb, 0x0c010c00u), rg0, rg0, d0.b, v1[0], v1[1]);
        jda_rgba_pair_rg<true>(jda_perm(0, yb, 0x0c030c02u), rg1, rg1, d1.b, v1[2], v1[3]);
    } else {                                                  // RGB565: v[0], v[1] hold pixel pairs
        v0[0] = jda_565_pair_t16<PT>(jda_perm(0, ya, 0x0c010c00u), d0);
        v0[1] = jda_565_pair_t16<PT>(jda_perm(0, ya, 0x0c030c02u), d1);
        v1[0] = jda_565_pair_t16<PT>(jda_perm(0, yb, 0x0c010c00u), d0);
        v1[1] = jda_565_pair_t16<PT>(jda_perm(0, yb, 0x0c030c02u), d1);
        v0[2] = v0[3] = v1[2] = v1[3] = 0;
    }
}
// item -> memory, whole groups: 16 (RGB8888) or 8 (RGB565) bytes per row
template <int PT>
JDA_HD void jda_p4_420_store(uint8_t JDA_GLOBAL *out, uint32_t off, uint32_t off1, const uint32_t v0[4], const uint32_t v1[4])
{
    if (PT == JDA_RGB8888) {
        jda_chunk16_alias q0, q1;
        q0.w[0] = v0[0]; q0.w[1] = v0[1]; q0.w[2] = v0[2]; q0.w[3] = v0[3];
        q1.w[0] = v1[0]; q1.w[1] = v1[1]; q1.w[2] = v1[2]; q1.w[3] = v1[3];
        *(jda_chunk16_alias JDA_GLOBAL *)(out + off) = q0;
        *(jda_chunk16_alias JDA_GLOBAL *)(out + off1) = q1;
    } else {
        *(jda_u64_alias JDA_GLOBAL *)(out + off) = (uint64_t)v0[0] | ((uint64_t)v0[1] << 32);
        *(jda_u64_alias JDA_GLOBAL *)(out + off1) = (uint64_t)v1[0] | ((uint64_t)v1[1] << 32);
    }
}

// a full, unclipped tile: five passes with the precomputed item addresses
template <int PT>
JDA_HD void jda_p4_420_full10(const jda_dev_desc &D, const jda_p4_pre &P, const uint8_t *plane_base, uint32_t x_base, uint32_t y_base)
{
    const uint32_t bpp = PT == JDA_RGB8888 ? 4u : 2u;
    const uint32_t pitch = D.out_pitch;
    uint8_t JDA_GLOBAL *tile = JDA_G(uint8_t, D.out) + (y_base * pitch + x_base * bpp);      // uniform
#pragma unroll
    for (int it = 0; it < JDA_P4_PASSES; it++) {
        const uint32_t ya = *(const jda_u32_alias *)(plane_base + P.yo[it]), yb = *(const jda_u32_alias *)(plane_base + P.yo[it] + 8);
        const uint8_t *cp = plane_base + P.co[it];
        uint32_t v0[4], v1[4];
        uint32_t cb0, cr0, cb1, cr1;
        jda_lds_bytes_apart(cp, 0, JDA_COEF_STRIDE, cb0, cr0);
        jda_lds_bytes_apart(cp, 1, JDA_COEF_STRIDE, cb1, cr1);
        jda_p4_420_item<PT>(ya, yb, cb0, cb1, cr0, cr1, v0, v1);
        jda_p4_420_store<PT>(tile, P.rel[it], P.rel[it] + pitch, v0, v1);
    }
}

template <int PT, bool CLIP>
JDA_HD void jda_p4_420_full(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                            uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const uint32_t groups = tile_w >> 2;                          // 4-pixel groups per row (tile_w is a multiple of 16)
    const uint32_t inv = jda_recip22(groups);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    const uint32_t pitch = D.out_pitch;
    // byte offset of the tile's first pixel: 32-bit arithmetic from here on (surfaces are < 4 GiB, checked
    // by jda_batch_create), so an address is the uniform base + one 32-bit lane offset
    const uint32_t bpp = PT == JDA_RGB8888 ? 4u : 2u;
    const uint32_t tile_off = y_base * pitch + x_base * bpp;
    for (uint32_t i = t; i < groups * 8; i += JDA_TILE_THREADS) {
        const uint32_t rp = jda_umul24(i, inv) >> 22, g = i - jda_umul24(rp, groups);   // row pair, 4-pixel group in the row
        const uint32_t x4 = g * 4;
        const uint32_t Y0 = y_base + 2 * rp, X = x_base + x4;
        if (CLIP && (Y0 >= D.out_rows || X >= D.out_w)) continue;
        // LDS byte offsets within the tile's planes: MCU g>>2; luma block (rp>>2)*2 + ((g>>1)&1), row 2rp&7,
        // column (g&1)*4; chroma row rp, column (g&3)*2
        const uint32_t po = jda_umul24(g >> 2, plane_stride);
        const uint32_t yo = po + (rp >> 2) * (2 * JDA_COEF_STRIDE) + (rp & 3u) * 16 + ((g >> 1) & 1u) * JDA_COEF_STRIDE + (g & 1u) * 4;
        const uint32_t co = po + 4 * JDA_COEF_STRIDE + rp * 8 + (g & 3u) * 2;
        const uint32_t ya = *(const jda_u32_alias *)(plane_base + yo), yb = *(const jda_u32_alias *)(plane_base + yo + 8);
        const uint8_t *cp = plane_base + co;
        uint32_t v0[4], v1[4];
        uint32_t cb0, cr0, cb1, cr1;
        jda_lds_bytes_apart(cp, 0, JDA_COEF_STRIDE, cb0, cr0);
        jda_lds_bytes_apart(cp, 1, JDA_COEF_STRIDE, cb1, cr1);
        jda_p4_420_item<PT>(ya, yb, cb0, cb1, cr0, cr1, v0, v1);
        if (!CLIP) {                                              // whole groups, 16 / 8 bytes per row
            const uint32_t off = tile_off + jda_umul24(rp, 2 * pitch) + x4 * bpp;
            jda_p4_420_store<PT>(out, off, off + pitch, v0, v1);
        } else {
            if (PT != JDA_RGB8888) {                              // pixel pairs -> single pixels for the clipped stores
                const uint32_t a01 = v0[0], a23 = v0[1], b01 = v1[0], b23 = v1[1];
                v0[0] = a01 & 0xffffu; v0[1] = a01 >> 16; v0[2] = a23 & 0xffffu; v0[3] = a23 >> 16;
                v1[0] = b01 & 0xffffu; v1[1] = b01 >> 16; v1[2] = b23 & 0xffffu; v1[3] = b23 >> 16;
            }
            uint8_t JDA_GLOBAL *row0 = out + (size_t)Y0 * pitch;
            jda_store4<PT, CLIP>(row0, X, D.out_w, v0);
            if (Y0 + 1 < D.out_rows) jda_store4<PT, CLIP>(row0 + pitch, X, D.out_w, v1);
        }
    }
}

// full-size 4:4:4 colour output (JPEGPutMCU11 scalar body, jpeg.inl:3519-3559)
template <int PT, bool CLIP>
JDA_HD void jda_p4_444_full(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                            uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const uint32_t groups = tile_w >> 2;
    const uint32_t inv = jda_recip22(groups);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    for (uint32_t i = t; i < groups * 8; i += JDA_TILE_THREADS) {
        const uint32_t r = (i * inv) >> 22, x4 = (i - r * groups) * 4;
        const uint32_t Y = y_base + r, X = x_base + x4;
        if (CLIP && (Y >= D.out_rows || X >= D.out_w)) continue;
        const uint8_t *P = plane_base + (x4 >> 3) * plane_stride + r * 8 + (x4 & 7u);
        const uint32_t y = *(const jda_u32_alias *)P, cb = *(const jda_u32_alias *)(P + JDA_COEF_STRIDE), cr = *(const jda_u32_alias *)(P + 2 * JDA_COEF_STRIDE);
        uint32_t v[4];
        if (PT == JDA_RGB8888) {
            jda_chroma2 c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) c[j] = jda_chroma_terms16((cb >> (8 * j)) & 255u, (cr >> (8 * j)) & 255u);
            jda_rgba_pair_rg<false>(jda_perm(0, y, 0x0c010c00u), jda_pack_hi16(c[0].r, c[0].g), jda_pack_hi16(c[1].r, c[1].g), jda_pack_hi16(c[0].b, c[1].b), v[0], v[1]);
            jda_rgba_pair_rg<false>(jda_perm(0, y, 0x0c030c02u), jda_pack_hi16(c[2].r, c[2].g), jda_pack_hi16(c[3].r, c[3].g), jda_pack_hi16(c[2].b, c[3].b), v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                v[j] = jda_rgb_pixel<PT>((y >> (8 * j)) & 255u, jda_chroma_terms((cb >> (8 * j)) & 255u, (cr >> (8 * j)) & 255u));
        }
        jda_store4<PT, CLIP>(out + (size_t)Y * D.out_pitch, X, D.out_w, v);
    }
}

// a full, unclipped 4:4:4 tile: five passes with the precomputed item addresses
template <int PT>
JDA_HD void jda_p4_444_full21(const jda_dev_desc &D, const jda_p4_pre &P, uint32_t t, const uint8_t *plane_base, uint32_t x_base, uint32_t y_base)
{
    const uint32_t bpp = PT == JDA_RGB8888 ? 4u : 2u;
    uint8_t JDA_GLOBAL *tile = JDA_G(uint8_t, D.out) + (y_base * D.out_pitch + x_base * bpp);      // uniform
#pragma unroll
    for (int it = 0; it < JDA_P4_PASSES_444; it++) {
        const uint8_t *Pp = plane_base + P.yo[it];
        const uint32_t y = *(const jda_u32_alias *)Pp, cb = *(const jda_u32_alias *)(Pp + JDA_COEF_STRIDE), cr = *(const jda_u32_alias *)(Pp + 2 * JDA_COEF_STRIDE);
        uint32_t v[4];
        if (PT == JDA_RGB8888) {
            jda_chroma2 c[4];
#pragma unroll
            for (int j = 0; j < 4; j++) c[j] = jda_chroma_terms16((cb >> (8 * j)) & 255u, (cr >> (8 * j)) & 255u);
            jda_rgba_pair_rg<false>(jda_perm(0, y, 0x0c010c00u), jda_pack_hi16(c[0].r, c[0].g), jda_pack_hi16(c[1].r, c[1].g), jda_pack_hi16(c[0].b, c[1].b), v[0], v[1]);
            jda_rgba_pair_rg<false>(jda_perm(0, y, 0x0c030c02u), jda_pack_hi16(c[2].r, c[2].g), jda_pack_hi16(c[3].r, c[3].g), jda_pack_hi16(c[2].b, c[3].b), v[2], v[3]);
            jda_chunk16_alias q;
            q.w[0] = v[0]; q.w[1] = v[1]; q.w[2] = v[2]; q.w[3] = v[3];
            *(jda_chunk16_alias JDA_GLOBAL *)(tile + P.rel[it]) = q;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                v[j] = jda_rgb_pixel<PT>((y >> (8 * j)) & 255u, jda_chroma_terms((cb >> (8 * j)) & 255u, (cr >> (8 * j)) & 255u));
            *(jda_u64_alias JDA_GLOBAL *)(tile + P.rel[it]) = (uint64_t)(v[0] | (v[1] << 16)) | ((uint64_t)(v[2] | (v[3] << 16)) << 32);
        }
    }
}

// full-size 4:2:2 colour output (JPEGPutMCU21 full-size body, jpeg.inl:4839-4867): a work item is 4 pixels of one
// row; pixels 0-1 share one chroma sample, pixels 2-3 the next
template <int PT, bool CLIP>
JDA_HD void jda_p4_422_full(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                            uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const uint32_t groups = tile_w >> 2;                          // tile_w is a multiple of 16
    const uint32_t inv = jda_recip22(groups);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    for (uint32_t i = t; i < groups * 8; i += JDA_TILE_THREADS) {
        const uint32_t r = jda_umul24(i, inv) >> 22, g = i - jda_umul24(r, groups);
        const uint32_t x4 = g * 4;
        const uint32_t Y = y_base + r, X = x_base + x4;
        if (CLIP && (Y >= D.out_rows || X >= D.out_w)) continue;
        // MCU g>>2, luma block (g>>1)&1, columns (g&1)*4..; chroma row r, columns ((g>>1)&1)*4 + (g&1)*2 ..
        const uint32_t po = jda_umul24(g >> 2, plane_stride);
        const uint32_t yo = po + ((g >> 1) & 1u) * JDA_COEF_STRIDE + r * 8 + (g & 1u) * 4;
        const uint32_t co = po + 2 * JDA_COEF_STRIDE + r * 8 + (g & 3u) * 2;
        const uint32_t y = *(const jda_u32_alias *)(plane_base + yo);
        const uint32_t cb2 = *(const uint16_t *)(plane_base + co), cr2 = *(const uint16_t *)(plane_base + co + JDA_COEF_STRIDE);
        uint32_t v[4];
        if (PT == JDA_RGB8888) {
            const uint8_t *cp = plane_base + co;
            uint32_t cb0, cr0, cb1, cr1;
            jda_lds_bytes_apart(cp, 0, JDA_COEF_STRIDE, cb0, cr0);
            jda_lds_bytes_apart(cp, 1, JDA_COEF_STRIDE, cb1, cr1);
            const jda_chroma2 d0 = jda_chroma_terms16(cb0, cr0);
            const jda_chroma2 d1 = jda_chroma_terms16(cb1, cr1);
            const uint32_t rg0 = jda_pack_hi16(d0.r, d0.g), rg1 = jda_pack_hi16(d1.r, d1.g);
            jda_rgba_pair_rg<true>(jda_perm(0, y, 0x0c010c00u), rg0, rg0, d0.b, v[0], v[1]);
            jda_rgba_pair_rg<true>(jda_perm(0, y, 0x0c030c02u), rg1, rg1, d1.b, v[2], v[3]);
        } else {
            const jda_chroma2 d0 = jda_chroma_terms16(cb2 & 255u, cr2 & 255u);
            const jda_chroma2 d1 = jda_chroma_terms16(cb2 >> 8, cr2 >> 8);
            const uint32_t a01 = jda_565_pair_t16<PT>(jda_perm(0, y, 0x0c010c00u), d0);
            const uint32_t a23 = jda_565_pair_t16<PT>(jda_perm(0, y, 0x0c030c02u), d1);
            v[0] = a01 & 0xffffu; v[1] = a01 >> 16; v[2] = a23 & 0xffffu; v[3] = a23 >> 16;
        }
        jda_store4<PT, CLIP>(out + (size_t)Y * D.out_pitch, X, D.out_w, v);
    }
}

// full-size 8-bit gray output of any source layout (JPEGPutMCU8BitGray full-size bodies, jpeg.inl:2828-2837,
// :2893-2902, :2943-2952, :3000-3034): the luma samples are the pixels.  A work item is one row of one luma block
// (8 bytes); items are dealt row-major so that consecutive lanes store consecutive 8-byte chunks.
template <int MODE, bool CLIP>
JDA_HD void jda_p4_gray8_full(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                              uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t nbx = (uint32_t)T::MCU_W / 8u;                 // luma blocks across an MCU: 1 or 2
    const uint32_t chunks = tile_w >> 3;                          // 8-pixel chunks per row of the tile
    const uint32_t inv = jda_recip22(chunks);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    const uint32_t pitch = D.out_pitch;
    const uint32_t tile_off = y_base * pitch + x_base;
    if (MODE == JDA_MODE_GRAY && !CLIP && chunks == JDA_TILE_THREADS) {      // a full tile of a gray file: pass = row, lane = block
        const jda_u32_alias *src = (const jda_u32_alias *)(plane_base + jda_umul24(t, plane_stride));
        uint8_t JDA_GLOBAL *dst = out + tile_off + t * 8u;
#pragma unroll
        for (uint32_t r = 0; r < 8; r++)
            *(jda_u64_alias JDA_GLOBAL *)(dst + jda_umul24(r, pitch)) = (uint64_t)src[2 * r] | ((uint64_t)src[2 * r + 1] << 32);
        return;
    }
    for (uint32_t i = t; i < chunks * (uint32_t)T::MCU_H; i += JDA_TILE_THREADS) {
        const uint32_t r = jda_umul24(i, inv) >> 22, c = i - jda_umul24(r, chunks);
        const uint32_t m = nbx == 2 ? (c >> 1) : c, bxq = nbx == 2 ? (c & 1u) : 0u;
        const uint32_t q = (r >> 3) * nbx + bxq;                  // luma block inside the MCU
        const jda_u32_alias *src = (const jda_u32_alias *)(plane_base + jda_umul24(m, plane_stride) + q * JDA_COEF_STRIDE + (r & 7u) * 8);
        const uint32_t lo = src[0], hi = src[1];
        const uint32_t X = x_base + c * 8, Y = y_base + r;
        if (!CLIP) {
            *(jda_u64_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + c * 8) = (uint64_t)lo | ((uint64_t)hi << 32);
        } else {
            if (Y >= D.out_rows || X >= D.out_w) continue;
            uint8_t JDA_GLOBAL *row = out + (size_t)Y * pitch;
            const uint32_t n = X + 8 <= D.out_w ? 8u : D.out_w - X;
            const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
            if (n == 8) *(jda_u64_alias JDA_GLOBAL *)(row + X) = v;
            else for (uint32_t j = 0; j < n; j++) row[X + j] = (uint8_t)(v >> (8 * j));
        }
    }
}

// half-size 8-bit gray output of any source layout (JPEGPutMCU8BitGray half-size bodies, jpeg.inl:2812-2827, :2857-2873,
// :2907-2923, :2979-2999): a pixel = (the 2x2 luma samples' sum + 2) >> 2.  A work item is one output row of one luma block:
// two source rows (16 bytes) -> 4 pixels, the sums two at a time in the halves of a word; items are dealt row-major so
// that consecutive lanes store consecutive dwords.
JDA_HD uint32_t jda_pair_sums(uint32_t v) { return (v & 0x00ff00ffu) + ((v >> 8) & 0x00ff00ffu); }       // [b0 + b1, b2 + b3] as 16-bit halves
// four output pixels of the half-size gray image from two source rows of eight samples (a0 a1 / b0 b1): (2x2 sum + 2) >> 2 each.
// A 2x2 sum is two byte dot products with ones in two places (the second accumulates onto the first, which starts at 2); the four
// sums are shifted two at a time and their bytes picked by one permute (13 instructions where masks and adds take 21).
JDA_HD uint32_t jda_half_gray4(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1)
{
    const uint32_t lo = 0x00000101u, hi = 0x01010000u;
    const uint32_t s0 = jda_udot4_acc(b0, lo, jda_udot4_acc(a0, lo, 2u)), s1 = jda_udot4_acc(b0, hi, jda_udot4_acc(a0, hi, 2u));
    const uint32_t s2 = jda_udot4_acc(b1, lo, jda_udot4_acc(a1, lo, 2u)), s3 = jda_udot4_acc(b1, hi, jda_udot4_acc(a1, hi, 2u));
    return jda_perm(((s3 << 16) | s2) >> 2, ((s1 << 16) | s0) >> 2, 0x06040200u);      // (sums <= 1022: the shifted halves are the bytes)
}
template <int MODE, bool CLIP>
JDA_HD void jda_p4_gray8_half(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                              uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t nbx = (uint32_t)T::MCU_W / 8u;                 // luma blocks across an MCU: 1 or 2
    const uint32_t chunks = tile_w >> 2;                          // 4-pixel chunks per output row of the tile (tile_w: output pixels)
    const uint32_t inv = jda_recip22(chunks);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    const uint32_t pitch = D.out_pitch;
    const uint32_t tile_off = y_base * pitch + x_base;
    if (MODE == JDA_MODE_GRAY && !CLIP && chunks == JDA_TILE_THREADS) {      // a full tile of a gray file: pass = output row, lane = block
        const jda_u32_alias *src = (const jda_u32_alias *)(plane_base + jda_umul24(t, plane_stride));
        uint8_t JDA_GLOBAL *dst = out + tile_off + t * 4u;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            *(jda_u32_alias JDA_GLOBAL *)(dst + jda_umul24(r, pitch)) = jda_half_gray4(src[4 * r], src[4 * r + 1], src[4 * r + 2], src[4 * r + 3]);
        }
        return;
    }
    for (uint32_t i = t; i < chunks * ((uint32_t)T::MCU_H >> 1); i += JDA_TILE_THREADS) {
        const uint32_t r = jda_umul24(i, inv) >> 22, c = i - jda_umul24(r, chunks);
        const uint32_t m = nbx == 2 ? (c >> 1) : c, bxq = nbx == 2 ? (c & 1u) : 0u;
        const uint32_t q = (r >> 2) * nbx + bxq;                  // luma block inside the MCU
        const jda_u32_alias *src = (const jda_u32_alias *)(plane_base + jda_umul24(m, plane_stride) + q * JDA_COEF_STRIDE + (r & 3u) * 16);
        const uint32_t v = jda_half_gray4(src[0], src[1], src[2], src[3]);
        const uint32_t X = x_base + c * 4, Y = y_base + r;
        if (!CLIP) *(jda_u32_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + c * 4) = v;
        else {
            if (Y >= D.out_rows || X >= D.out_w) continue;
            uint8_t JDA_GLOBAL *row = out + (size_t)Y * pitch;
            const uint32_t n = X + 4 <= D.out_w ? 4u : D.out_w - X;
            if (n == 4) *(jda_u32_alias JDA_GLOBAL *)(row + X) = v;
            else for (uint32_t j = 0; j < n; j++) row[X + j] = (uint8_t)(v >> (8 * j));
        }
    }
}

// Half-size RGB8888 output of 4:2:0 (JPEGPutMCU22 half-size body, jpeg.inl:3577-3626): a pixel's luma is the SUM of its 2x2
// samples, << 10, its chroma the one sample at its place (no averaging), and each channel clamp((k . c + (sum << 10)) >> 12)
// = clamp((((k . c) >> 10) + sum) >> 2) -- two pixels per instruction in the halves of a word.  A work item is 4 pixels of
// one output row: two rows of one luma block (16 bytes) and four Cb / Cr samples.
// per 16-bit lane: arithmetic shift right by 2 (v_pk_ashrrev_i16)
JDA_HD uint32_t jda_pk_ashr2(uint32_t a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short jda_s2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(jda_s2, a) >> 2);
#else
    const int32_t lo = (int16_t)(a & 0xffffu), hi = (int16_t)(a >> 16);
    return ((uint32_t)(lo >> 2) & 0xffffu) | ((uint32_t)(hi >> 2) << 16);
#endif
}
// the three chroma terms x 64 (the wanted (k . c) >> 10 sits in bits 31:16)
JDA_HD jda_chroma2 jda_chroma_terms64(uint32_t cb8, uint32_t cr8)
{
    const int32_t cb = (int32_t)cb8, cr = (int32_t)cr8;
    jda_chroma2 t;
    t.r = (uint32_t)(64 * 5742 * cr - 64 * 5742 * 128);
    t.g = (uint32_t)(-64 * 1409 * cb - 64 * 2925 * cr + 64 * (1409 + 2925) * 128);
    t.b = (uint32_t)(64 * 7258 * cb - 64 * 7258 * 128);
    return t;
}
JDA_HD void jda_rgba_pair_half(uint32_t ysum2, uint32_t tr, uint32_t tg, uint32_t tb, uint32_t &px0, uint32_t &px1)
{
    const uint32_t r2 = jda_sat_pk_u8(jda_pk_ashr2(jda_pk_add16(ysum2, tr)));
    const uint32_t g2 = jda_sat_pk_u8(jda_pk_ashr2(jda_pk_add16(ysum2, tg)));
    const uint32_t b2 = jda_sat_pk_u8(jda_pk_ashr2(jda_pk_add16(ysum2, tb)));
    const uint32_t rg = jda_perm(g2, r2, 0x05010400u);
    px0 = jda_perm(b2, rg, 0x0d040100u);
    px1 = jda_perm(b2, rg, 0x0d050302u);
}
// the same for RGB565 (JPEGPixelLE / BE through the 10-bit wrapping tables: the identity on this value range, as in jda_565_pair)
template <int PT>
JDA_HD uint32_t jda_565_pair_half(uint32_t ysum2, uint32_t tr, uint32_t tg, uint32_t tb)
{
    const uint32_t r2 = jda_pk_clamp255(jda_pk_ashr2(jda_pk_add16(ysum2, tr)));
    const uint32_t g2 = jda_pk_clamp255(jda_pk_ashr2(jda_pk_add16(ysum2, tg)));
    const uint32_t b2 = jda_pk_clamp255(jda_pk_ashr2(jda_pk_add16(ysum2, tb)));
    uint32_t v = ((r2 & 0x00f800f8u) << 8) | ((g2 & 0x00fc00fcu) << 3) | ((b2 >> 3) & 0x001f001fu);
    if (PT == JDA_RGB565_BIG_ENDIAN) v = jda_perm(0, v, 0x02030001u);
    return v;
}
template <int PT, bool CLIP>
JDA_HD void jda_p4_420_half_rgba(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                                 uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const uint32_t bpp = PT == JDA_RGB8888 ? 4u : 2u;
    const uint32_t groups = tile_w >> 2;                          // tile_w (output pixels) is a multiple of 8
    const uint32_t inv = jda_recip22(groups);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    const uint32_t pitch = D.out_pitch;
    const uint32_t tile_off = y_base * pitch + x_base * bpp;
    for (uint32_t i = t; i < groups * 8; i += JDA_TILE_THREADS) {
        const uint32_t r = jda_umul24(i, inv) >> 22, g = i - jda_umul24(r, groups);     // output row 0..7, group in the row
        const uint32_t X = x_base + g * 4, Y = y_base + r;
        if (CLIP && (Y >= D.out_rows || X >= D.out_w)) continue;
        // MCU g >> 1, luma block (r >> 2) * 2 + (g & 1), its rows 2 (r & 3) and 2 (r & 3) + 1; chroma row r, columns (g & 1) * 4 ..
        const uint32_t po = jda_umul24(g >> 1, plane_stride);
        const jda_u32_alias *ys = (const jda_u32_alias *)(plane_base + po + ((r >> 2) * 2u + (g & 1u)) * JDA_COEF_STRIDE + (r & 3u) * 16);
        const uint32_t co = po + 4 * JDA_COEF_STRIDE + r * 8 + (g & 1u) * 4;
        const uint32_t s01 = jda_pair_sums(ys[0]) + jda_pair_sums(ys[2]), s23 = jda_pair_sums(ys[1]) + jda_pair_sums(ys[3]);
        const uint32_t cb = *(const jda_u32_alias *)(plane_base + co), cr = *(const jda_u32_alias *)(plane_base + co + JDA_COEF_STRIDE);
        jda_chroma2 c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = jda_chroma_terms64((cb >> (8 * j)) & 255u, (cr >> (8 * j)) & 255u);
        uint32_t v[4];
        if (PT == JDA_RGB8888) {
            jda_rgba_pair_half(s01, jda_pack_hi16(c[0].r, c[1].r), jda_pack_hi16(c[0].g, c[1].g), jda_pack_hi16(c[0].b, c[1].b), v[0], v[1]);
            jda_rgba_pair_half(s23, jda_pack_hi16(c[2].r, c[3].r), jda_pack_hi16(c[2].g, c[3].g), jda_pack_hi16(c[2].b, c[3].b), v[2], v[3]);
            if (!CLIP) {
                jda_chunk16_alias q;
                q.w[0] = v[0]; q.w[1] = v[1]; q.w[2] = v[2]; q.w[3] = v[3];
                *(jda_chunk16_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + g * 16u) = q;
            } else jda_store4<PT, true>(out + (size_t)Y * pitch, X, D.out_w, v);
        } else {
            const uint32_t a01 = jda_565_pair_half<PT>(s01, jda_pack_hi16(c[0].r, c[1].r), jda_pack_hi16(c[0].g, c[1].g), jda_pack_hi16(c[0].b, c[1].b));
            const uint32_t a23 = jda_565_pair_half<PT>(s23, jda_pack_hi16(c[2].r, c[3].r), jda_pack_hi16(c[2].g, c[3].g), jda_pack_hi16(c[2].b, c[3].b));
            if (!CLIP) *(jda_u64_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + g * 8u) = (uint64_t)a01 | ((uint64_t)a23 << 32);
            else {
                v[0] = a01 & 0xffffu; v[1] = a01 >> 16; v[2] = a23 & 0xffffu; v[3] = a23 >> 16;
                jda_store4<PT, true>(out + (size_t)Y * pitch, X, D.out_w, v);
            }
        }
    }
}

// Half-size colour output of 4:4:4 (JPEGPutMCU11 half-size body, jpeg.inl:3297-3322): luma = the 2x2 sum << 10 as above, each
// chroma sample = (its 2x2 sum + 2) >> 2.  A work item is one output row of one MCU (4 pixels): rows 2r, 2r + 1 of its Y, Cb
// and Cr blocks.
template <int PT, bool CLIP>
JDA_HD void jda_p4_444_half(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                            uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const uint32_t bpp = PT == JDA_RGB8888 ? 4u : 2u;
    const uint32_t groups = tile_w >> 2;                          // = MCUs in the tile
    const uint32_t inv = jda_recip22(groups);
    uint8_t JDA_GLOBAL *out = JDA_G(uint8_t, D.out);
    const uint32_t pitch = D.out_pitch;
    const uint32_t tile_off = y_base * pitch + x_base * bpp;
    for (uint32_t i = t; i < groups * 4; i += JDA_TILE_THREADS) {
        const uint32_t r = jda_umul24(i, inv) >> 22, g = i - jda_umul24(r, groups);     // output row 0..3, MCU
        const uint32_t X = x_base + g * 4, Y = y_base + r;
        if (CLIP && (Y >= D.out_rows || X >= D.out_w)) continue;
        const jda_u32_alias *ys = (const jda_u32_alias *)(plane_base + jda_umul24(g, plane_stride) + r * 16);
        const jda_u32_alias *bs = (const jda_u32_alias *)((const uint8_t *)ys + JDA_COEF_STRIDE), *rs = (const jda_u32_alias *)((const uint8_t *)ys + 2 * JDA_COEF_STRIDE);
        const uint32_t s01 = jda_pair_sums(ys[0]) + jda_pair_sums(ys[2]), s23 = jda_pair_sums(ys[1]) + jda_pair_sums(ys[3]);
        const uint32_t b01 = ((jda_pair_sums(bs[0]) + jda_pair_sums(bs[2]) + 0x00020002u) >> 2) & 0x00ff00ffu, b23 = ((jda_pair_sums(bs[1]) + jda_pair_sums(bs[3]) + 0x00020002u) >> 2) & 0x00ff00ffu;
        const uint32_t r01 = ((jda_pair_sums(rs[0]) + jda_pair_sums(rs[2]) + 0x00020002u) >> 2) & 0x00ff00ffu, r23 = ((jda_pair_sums(rs[1]) + jda_pair_sums(rs[3]) + 0x00020002u) >> 2) & 0x00ff00ffu;
        jda_chroma2 c[4];
        c[0] = jda_chroma_terms64(b01 & 0xffffu, r01 & 0xffffu); c[1] = jda_chroma_terms64(b01 >> 16, r01 >> 16);
        c[2] = jda_chroma_terms64(b23 & 0xffffu, r23 & 0xffffu); c[3] = jda_chroma_terms64(b23 >> 16, r23 >> 16);
        uint32_t v[4];
        if (PT == JDA_RGB8888) {
            jda_rgba_pair_half(s01, jda_pack_hi16(c[0].r, c[1].r), jda_pack_hi16(c[0].g, c[1].g), jda_pack_hi16(c[0].b, c[1].b), v[0], v[1]);
            jda_rgba_pair_half(s23, jda_pack_hi16(c[2].r, c[3].r), jda_pack_hi16(c[2].g, c[3].g), jda_pack_hi16(c[2].b, c[3].b), v[2], v[3]);
            if (!CLIP) {
                jda_chunk16_alias q;
                q.w[0] = v[0]; q.w[1] = v[1]; q.w[2] = v[2]; q.w[3] = v[3];
                *(jda_chunk16_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + g * 16u) = q;
            } else jda_store4<PT, true>(out + (size_t)Y * pitch, X, D.out_w, v);
        } else {
            const uint32_t a01 = jda_565_pair_half<PT>(s01, jda_pack_hi16(c[0].r, c[1].r), jda_pack_hi16(c[0].g, c[1].g), jda_pack_hi16(c[0].b, c[1].b));
            const uint32_t a23 = jda_565_pair_half<PT>(s23, jda_pack_hi16(c[2].r, c[3].r), jda_pack_hi16(c[2].g, c[3].g), jda_pack_hi16(c[2].b, c[3].b));
            if (!CLIP) *(jda_u64_alias JDA_GLOBAL *)(out + tile_off + jda_umul24(r, pitch) + g * 8u) = (uint64_t)a01 | ((uint64_t)a23 << 32);
            else {
                v[0] = a01 & 0xffffu; v[1] = a01 >> 16; v[2] = a23 & 0xffffu; v[3] = a23 >> 16;
                jda_store4<PT, true>(out + (size_t)Y * pitch, X, D.out_w, v);
            }
        }
    }
}

// pixel type and clipping are decided once per tile (uniform), so the item loops are branch-free
template <int MODE, bool CLIP>
JDA_HD void jda_p4_full_colour(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                               uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    const int pt = D.pixel_type;
    if (MODE == JDA_MODE_420) {
        if (pt == JDA_RGB8888) jda_p4_420_full<JDA_RGB8888, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else if (pt == JDA_RGB565_LITTLE_ENDIAN) jda_p4_420_full<JDA_RGB565_LITTLE_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else jda_p4_420_full<JDA_RGB565_BIG_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
    } else if (MODE == JDA_MODE_422) {
        if (pt == JDA_RGB8888) jda_p4_422_full<JDA_RGB8888, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else if (pt == JDA_RGB565_LITTLE_ENDIAN) jda_p4_422_full<JDA_RGB565_LITTLE_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else jda_p4_422_full<JDA_RGB565_BIG_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
    } else {
        if (pt == JDA_RGB8888) jda_p4_444_full<JDA_RGB8888, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else if (pt == JDA_RGB565_LITTLE_ENDIAN) jda_p4_444_full<JDA_RGB565_LITTLE_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
        else jda_p4_444_full<JDA_RGB565_BIG_ENDIAN, CLIP>(D, t, plane_base, plane_stride, tile_w, x_base, y_base);
    }
}

// everything else (scaled outputs, luma-only, gray JPEGs): generic per-pixel fetch
template <int MODE>
JDA_HD void jda_p4_generic(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                           uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    typedef jda_mode_traits<MODE> T;
    const int shift = D.scale_shift;
    const uint32_t mw_log2 = (uint32_t)T::MCU_W_LOG2 - (uint32_t)shift;       // MCU tile edge in output px = 1 << mw_log2
    const uint32_t mh = (uint32_t)T::MCU_H >> shift;
    const int pt = D.pixel_type;
    const uint32_t groups = (tile_w + 3) >> 2;
    const uint32_t inv = jda_recip22(groups);
    for (uint32_t i = t; i < groups * mh; i += JDA_TILE_THREADS) {
        const uint32_t row = (i * inv) >> 22, x4 = (i - row * groups) * 4;
        const uint32_t Y = y_base + row, X = x_base + x4;
        if (Y >= D.out_rows || X >= D.out_w) continue;
        uint32_t v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t x = x4 + j;
            v[j] = 0;
            if (x < tile_w) {
                const uint32_t m = x >> mw_log2;
                v[j] = jda_output_pixel<MODE>(plane_base + m * plane_stride, x - (m << mw_log2), row, shift, pt);
            }
        }
        jda_store4_rt<true>(JDA_G(uint8_t, D.out) + (size_t)Y * D.out_pitch, X, D.out_w, pt, v, tile_w - x4 < 4u ? tile_w - x4 : 4u);
    }
}

// 1/4 and 1/8 outputs are a few pixels per MCU (a 4:2:0 tile at 1/8: 20 x 2): one pixel per lane instead of the generic
// stage's four -- the wavefront runs a quarter of the per-pixel code --, consecutive lanes store consecutive pixels
template <int MODE>
JDA_HD void jda_p4_thumb(const jda_dev_desc &D, uint32_t t, const uint8_t *plane_base,
                         uint32_t plane_stride, uint32_t tile_w, uint32_t x_base, uint32_t y_base)
{
    typedef jda_mode_traits<MODE> T;
    const int shift = D.scale_shift;
    const uint32_t mw_log2 = (uint32_t)T::MCU_W_LOG2 - (uint32_t)shift;       // MCU tile edge in output px = 1 << mw_log2
    const uint32_t mh = (uint32_t)T::MCU_H >> shift;
    const int pt = D.pixel_type;
    const uint32_t inv = jda_recip22(tile_w);
    for (uint32_t i = t; i < tile_w * mh; i += JDA_TILE_THREADS) {
        const uint32_t row = jda_umul24(i, inv) >> 22, x = i - jda_umul24(row, tile_w);
        const uint32_t Y = y_base + row, X = x_base + x;
        if (Y >= D.out_rows || X >= D.out_w) continue;
        const uint32_t m = x >> mw_log2;
        const uint32_t v = jda_output_pixel<MODE>(plane_base + jda_umul24(m, plane_stride), x - (m << mw_log2), row, shift, pt);
        uint8_t JDA_GLOBAL *rowp = JDA_G(uint8_t, D.out) + (size_t)Y * D.out_pitch;
        if (pt == JDA_RGB8888) ((jda_u32_alias JDA_GLOBAL *)rowp)[X] = v;
        else if (pt == JDA_EIGHT_BIT_GRAYSCALE) rowp[X] = (uint8_t)v;
        else ((uint16_t JDA_GLOBAL *)rowp)[X] = (uint16_t)v;
    }
}

// the precomputed item addresses belong to an image (pitch, pixel size): made when a wavefront meets a new image
template <int MODE>
JDA_HD void jda_p4_prepare(jda_p4_pre &P, const jda_dev_desc &D, uint32_t t)
{
    typedef jda_lds_layout<MODE> L;
    if (MODE == JDA_MODE_420) jda_p4_precompute(P, t, L::PLANE_STRIDE, D.out_pitch, D.pixel_type == JDA_RGB8888 ? 4u : 2u);
    else if (MODE == JDA_MODE_444) jda_p4_precompute_444(P, t, L::PLANE_STRIDE, D.out_pitch, D.pixel_type == JDA_RGB8888 ? 4u : 2u);
    else {
#pragma unroll
        for (int it = 0; it < JDA_P4_PASSES; it++) P.yo[it] = P.co[it] = P.rel[it] = 0;
    }
}

template <int MODE>
JDA_HD void jda_p4_output_at(const jda_dev_desc &D, const jda_strip &S, const jda_tile_ctx &C, uint32_t t, const uint8_t *wl, const jda_p4_pre &P, bool precomputed);
template <int MODE>
JDA_HD void jda_p4_output(const jda_dev_desc &D, const jda_strip &S, const jda_tile_ctx &C, uint32_t t, const uint8_t *wl, const jda_p4_pre &P)
{
    typedef jda_mode_traits<MODE> T;
    if (C.count == 0) return;
    if (D.strip_mcus == 0) { jda_p4_output_at<MODE>(D, S, C, t, wl, P, true); return; }
    // A strip-major surface: the tile lies inside ONE strip (the tile list is cut at strip edges), and inside it the surface is an
    // ordinary one -- base and pitch of the strip, shifted so that the tile's absolute pixel coordinates address it
    const int shift = D.scale_shift;
    const uint32_t mw = (uint32_t)T::MCU_W >> shift, mh = (uint32_t)T::MCU_H >> shift;
    const uint32_t bpp = D.pixel_type == JDA_RGB8888 ? 4u : (D.pixel_type == JDA_EIGHT_BIT_GRAYSCALE ? 1u : 2u);
    const uint32_t n_sx = (D.mcus_x + D.strip_mcus - 1u) / D.strip_mcus, sx = S.mcu_x0 / D.strip_mcus;
    const uint32_t in_strip = D.mcus_x - sx * D.strip_mcus < D.strip_mcus ? D.mcus_x - sx * D.strip_mcus : D.strip_mcus;      // MCUs of this strip (a row's last one may be narrower)
    const uint32_t pitch = in_strip * mw * bpp;
    const size_t strip_bytes = (size_t)D.strip_mcus * mw * mh * bpp;
    jda_dev_desc V = D;
    V.out_pitch = pitch;
    V.out = D.out + ((size_t)S.mcu_y * n_sx + sx) * strip_bytes - ((size_t)S.mcu_y * mh * pitch + (size_t)sx * D.strip_mcus * mw * bpp);
    jda_p4_output_at<MODE>(V, S, C, t, wl, P, false);
}
template <int MODE>
JDA_HD void jda_p4_output_at(const jda_dev_desc &D, const jda_strip &S, const jda_tile_ctx &C, uint32_t t, const uint8_t *wl, const jda_p4_pre &P, bool precomputed)
{
    typedef jda_mode_traits<MODE> T;
    typedef jda_lds_layout<MODE> L;
    if (C.count == 0) return;
    const int shift = D.scale_shift;
    const uint32_t mw = (uint32_t)T::MCU_W >> shift, mh = (uint32_t)T::MCU_H >> shift;   // MCU tile in output px
    const uint32_t tile_w = C.count * mw;                         // output pixels per row of the tile
    const uint32_t x_base = S.mcu_x0 * mw, y_base = S.mcu_y * mh;
    const uint8_t *plane_base = wl + L::PLANE_OFF;
    const bool colour_out = D.pixel_type != JDA_EIGHT_BIT_GRAYSCALE;
    if ((MODE == JDA_MODE_444 || MODE == JDA_MODE_420 || MODE == JDA_MODE_422) && shift == 0 && colour_out) {      // specialised full-size colour paths
        const bool inside = x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows;   // no clipping in this tile
        if (MODE == JDA_MODE_420 && inside && precomputed && C.count == (uint32_t)L::MCUS) {          // the common case: a full tile
            const int pt = D.pixel_type;
            if (pt == JDA_RGB8888) jda_p4_420_full10<JDA_RGB8888>(D, P, plane_base, x_base, y_base);
            else if (pt == JDA_RGB565_LITTLE_ENDIAN) jda_p4_420_full10<JDA_RGB565_LITTLE_ENDIAN>(D, P, plane_base, x_base, y_base);
            else jda_p4_420_full10<JDA_RGB565_BIG_ENDIAN>(D, P, plane_base, x_base, y_base);
            return;
        }
        if (MODE == JDA_MODE_444 && inside && precomputed && C.count == (uint32_t)L::MCUS) {          // a full 4:4:4 tile
            const int pt = D.pixel_type;
            if (pt == JDA_RGB8888) jda_p4_444_full21<JDA_RGB8888>(D, P, t, plane_base, x_base, y_base);
            else if (pt == JDA_RGB565_LITTLE_ENDIAN) jda_p4_444_full21<JDA_RGB565_LITTLE_ENDIAN>(D, P, t, plane_base, x_base, y_base);
            else jda_p4_444_full21<JDA_RGB565_BIG_ENDIAN>(D, P, t, plane_base, x_base, y_base);
            return;
        }
        if (inside) jda_p4_full_colour<MODE, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        else jda_p4_full_colour<MODE, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
    } else if (shift == 0 && !colour_out) {                       // 8-bit gray, full size: the luma samples are the pixels
        const bool inside = x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows;
        if (inside) jda_p4_gray8_full<MODE, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        else jda_p4_gray8_full<MODE, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
    } else if (MODE == JDA_MODE_420 && shift == 1 && colour_out) {      // the most used scaled colour outputs
        const bool inside = x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows;
        const int pt = D.pixel_type;
        if (pt == JDA_RGB8888) {
            if (inside) jda_p4_420_half_rgba<JDA_RGB8888, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_420_half_rgba<JDA_RGB8888, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        } else if (pt == JDA_RGB565_LITTLE_ENDIAN) {
            if (inside) jda_p4_420_half_rgba<JDA_RGB565_LITTLE_ENDIAN, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_420_half_rgba<JDA_RGB565_LITTLE_ENDIAN, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        } else {
            if (inside) jda_p4_420_half_rgba<JDA_RGB565_BIG_ENDIAN, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_420_half_rgba<JDA_RGB565_BIG_ENDIAN, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        }
    } else if (MODE == JDA_MODE_444 && shift == 1 && colour_out) {
        const bool inside = x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows;
        const int pt = D.pixel_type;
        if (pt == JDA_RGB8888) {
            if (inside) jda_p4_444_half<JDA_RGB8888, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_444_half<JDA_RGB8888, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        } else if (pt == JDA_RGB565_LITTLE_ENDIAN) {
            if (inside) jda_p4_444_half<JDA_RGB565_LITTLE_ENDIAN, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_444_half<JDA_RGB565_LITTLE_ENDIAN, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        } else {
            if (inside) jda_p4_444_half<JDA_RGB565_BIG_ENDIAN, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
            else jda_p4_444_half<JDA_RGB565_BIG_ENDIAN, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        }
    } else if (shift == 1 && !colour_out) {                       // 8-bit gray, half size: 2x2 luma sums
        const bool inside = x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows && (x_base & 3u) == 0;
        if (inside) jda_p4_gray8_half<MODE, false>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
        else jda_p4_gray8_half<MODE, true>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
    } else if (MODE == JDA_MODE_GRAY && shift >= 2 && !colour_out && C.count == (uint32_t)L::MCUS &&
               x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows) {
        // 8-bit gray thumbnails of a gray file (1/4: the block's 2x2 samples, 1/8: its one), a full unclipped tile: lane = block,
        // the 64 lanes of one store instruction cover one contiguous piece of an output row
        const uint8_t *blk = plane_base + jda_umul24(t, (uint32_t)L::PLANE_STRIDE);
        uint8_t JDA_GLOBAL *o = JDA_G(uint8_t, D.out) + (y_base * D.out_pitch + x_base);
        if (shift == 3) o[t] = blk[0];
        else {
            *(uint16_t JDA_GLOBAL *)(o + 2u * t) = *(const uint16_t *)blk;
            *(uint16_t JDA_GLOBAL *)(o + D.out_pitch + 2u * t) = *(const uint16_t *)(blk + 2);
        }
    } else if (shift >= 2)
        jda_p4_thumb<MODE>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
    else
        jda_p4_generic<MODE>(D, t, plane_base, L::PLANE_STRIDE, tile_w, x_base, y_base);
}

// ---- 1/4 scale: a kernel of its own (jda_quarter_tiles in jda_kernels.hip; tests/hostsim steps the same functions) ---------------
// At 1/4 a block is the 2x2 IDCT of coefficients 0, 1, 8, 9 (jpeg.inl:2305-2326) -- zigzag positions 0, 1, 2, 4: the reference's
// JPEGDecodeMCU stores nothing behind position 4 (:2117-2119), and with a per-block index nothing behind it has to be read either.
// So a block is its DC value (index format 2), at most FOUR AC symbols -- they lie in the 104 bits behind its index entry -- and
// sixteen multiplies: no window in LDS, no coefficient slot, no work lists.  lane = block; the five dwords of the scan that hold those bits
// come straight from memory (neighbouring lanes read neighbouring bytes) and the reader slides through them in registers.
struct jda_q4_bits { uint32_t d[5]; };      // scan dwords, big-endian (byte-swapped): d[0] holds the last CONSUMED bit (jda_q4_first_dword)
struct jda_q4_quant { int32_t q0, q1, q8, q9; };
// the dword of the scan that holds the bit in front of the block's first AC symbol
JDA_HD uint32_t jda_q4_first_dword(uint32_t ix)
{
    const uint32_t p = (ix >> JDA_INDEX_OFF_BITS) * 8u + (ix & (JDA_INDEX_TRUNC - 1u));      // (>= 2: a DC symbol lies in front of it)
    return (p - 1u) >> 5;
}
JDA_HD uint32_t jda_bswap32(uint32_t v) { return jda_perm(0, v, 0x00010203u); }
JDA_HD jda_q4_bits jda_q4_load(const uint8_t *scan, uint32_t scan_len, uint32_t ix)
{
    // (the scan section is 16-byte aligned and JDA_SCAN_PAD zero bytes longer than the scan: 20 bytes from a decoded block's dword stay
    // inside.  The streamed pipeline launches the decode before the device pre-scan's verdict is read: an image that fails it is decoded
    // again, but its entries may point anywhere meanwhile -- so the dword is clamped to the section)
    uint32_t a = jda_q4_first_dword(ix);
    const uint32_t a_max = (scan_len + JDA_SCAN_PAD - 20u) >> 2;
    a = a < a_max ? a : a_max;
    const jda_u32_alias JDA_GLOBAL *w = JDA_G(const jda_u32_alias, scan) + a;
    jda_q4_bits B;
#pragma unroll
    for (int i = 0; i < 5; i++) B.d[i] = w[i];
    return B;
}
// the block's four samples: row 0 in bytes 0-1, row 1 in bytes 2-3 (jda_idct_2x2's order).  EXACT / trunc: as jda_decode_block_win
template <bool EXACT>
JDA_HD uint32_t jda_q4_block(uint32_t ix, int32_t dc, jda_q4_bits B, const uint16_t *ac, const jda_q4_quant &Q, bool dc_only, bool trunc)
{
    const uint32_t off = ix & (JDA_INDEX_TRUNC - 1u);
    const uint32_t p = (ix >> JDA_INDEX_OFF_BITS) * 8u + off;
    uint32_t cm = (p - 1u) & 31u;                         // bit of d0 consumed last
    uint32_t d0 = jda_bswap32(B.d[0]), d1 = jda_bswap32(B.d[1]), d2 = jda_bswap32(B.d[2]), d3 = jda_bswap32(B.d[3]), d4 = jda_bswap32(B.d[4]);
    uint32_t roff = EXACT ? jda_ref_refill(off) : 0u;    // the reference's ulBitOff at the block's first AC symbol
    int32_t c1 = 0, c8 = 0, c9 = 0;                      // (:2118)
    // Four trips without a branch: every symbol moves k on by one at least, so four reach the limit of :2117-2119 (k = 5); a lane whose
    // block ended earlier (EOB, or a symbol that went past the limit) keeps looking symbols up -- in bits that are not its block's any more,
    // harmless: k >= 5 stores nothing -- instead of leaving the loop, because every way out of it is an exec-mask region on the dependent
    // chain (peek -> LUT -> position) and the wavefront runs as many trips as its longest block anyway.
    uint32_t k = dc_only ? 8u : 1u;
#pragma unroll
    for (int trip = 0; trip < 4; trip++) {               // (:2223-2265)
        const uint32_t w = jda_alignbit(d0, d1, 31u - cm);
        const uint32_t e = ac[jda_bfe(w, w >= 0xfc000000u ? 16u : 22u, 11u)];
        const uint32_t len = (e >> 12) + 1u, ms = (e >> 8) & 0xfu;
        k = JDA_AC_STOPS(e) ? 8u : k + ((e >> 1) & 0xfu);    // EOB ends the block; else the run
        uint32_t m = w << len;
        const uint32_t n = len + ms;
        if (EXACT) {
            roff += n;                                   // the reference's ulBitOff after its magnitude read (:2249-2252)
            if (roff > 64u && trunc) m &= ~(0xffffffffu >> ((64u + ms - roff) & 31u));      // its window ended inside the magnitude
            roff = jda_ref_refill(roff);
        }
        const int32_t v = (int16_t)jda_extend_top(m, ms);    // (a symbol without a value is ZRL: it goes past the limit, nothing is stored)
        c1 = k == 1u ? v : c1; c8 = k == 2u ? v : c8; c9 = k == 4u ? v : c9;
        k++;
        if (trip < 3) {
            cm += n;
            if (cm >= 32u) { d0 = d1; d1 = d2; d2 = d3; d3 = d4; cm -= 32u; }      // (a symbol is 26 bits at most: one step)
        }
    }
    const int32_t a = (int32_t)(int16_t)dc * Q.q0, b = c8 * Q.q8, c = c1 * Q.q1, d = c9 * Q.q9;
    const int32_t t0 = a + b, t2 = a - b, t1 = c + d, t3 = c - d;
    return jda_range_limit5(t0 + t1) | (jda_range_limit5(t0 - t1) << 8) | (jda_range_limit5(t2 + t3) << 16) | (jda_range_limit5(t2 - t3) << 24);
}
// what a lane does with a tile's samples: output pixel i of the tile's (count * MCU_W / 4) x (MCU_H / 4) pixels, row-major, in passes
// of 64; a pixel's samples come from the lanes that decoded its blocks (all: every lane's block samples; GPU: ds_bpermute, so every
// lane runs every pass).  Pixels as JPEGPutMCU* make them at iScaleShift 2 (jda_fetch's shift == 2 cases).
template <int MODE>
JDA_HD void jda_q4_store(const jda_dev_desc &D, const jda_strip &S, uint32_t count, uint32_t lane, uint32_t px, const uint32_t *all)
{
    typedef jda_mode_traits<MODE> T;
    const uint32_t mw = (uint32_t)T::MCU_W >> 2, mh = (uint32_t)T::MCU_H >> 2;          // an MCU's pixels: 2 x 2, 4 x 4, 4 x 2, 2 x 4
    const uint32_t tile_w = count * mw, n_px = tile_w * mh;
    const uint32_t x_base = S.mcu_x0 * mw, y_base = S.mcu_y * mh;
    const int pt = D.pixel_type;
    if (MODE == JDA_MODE_GRAY && pt == JDA_EIGHT_BIT_GRAYSCALE && count == 64u && (x_base & 3u) == 0u && x_base + tile_w <= D.out_w && y_base + mh <= D.out_rows) {
        // a whole gray tile to 8-bit gray: two lanes share their blocks' rows -- the even one stores row 0 of both, the odd one row 1
        const uint32_t nb = jda_lane_pull(px, lane ^ 1u, all);
        const uint32_t v = (lane & 1u) ? jda_perm(px, nb, 0x07060302u) : jda_perm(nb, px, 0x05040100u);
        uint8_t JDA_GLOBAL *o = JDA_G(uint8_t, D.out) + ((size_t)(y_base + (lane & 1u)) * D.out_pitch + x_base + 2u * (lane & ~1u));
        *(jda_u32_alias JDA_GLOBAL *)o = v;
        return;
    }
    for (uint32_t i0 = 0; i0 < n_px; i0 += JDA_TILE_THREADS) {
        const uint32_t i = i0 + lane;
        uint32_t row = i >= tile_w ? 1u : 0u;
        if (mh == 4u) row += (i >= 2u * tile_w ? 1u : 0u) + (i >= 3u * tile_w ? 1u : 0u);
        const uint32_t x = i - jda_umul24(row, tile_w);
        const uint32_t m = mw == 4u ? x >> 2 : x >> 1, ax = x & (mw - 1u);
        // the luma block of the MCU and the sample in it; where the chroma sample lies in its block
        uint32_t q = 0, ys, cs;
        if (MODE == JDA_MODE_420) { q = (row >> 1) * 2u + (ax >> 1); ys = (row & 1u) * 2u + (ax & 1u); cs = q; }                   // :3664-3748
        else if (MODE == JDA_MODE_422) { q = ax >> 1; ys = row * 2u + (ax & 1u); cs = row * 2u + q; }                               // :4789-4838
        else if (MODE == JDA_MODE_440) { q = row >> 1; ys = (row & 1u) * 2u + ax; cs = q * 2u + ax; }                               // :4610-4682
        else { ys = row * 2u + ax; cs = ys; }
        const uint32_t yl = (m * (uint32_t)T::NBLK + q) & 63u, cl = (m * (uint32_t)T::NBLK + (uint32_t)T::NLUMA) & 63u;
        const uint32_t yw = jda_lane_pull(px, yl, all);
        uint32_t cbw = 0, crw = 0;
        if (MODE != JDA_MODE_GRAY) { cbw = jda_lane_pull(px, cl, all); crw = jda_lane_pull(px, (cl + 1u) & 63u, all); }
        const uint32_t X = x_base + x, Y = y_base + row;
        if (i >= n_px || X >= D.out_w || Y >= D.out_rows) continue;
        const uint32_t y = (yw >> (8u * ys)) & 0xffu;
        uint8_t JDA_GLOBAL *rowp = JDA_G(uint8_t, D.out) + (size_t)Y * D.out_pitch;
        if (pt == JDA_EIGHT_BIT_GRAYSCALE) rowp[X] = (uint8_t)y;
        else if (MODE == JDA_MODE_GRAY) ((uint16_t JDA_GLOBAL *)rowp)[X] = (uint16_t)jda_gray_565(y, pt != JDA_RGB565_LITTLE_ENDIAN);      // JPEGPutMCUGray
        else {
            jda_ycc c;
            c.y = (int32_t)(y << 12); c.cb = (int32_t)((cbw >> (8u * cs)) & 0xffu); c.cr = (int32_t)((crw >> (8u * cs)) & 0xffu);
            if (pt == JDA_RGB8888) ((jda_u32_alias JDA_GLOBAL *)rowp)[X] = jda_pixel_rgba(c);
            else ((uint16_t JDA_GLOBAL *)rowp)[X] = (uint16_t)jda_pixel_565(c, pt == JDA_RGB565_BIG_ENDIAN);
        }
    }
}

#endif // JDA_DEVICE_CORE_H
