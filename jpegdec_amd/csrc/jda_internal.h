// jda_internal.h -- structures shared by the host front end, the device runtime and the kernels.
#ifndef JDA_INTERNAL_H
#define JDA_INTERNAL_H

#include <stdint.h>

#include "../../include/jpegdec_amd.h"

// ---- table blob uploaded per image (16-byte aligned, one contiguous copy into LDS) ----
// layout mirrors what the reference keeps in JPEGIMAGE (src/JPEGDEC.h:235-238) so that the
// kernels index the LUTs exactly like JPEGDecodeMCU does (jpeg.inl:2129-2136, 2231-2236).
#define JDA_TB_DC        0        // 2 x 1024 bytes   (ucHuffDC)
#define JDA_TB_AC        2048     // 2 x 2048 uint16  (usHuffAC)
#define JDA_TB_QUANT     10240    // 4 x 64 int16     (sQuantTable after JPEGFixQuantD)
#define JDA_TB_ZIGZAG    10752    // 64 bytes         (cZigZag2: zigzag position -> natural index)
#define JDA_TB_EOB       10816    // 2 x uint32       (ours: the end-of-block code of each AC table, (32 - length) << 16 | code,
                                  //                   so that P1 recognises EOB by comparing stream bits; JDA_EOB_NONE = no such code)
#define JDA_EOB_NONE     ((31u << 16) | 2u)      // a one-bit field never reads 2
#define JDA_TABLE_BYTES  10832

#define JDA_SCAN_PAD     32       // zero bytes after the filtered scan (window loads overrun)
#define JDA_INDEX_OFF_BITS 7      // index entry (format 2) = (byte position << 7) | flags/bit offset: the reference's bit reader at the block's
                                  // FIRST AC SYMBOL, behind the refill at the top of its AC loop (jpeg.inl:2225-2230; bit offset 0..47 in bits
                                  // 5:0), bit 6 = JDA_INDEX_TRUNC; the block's own DC value rides beside it (blk_dc).  (The closing entry behind
                                  // the last block bounds the scan: the reader as the last block left it, offset 0..64, no flag -- or, from the
                                  // device pre-scan, up to 41 bits further on.)
#define JDA_INDEX_TRUNC 0x40u     // the reference truncates a magnitude read of this block: P1 must follow its ulBitOff
// Continuation entries (optional part of the index): a long block can be entered every JDA_CONT_SYMS AC symbols, so that several lanes
// of P1 share it -- the wavefront runs as long as its longest CHUNK, not its longest block (photographs: luma blocks of 40 symbols beside
// chroma blocks of 4).  blk_cont_first[g] .. blk_cont_first[g + 1] are block g's entries in blk_cont; an entry: the bit position of
// its first symbol relative to the block's first AC symbol (bits 11:0), the zigzag index of its first coefficient (bits 17:12), the low
// seven bits of the block's ordinal (bits 24:18: which lane of a tile owns it).  A block flagged JDA_INDEX_TRUNC is decoded whole.
#define JDA_CONT_SYMS 8u
#define JDA_CONT_ENTRY(rel, k, g) ((uint32_t)(rel) | ((uint32_t)(k) << 12) | (((uint32_t)(g) & 127u) << 18))
#define JDA_CONT_REL(e) ((e) & 0xfffu)
#define JDA_CONT_K(e) (((e) >> 12) & 63u)
#define JDA_CONT_G7(e) (((e) >> 18) & 127u)

// ---- the segment walk's 11-bit table key (jda_device_core.h, JDA_WT_*) and the reference's DC LUT index, shared with the front end
#if defined(__HIPCC__)
#define JDA_HD_INLINE __host__ __device__ static inline
#else
#define JDA_HD_INLINE static inline
#endif
// the reference's DC LUT index for the 12 stream bits code12 (jpeg.inl:2129-2136: short half by the top 6 bits; codes 11111..: 128 + the low 7 bits)
JDA_HD_INLINE uint32_t jda_dc_lut_index(uint32_t code12) { return code12 >= 0xf80u ? (code12 & 0xffu) : (code12 >> 6); }
// the 12 stream bits an 11-bit walk key stands for (low: the two bits a short key does not have)
JDA_HD_INLINE uint32_t jda_walk_key_code12(uint32_t key, uint32_t low) { return key < 1024u ? ((key << 2) | low) : (0xfc0u | ((key - 1024u) >> 4)); }
// Can the walk's DC tables stand for DC LUT t of the blob?  Only a short key of the form 111110xxxx leaves bits the reference
// looks at (it takes 12) undetermined: it must not matter what they are.
static inline int jda_dc_lut_walkable(const uint8_t *tables, uint32_t t)
{
    const uint8_t *dc = tables + JDA_TB_DC + t * 1024u;
    for (uint32_t key = 992u; key < 1008u; key++) {
        const uint32_t i0 = jda_dc_lut_index(jda_walk_key_code12(key, 0u));
        for (uint32_t low = 1; low < 4u; low++) {
            const uint32_t i = jda_dc_lut_index(jda_walk_key_code12(key, low));
            if (dc[i] != dc[i0] || dc[i + 512u] != dc[i0 + 512u]) return 0;
        }
    }
    return 1;
}

// Measuring switches (environment variables that change what a build does) exist only in laboratory builds: make lib EXTRA=-DJDA_LAB.
// The product library reads JPEGDEC_AMD_DEVICE (which GPU the drop-in class uses) and nothing else.
#ifdef JDA_LAB
#include <stdlib.h>
#define JDA_LAB_ENV(name) getenv(name)
#else
#define JDA_LAB_ENV(name) ((const char *)0)
#endif

// kinds of MCU the kernels are specialised for
enum { JDA_MODE_GRAY = 0, JDA_MODE_444 = 1, JDA_MODE_420 = 2, JDA_MODE_422 = 3 /* h2v1, MCU 16x8 */, JDA_MODE_440 = 4 /* h1v2, MCU 8x16 */, JDA_N_MODES = 5 };

// Pointers stored in descriptors are loaded from memory, so the compiler cannot know they point to
// global memory and would emit slow generic (flat_*) accesses; device code casts them with JDA_G().
#if defined(__HIP_DEVICE_COMPILE__)
#define JDA_GLOBAL __attribute__((address_space(1)))
#else
#define JDA_GLOBAL
#endif
#define JDA_G(T, p) ((T JDA_GLOBAL *)(p))

// ---- device-side descriptors ----
struct jda_dev_desc {             // one per image of a batch, 104 bytes
    const uint8_t *scan;          // filtered entropy-coded bytes (4-byte aligned, padded)
    const uint32_t *blk_index;    // n_blocks+1 entries: (byte position << 7) | bit offset at each block's first AC symbol
    const int16_t *blk_dc;        // n_blocks: the block's own DC value
    const uint8_t *tables;        // JDA_TABLE_BYTES
    uint8_t *out;                 // output surface
    const uint32_t *blk_cont_first;   // continuation entries (JDA_CONT_*; NULL: the image has none / is decoded block by block): n_blocks + 1 offsets ..
    const uint32_t *blk_cont;         // .. into the entries
    uint32_t out_pitch;           // bytes
    uint32_t out_w, out_rows;     // clip in output pixels / rows
    uint32_t mcus_x, mcus_y;
    uint32_t n_mcus_ok;           // MCUs the pre-scan validated (others are not decoded)
    uint32_t scan_len;
    uint32_t strip_mcus;          // != 0: the surface is STRIP-MAJOR -- the JPEGDRAW strips of the reference's callback (jpeg.inl:5300-5336), strip_mcus MCUs
                                  // wide and one MCU row high, in raster order, each strip's pixels contiguous (pitch = the strip's own width), a strip
                                  // every strip_mcus x MCU width x MCU height x bytes per pixel; out_pitch then only bounds the surface
    union {                       // 16 bytes of small fields; the kernels carry them as four dwords in SGPRs (cfg)
        struct {
            uint8_t mode;                 // JDA_MODE_*
            uint8_t ncomp;
            uint8_t pixel_type;           // JDA_RGB565_* / RGB8888 / EIGHT_BIT_GRAYSCALE (after LUMA_ONLY folding)
            uint8_t scale_shift;          // 0..3
            uint8_t dc_id[3], ac_id[3], q_id[3];
            uint8_t gray_from_color;      // colour JPEG decoded to GRAY8: chroma blocks are not decoded
            uint8_t fast_mul;             // 1: every IDCT multiply operand fits 24 bits (host-checked bound)
            uint8_t pad_[1];
        };
        uint32_t cfg[4];
    };
};

// descriptor byte pad_[0]: bits 1:0 profiling switches, bit 2 below, bit 3 = the scan holds DC symbols only (the first scan of a
// progressive file, decoded as a thumbnail: jpeg.inl:4964-4966), bits 7:4 = Al, the point transform of those DC differences
#define JDA_DESC_DC_ONLY 8u
// bit 2 = P1 must take its general bit reader: an AC table codes the end-of-block symbol more than once (a malformed but
// decodable DHT), so one compare of stream bits cannot recognise EOB
#define JDA_DESC_GENERAL_P1 4u

struct jda_strip {                // one wavefront's tile: <= 64 consecutive blocks (10/21/64 MCUs) of one MCU row; 16 bytes
    uint32_t image;               // index into the descriptor array
    uint16_t mcu_y;               // (a JPEG has at most 65535 / 8 MCU rows and columns)
    uint16_t mcu_x0;
    uint8_t count;                // MCUs in the tile (0 = padding entry)
    uint8_t first;                // 1: the first tile of its image (the wavefront that draws it restages the tables)
    uint16_t pad_;
    uint32_t ord;                 // which set of tables, counted along this launch's tile list (0, 1, 2, ..): the image's own position in the list unless
                                  // consecutive images share their tables (jda_pipeline: the workgroups then pass from one to the next without restaging)
};

// device marker / stuffing filter (jda_filter_scan), one per image
struct jda_filter_params {
    const uint8_t *raw;          // unfiltered entropy-coded segment (from the first SOS payload byte to the end of the file)
    uint8_t *out;                // filtered scan (zero-initialised by the caller: the padding behind it stays zero)
    uint32_t *restart_pos;       // [0] = 0, then the filtered offset at which each RSTn marker stood (first restart_cap entries)
    uint32_t *result;            // [0] filtered length, [1] number of RSTn markers
    uint32_t raw_len, restart_cap;   // raw_len: bytes from `raw` on, the skipped ones included
    uint32_t *work;              // JDA_FILTER_WORK_BYTES(raw_len): the chunks' transition functions and entry values
    uint32_t raw_skip;           // 0..15 bytes at `raw` that are not the stream's: `raw` is 16-byte aligned, a file that the copy engine took from where the
    uint32_t pad_;               // caller had it (JDA_SUBMIT_PINNED_INPUT) keeps its alignment -- its first byte is raw[raw_skip]
};
#define JDA_FILTER_WORK_BYTES(raw_len) (((size_t)(raw_len) / 16384u + 1u) * 20u)


// The tile list of a whole image, made where it is used (jda_pipeline: the lists of a batch of 64 x 4096x4096 are 6.8 MB -- written by
// the host and carried over the bus they were 6 % of a batch's traffic): one of these per image, the kernel writes the records
struct jda_strips_params {
    jda_strip *dst;              // the image's place in its launch list (n_padded records)
    uint32_t n_padded;           // tiles, padded to whole workgroups with empty records
    uint32_t image, ord;
    uint32_t mcus_x, mcus_y, per;   // per: MCUs in a tile (jda_mcus_per_tile)
};

// Slots a 256-byte segment's block records need (RECORD mode of the device pre-scan, jda_device_core.h): more than its 2,048 bits
// can start blocks.  mcu_min_bits: the fewest bits an MCU of the image can take -- per block the shortest DC code + the shorter of
// the EOB code and four of the shortest AC codes (63 coefficients take at least four symbols) -- from the DHT segments
// (jda_frontend.cpp).  A multiple of four (records are stored in 16-byte groups).
#ifndef JDA_SEG_BYTES
#define JDA_SEG_BYTES 256u           // the device pre-scan's segment: one lane walks one (jda_device_core.h)
#endif
static inline uint32_t jda_record_cap(uint32_t mcu_min_bits, uint32_t nblocks)
{
    if (mcu_min_bits < 2u * nblocks) mcu_min_bits = 2u * nblocks;
    return (((JDA_SEG_BYTES * 8u) / mcu_min_bits + 2u) * nblocks + 4u + 3u) & ~3u;
}

// What the host makes of one file when the GPU does everything else (jda_pipeline): see jda_front_prepare in jda_frontend.cpp
struct jda_front {
    jda_image_info info;
    uint8_t dc_id[3], ac_id[3], q_id[3];
    uint8_t general_p1;          // JDA_DESC_GENERAL_P1
    uint8_t progressive;
    uint8_t device_ok;           // filter + pre-scan can run on the device (else: the serial host path)
    uint8_t fast_provable;       // the 24-bit-multiply bound holds for every legal stream with these quantisers
    uint32_t raw_off, raw_len;   // the entropy-coded segment inside the file (unfiltered)
    uint32_t n_intervals;        // restart intervals (0: no DRI)
    uint32_t rec_cap;            // jda_record_cap of the image's tables
};

#endif
