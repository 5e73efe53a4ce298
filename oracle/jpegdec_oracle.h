/* oracle/jpegdec_oracle.h -- CPU restatement of the reference decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker -- never as the thing measured or shipped.
 * The product (libjpegdec_amd.so) neither links nor loads it.
 *
 * Parity status: PINNED.  The reference holds no golden pixel vectors, so the
 * restatement is pinned against outputs of the reference itself: oracle/_ref (the unmodified
 * reference compiled with -DNO_SIMD by oracle/Makefile) in tests/test_oracle_vs_ref.py, and
 * against the hashes of those outputs committed under tests/golden/ (generator:
 * tests/golden/make_golden.py) so the pin also holds where /root/reference is absent.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root; jpeg.inl = src/jpeg.inl).
 */
#ifndef JPEGDEC_ORACLE_H
#define JPEGDEC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* values = the reference's public constants (src/JPEGDEC.h:68-75, 102-126) */
enum { ORC_RGB565_LE = 0, ORC_RGB565_BE = 1, ORC_RGB8888 = 2, ORC_GRAY8 = 3 };
enum { ORC_SCALE_HALF = 2, ORC_SCALE_QUARTER = 4, ORC_SCALE_EIGHTH = 8, ORC_LUMA_ONLY = 64 };
enum { ORC_SUCCESS = 0, ORC_INVALID_PARAMETER, ORC_DECODE_ERROR, ORC_UNSUPPORTED_FEATURE,
       ORC_INVALID_FILE, ORC_ERROR_MEMORY };

typedef struct orc_info {
    int width, height;
    int ncomp;
    int subsample;        /* 0x00 gray, 0x11, 0x12, 0x21, 0x22 (Y sampling byte; jpeg.inl:1698-1713) */
    int mode;             /* 0xc0 baseline / 0xc2 progressive */
    int restart_interval;
    int quant_id[4], dc_id[4], ac_id[4];
    int scan_offset;      /* byte offset of the first entropy-coded byte */
    int error;            /* ORC_* */
    int scan_start, scan_end, approx;   /* Ss, Se, Ah << 4 | Al of the first scan (jpeg.inl:1416-1420) */
} orc_info;

/* header parse (jpeg.inl:1572-1785).  returns 1 ok / 0 fail (error in info->error) */
int orc_get_info(const uint8_t *data, int len, orc_info *info);

/* whole-scan marker/stuffing filter (jpeg.inl:1431-1540); out must hold len bytes; returns out length */
int orc_filter(const uint8_t *in, int len, uint8_t *out);

/* expanded Huffman LUTs exactly as the reference lays them out (jpeg.inl:1066-1275):
 * dc: 2 x 1024 bytes, ac: 2 x 2048 uint16.  returns 1 ok / 0 unsupported */
int orc_huff_tables(const uint8_t *data, int len, uint8_t *dc, uint16_t *ac);

/* prescaled quant tables (jpeg.inl:1789-1811): 4 x 64 int16, natural order */
int orc_quant_tables(const uint8_t *data, int len, int16_t *q);

/* entropy stage only (jpeg.inl:2090-2274 driven as in :5109-5353): for every block in scan
 * order writes 64 int16 coefficients (natural order) and the reference's u16MCUFlags.
 * Also optionally records the bit-reader phase at each MCU start: mcu_state[2*i+0]=byte position
 * in the filtered stream, [2*i+1]=bit offset (0..64), and mcu_dcpred[3*i..] the three predictors.
 * blk_state / blk_dcpred (optional) record the same per BLOCK: the reader phase on entry to
 * JPEGDecodeMCU and that block's DC predictor before it.
 * returns number of blocks decoded (<0: error). */
int orc_entropy(const uint8_t *data, int len, int options, int max_blocks,
                int16_t *coefs, uint16_t *flags, uint32_t *mcu_state, int32_t *mcu_dcpred,
                uint32_t *blk_state, int32_t *blk_dcpred);

/* one block: dequant + IDCT + range limit (jpeg.inl:2278-2326, 2553-2797 and the DC-only
 * bypass :5146-5154).  pred = running DC predictor (int), coef[0] must hold (int16)pred.
 * Writes 64 (full/half), or 4 (quarter/eighth) bytes to out. */
void orc_block_pixels(const int16_t *coef, uint16_t flags, int pred, const int16_t *q,
                      int options, uint8_t *out);

/* full decode into an MCU-padded canvas (= what the reference's draw callbacks assemble, or
 * its framebuffer mode when the width is an MCU multiple).  pitch_bytes >= padded width * bpp,
 * rows >= padded height.  returns 1 ok / 0 fail; *error gets ORC_*. */
int orc_decode(const uint8_t *data, int len, int pixel_type, int options,
               uint8_t *canvas, int pitch_bytes, int rows, int *error);

/* draw-callback plan (jpeg.inl:5062-5084, 5300-5336): fills rects[6*i] = x,y,iWidth,iHeight,
 * iWidthUsed,iBpp for each JPEGDRAW the reference would issue; returns the count. */
int orc_draw_plan(const uint8_t *data, int len, int pixel_type, int options, int max_mcus,
                  int uses_dma, int *rects, int max_rects);

/* closed-form tables the reference ships as literals (jpeg.inl:159-555) -- exposed so tests can
 * compare them with the reference's arrays where the reference tree is available */
uint8_t orc_range_limit(int idx10);               /* ucRangeTable[idx & 0x3ff] */
uint16_t orc_range565(int comp, int idx10);       /* comp 0=R 1=G 2=B : usRangeTableR/G/B */
uint16_t orc_gray565(int y);                      /* usGrayTo565[y] */
int orc_aan_scale(int n);                         /* iScaleBits[n] */
int orc_zigzag_to_natural(int k);                 /* cZigZag2[k] */

#ifdef __cplusplus
}
#endif
#endif
