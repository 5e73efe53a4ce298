/* include/jpegdec_amd.h -- C-ABI of the MI355X-native baseline-JPEG decode path.
 *
 * This is the drop-in boundary for ONE hot path of bitbank2/JPEGDEC: the per-MCU loop of
 * DecodeJPEG (reference src/jpeg.inl:5109-5353: JPEGDecodeMCU -> JPEGIDCT -> JPEGPutMCU*).
 * Plain C types only: no C++, no exceptions, no torch types cross this boundary.  The reference
 * API (openRAM/openFLASH/decode()/JPEG_DRAW_CALLBACK, include/JPEGDEC.h here) is implemented on
 * top of these entry points; INTEGRATION.md shows the binding a maintainer of the reference
 * would add at jpeg.inl:5109.
 *
 * Stages and the reference code each entry point replaces (paths relative to the reference):
 *   jda_parse        host   JPEGParseInfo + JPEGGetSOS + JPEGGetHuffTables   src/jpeg.inl:1572-1785, 1378-1425, 837-873
 *   jda_prepare      host   JPEGMakeHuffTables, JPEGFilter over the whole scan, JPEGFixQuantD and the serial
 *                           entropy pre-scan (JPEGDecodeMCU in skip mode)    src/jpeg.inl:1066-1275, 1431-1540, 1789-1811, 2090-2274
 *   jda_upload       H2D    (no reference equivalent: the reference streams 2 KiB at a time, :1544-1566)
 *   jda_batch_decode GPU    the MCU loops of DecodeJPEG for every image of a batch   src/jpeg.inl:5109-5353
 *                           = JPEGDecodeMCU :2090-2274, JPEGIDCT :2278-2798 (+ DC-only bypass :5146-5154),
 *                             JPEGPutMCU8BitGray/Gray/11/22 :2799-4544, JPEGPixel* :3101-3278
 *
 * Conventions: functions returning int return JDA_SUCCESS (0) or one of the JDA_* error codes,
 * whose values equal the reference's enum (src/JPEGDEC.h:119-126) so getLastError() can pass
 * them through.  Every GPU entry point fails with JDA_ERROR_NO_DEVICE when no HIP device is
 * usable -- there is no CPU fallback in this library.
 */
#ifndef JPEGDEC_AMD_H
#define JPEGDEC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JDA_ABI_VERSION 1

/* error codes: 0..5 are the reference's (src/JPEGDEC.h:119-126) */
enum {
    JDA_SUCCESS = 0,
    JDA_INVALID_PARAMETER = 1,
    JDA_DECODE_ERROR = 2,
    JDA_UNSUPPORTED_FEATURE = 3,
    JDA_INVALID_FILE = 4,
    JDA_ERROR_MEMORY = 5,
    JDA_ERROR_NO_DEVICE = 6,   /* no usable HIP device / runtime error (detail: jda_last_hip_error) */
    JDA_ERROR_HIP = 7
};

/* pixel types: the reference's enum (src/JPEGDEC.h:102-111) */
enum {
    JDA_RGB565_LITTLE_ENDIAN = 0,
    JDA_RGB565_BIG_ENDIAN = 1,
    JDA_RGB8888 = 2,            /* memory order R,G,B,A as the reference's scalar path (jpeg.inl:3162-3175) */
    JDA_EIGHT_BIT_GRAYSCALE = 3
};

/* decode options: the reference's bits (src/JPEGDEC.h:68-75) */
enum {
    JDA_SCALE_HALF = 2,
    JDA_SCALE_QUARTER = 4,
    JDA_SCALE_EIGHTH = 8,
    JDA_LUMA_ONLY = 64
};

/* ------------------------------------------------------------------ host front end */

typedef struct jda_image_info {
    int32_t width, height;        /* SOF0 (jpeg.inl:1684-1685) */
    int32_t ncomp;                /* 1 or 3 */
    int32_t subsample;            /* reference ucSubSample: 0x00 gray, 0x11, 0x12, 0x21, 0x22 (jpeg.inl:1698-1713) */
    int32_t bpp;                  /* bits per sample * ncomp (jpeg.inl:1687) */
    int32_t jpeg_type;            /* 0 baseline, 1 progressive (src/JPEGDEC.h:94-99) */
    int32_t restart_interval;     /* DRI (jpeg.inl:1715-1718) */
    int32_t orientation;          /* EXIF tag 274 if present else 0 */
    int32_t mcu_w, mcu_h;         /* MCU size in source pixels */
    int32_t mcus_x, mcus_y;       /* MCU grid (jpeg.inl:5013-5037) */
    int32_t scan_offset;          /* byte offset of the entropy-coded data */
    int32_t blocks_per_mcu;       /* 1 gray, 3 for 4:4:4, 4 for 4:2:2 / 4:4:0, 6 for 4:2:0 (Y.. Cb Cr in scan order) */
    int32_t has_thumb;            /* EXIF IFD1 present (jpeg.inl:1667-1675) */
    int32_t thumb_w, thumb_h;     /* EXIF tags 256 / 257 (0 when absent, as in the reference) */
    int32_t thumb_offset;         /* file offset of the embedded thumbnail JPEG (tag 513 + TIFF base) */
    int32_t scan_start, scan_end; /* Ss, Se of the first scan (jpeg.inl:1416-1417) */
    int32_t approx;               /* Ah << 4 | Al of the first scan (jpeg.inl:1418-1420) */
} jda_image_info;

/* The option bits an image is really decoded with: a progressive file (jpeg_type 1) is decoded from its first (DC)
 * scan only, as a 1/8 thumbnail -- JPEG_SCALE_EIGHTH is OR-ed in (jpeg.inl:4964-4966) before the HALF / QUARTER /
 * EIGHTH chain (:4978-4990) picks the first bit that is set. */
int32_t jda_effective_options(const jda_image_info *info, int32_t options);

/* Header parse only.  Accept/reject rules follow JPEGParseInfo (jpeg.inl:1572-1785). */
int jda_parse(const uint8_t *jpeg, int32_t len, jda_image_info *info);

/* An image made ready for the GPU (host memory): expanded Huffman LUTs, prescaled quant tables,
 * the filtered scan and the per-block index produced by the serial pre-scan. */
typedef struct jda_image jda_image;

/* Parse + table build + filter + pre-scan.  `options` are the JDA_SCALE_* / JDA_LUMA_ONLY bits the
 * image will be decoded with (they do not change the index; they are validated here).
 * Returns NULL on failure with *err set.  The JPEG buffer is not referenced after return. */
jda_image *jda_prepare(const uint8_t *jpeg, int32_t len, int32_t *err);

/* The same with options.  JDA_PREPARE_DEVICE_PRESCAN: skip the serial Huffman pre-scan on the host; the per-block
 * index is then made on the GPU when the image is uploaded (jda_upload / jda_upload_batch) by the segment walk --
 * one lane per 256-byte segment of the filtered scan, with or without restart markers (DRI, jpeg.inl:1715-1718,
 * 5337-5348) -- the same index as the serial pre-scan's in the sense of jda_index_equivalent.  The upload falls back to the host pre-scan by itself
 * when the walk cannot guarantee that (an invalid code, a marker out of place, states that do not settle); files the
 * walk cannot take at all (progressive, one restart interval, table ids 2-3, DC codes its table key cannot tell
 * apart) are pre-scanned here whatever the flag says: jda_image_prescan_pending tells. */
#define JDA_PREPARE_DEVICE_PRESCAN 1
/* Continuation entries (jda_image_block_cont) for every image the serial pre-scan indexes / for none; default: for the images in the
 * window of bits per block in which the decode kernel's chunked entropy phase was measured to pay (56 .. 112). */
#define JDA_PREPARE_CONT_ALWAYS 2
#define JDA_PREPARE_CONT_NEVER 4
/* The host pre-scan of a stream with restart intervals (>= 4 of them, >= 12 KB of scan) decodes its intervals side by side on a few
 * helper threads the library keeps (an interval starts at a marker with predictors zero; the reader's phase across intervals -- the
 * one thing that carries over, SURVEY fact 6 -- is settled afterwards); a stream without them (>= 16 KB of scan) is walked in chunks
 * from a guess, the true path spliced in front of where each walker fell into step.  Either way the same index as the serial
 * pre-scan's in the sense of jda_index_equivalent.  This flag keeps the pre-scan on the calling thread. */
#define JDA_PREPARE_SERIAL_PRESCAN 8
/* .. and this one takes the helper threads for every stream that admits it, whatever its size (the library's own size limits are where
 * the threads were measured to pay; tests use it on small files). */
#define JDA_PREPARE_PARALLEL_PRESCAN 16
jda_image *jda_prepare_ex(const uint8_t *jpeg, int32_t len, int32_t flags, int32_t *err);
/* The helper threads (process-wide): n < 0 as many as the library chooses (the default: up to five, none on fewer than four usable CPUs;
 * made at the first image that takes them, asleep unless one caller decodes image after image within 2 ms of each other), n == 0 none --
 * no thread is made, every pre-scan runs on its caller's thread --, n > 0 at most n in a job.  Returns the setting it replaces.
 * jda_prepare_batch's workers (threads > 1) and the pipeline's workers never use them. */
int jda_set_host_prescan_helpers(int32_t n);
/* jda_prepare_ex for n images on `threads` host threads (<= 0: as many as the process may keep busy -- hardware threads, its affinity mask, a cgroup quota); out[i] / errs[i] per image
 * (errs may be NULL).  Returns JDA_SUCCESS or the first error met. */
int jda_prepare_batch(int32_t n, const uint8_t *const *jpegs, const int32_t *lens, int32_t flags, int32_t threads,
                      jda_image **out, int32_t *errs);
/* 1 while the image's block index has not been made yet (deferred to jda_upload). */
int jda_image_prescan_pending(const jda_image *img);
void jda_image_free(jda_image *img);

const jda_image_info *jda_image_get_info(const jda_image *img);
/* views into the prepared image (owned by img): the filtered scan; the per-BLOCK index, FORMAT 2
 * (n_blocks+1 entries, n_blocks = mcus_x*mcus_y*blocks_per_mcu, scan order): (byte position << 7) | flag << 6 |
 * bit offset = the reference bit reader's state (bb.pBuf, bb.ulBitOff) at the block's FIRST AC SYMBOL, behind the
 * refill at the top of JPEGDecodeMCU's AC loop (jpeg.inl:2225-2230: offset 0..47); and the block's OWN DC value
 * (n_blocks int16: the pre-scan decoded the DC symbol, :2129-2165).  The decode kernel starts every block at
 * coefficient 1; a 1/8 or progressive thumbnail (:5146-5154, :4964-4966) reads the DC values and nothing else.
 * Flag (bit 6): the reference truncates a magnitude read of this block (its window is not refilled between a code and its
 * magnitude, :2249-2252) -- the kernel emulates that from the entry's exact (pBuf, ulBitOff).  The last entry closes
 * the index: a bit position at or behind the last block's last bit (see jda_index_equivalent).
 * *n_mcus_ok < mcus_x*mcus_y means the pre-scan hit an invalid code in that MCU (the reference
 * returns JPEG_DECODE_ERROR there, jpeg.inl:2137, 2237, 5354-5356). */
const uint8_t *jda_image_scan(const jda_image *img, uint32_t *len);
const uint32_t *jda_image_block_index(const jda_image *img, uint32_t *n_mcus_ok);
const int16_t *jda_image_block_dc(const jda_image *img);
/* The optional part of the index: continuation entries, one every 8 AC symbols of a long block, through which several lanes of
 * the decode kernel share it (photographs: luma blocks of forty symbols beside chroma blocks of four -- a wavefront runs as long as
 * its longest CHUNK then, not its longest block).  cont_first[g] .. cont_first[g + 1] (n_blocks + 1 offsets) are block g's entries in
 * the returned array of *n_cont entries: bits 11:0 the bit position of the entry's first symbol relative to the block's first AC
 * symbol, bits 17:12 the zigzag index of its first coefficient, bits 24:18 the low bits of g.  The serial pre-scan writes them for
 * the images JDA_PREPARE_CONT_* names; a decode to RGB8888 of a 4:2:0 or 4:4:4 image that has entries takes them. */
const uint32_t *jda_image_block_cont(const jda_image *img, const uint32_t **cont_first, uint32_t *n_cont);
/* Do two indexes of n_blocks + 1 entries (one from the serial pre-scan, one read back from the device: jda_dev_image_read_index,
 * jda_pipeline_read_index) name the same decode?  The contract between the two pre-scans: every block's entry has the same bit
 * position (byte position * 8 + bit offset) and the same flag; a FLAGGED block's entry is identical (the reference reader's exact
 * phase); an unflagged block's entry from the device is canonical -- (p >> 3) << 7 | (p & 7) for its bit position p -- where the
 * serial pre-scan stores the reader's phase; the closing entries lie within 41 bits of each other (the device's is behind the DC
 * symbol that the stream's padding decodes to, and rounded up to a byte in a stream with restart intervals).  Returns 1 / 0. */
int jda_index_equivalent(const uint32_t *a, const uint32_t *b, uint32_t n_blocks);
/* Which kernels of the library's code object this process has launched, and how often: "<kernel symbol> <launches>\n" per kernel
 * into buf (NUL-terminated, truncated at cap); returns the bytes the whole report needs.  (Diagnostics: the GPU test-suite
 * holds itself to every kernel the library ships.) */
int jda_kernel_launch_counts(char *buf, int cap);
/* the table blob uploaded to the GPU: DC LUTs 2x1024 B, AC LUTs 2x2048 uint16, quant 4x64 int16, zigzag 64 B,
 * and (ours) the end-of-block code of each AC table, 2 x uint32 = (32 - length) << 16 | code */
const uint8_t *jda_image_tables(const jda_image *img, uint32_t *bytes);
/* number of places where the reference's un-refilled magnitude read drops low bits
 * (SURVEY.md fact 6); informational */
uint32_t jda_image_truncation_events(const jda_image *img);
/* 1: an AC table codes the end-of-block symbol more than once (malformed DHT; jpeg.inl:1066-1275 builds its LUTs per
 * code and decodes it all the same): the kernels then take their general bit reader; informational */
uint32_t jda_image_general_p1(const jda_image *img);

/* Geometry of the decoded surface for (pixel_type, options): bytes per pixel, and the
 * MCU-padded canvas size in output pixels (what the reference's draw callbacks tile). */
int jda_output_geometry(const jda_image_info *info, int32_t pixel_type, int32_t options,
                        int32_t *bytes_per_pixel, int32_t *out_w, int32_t *out_h,
                        int32_t *canvas_w, int32_t *canvas_h);

/* Draw-callback plan (jpeg.inl:5062-5084, 5300-5336): rects[6*i..] = x, y, iWidth, iHeight,
 * iWidthUsed, iBpp of every JPEGDRAW the reference issues.  Returns the count (or -1). */
int jda_draw_plan(const jda_image_info *info, int32_t pixel_type, int32_t options, int32_t max_mcus,
                  int32_t uses_dma, int32_t *rects, int32_t max_rects);

/* The same with a crop rectangle (JPEG_setCropArea, jpeg.inl:682-727; skip logic :5111, :5134-5137).
 * jda_crop_round applies the reference's MCU rounding to a request in place.  crop = {x, y, w, h}
 * already rounded, or NULL.  rects[8*i..] = x, y, iWidth, iHeight, iWidthUsed, iBpp, src_x, src_y with
 * (src_x, src_y) the strip's position in the decoded canvas. */
void jda_crop_round(const jda_image_info *info, int32_t *x, int32_t *y, int32_t *w, int32_t *h);
int jda_draw_plan_ex(const jda_image_info *info, int32_t pixel_type, int32_t options, int32_t max_mcus,
                     int32_t uses_dma, const int32_t *crop, int32_t *rects, int32_t max_rects);
/* The same for decode(xoff, ..): the reference's strip widths of a CROPPED decode depend on the x offset passed to decode()
 * (jpeg.inl:5328 compares jd.x, offset included, with iCropX + iCropCX).  rects as above, x / y relative to the offset. */
int jda_draw_plan_at(const jda_image_info *info, int32_t pixel_type, int32_t options, int32_t max_mcus,
                     int32_t uses_dma, const int32_t *crop, int32_t xoff, int32_t *rects, int32_t max_rects);

/* ------------------------------------------------------------------ device runtime */

typedef struct jda_ctx jda_ctx;        /* one per process per GPU: device, stream, events */
typedef struct jda_dev_image jda_dev_image; /* an image's inputs resident in HBM */
typedef struct jda_batch jda_batch;    /* a launch plan over many resident images */

int jda_device_count(void);
jda_ctx *jda_create(int32_t device, int32_t *err);
void jda_destroy(jda_ctx *ctx);
const char *jda_last_hip_error(const jda_ctx *ctx);
void *jda_stream(jda_ctx *ctx);        /* the hipStream_t every launch of this ctx goes to */

/* device memory helpers (so callers need no HIP binding of their own) */
/* Page-locked HOST memory (hipHostMalloc): a destination the copy back of jda_decode_to_host* reaches at link speed and truly
 * asynchronously -- what lets jda_decode_to_host_bands overlap the copy with the caller's work.  NULL when it cannot be had. */
void *jda_host_alloc(size_t bytes);
void jda_host_free(void *p);
/* page-lock memory the caller already has (a file cache, a receive buffer) for JDA_SUBMIT_PINNED_INPUT; undo before freeing it */
int jda_host_register(void *p, size_t bytes);
int jda_host_unregister(void *p);
void *jda_malloc(jda_ctx *ctx, size_t bytes);
void jda_free(jda_ctx *ctx, void *dptr);
int jda_memset(jda_ctx *ctx, void *dptr, int value, size_t bytes);
int jda_copy_to_host(jda_ctx *ctx, void *host, const void *dptr, size_t bytes);   /* synchronous */
int jda_copy_to_device(jda_ctx *ctx, void *dptr, const void *host, size_t bytes); /* synchronous */

/* H2D: tables + index + filtered scan of one prepared image into one HBM allocation.  For an image prepared
 * with JDA_PREPARE_DEVICE_PRESCAN whose index is still pending, the index is made here on the GPU, equivalent to
 * the serial pre-scan's (jda_index_equivalent; the DC values are equal): one lane per 256 bytes of the scan, whose decoder states settle by
 * self-synchronisation in a few speculative rounds (restart intervals end where the filter found the markers).  A marker that is not where the MCU count puts it, a corrupt or truncated stream, or states that do
 * not settle send the image to the serial host pre-scan instead (the image object is completed in place). */
jda_dev_image *jda_upload(jda_ctx *ctx, jda_image *img, int32_t *err);
/* The same for n images at once: all pending block indexes are made by the same launches (the walk is latency-bound
 * per lane, so throughput comes from the number of segments in flight).  out[i] = device image.
 * Returns JDA_SUCCESS or the first error (then every out[i] is NULL). */
int jda_upload_batch(jda_ctx *ctx, int32_t n, jda_image *const *imgs, jda_dev_image **out);
/* The same, tolerant of holes: imgs[i] == NULL (a file jda_prepare_batch rejected) gives out[i] = NULL and status[i] =
 * JDA_INVALID_PARAMETER; every other image is uploaded (status[i] = JDA_SUCCESS, or the batch's HIP error).  The arrays stay
 * index-aligned with the caller's file list; jda_batch_create accepts the holes and launches nothing for them. */
int jda_upload_batch_ex(jda_ctx *ctx, int32_t n, jda_image *const *imgs, jda_dev_image **out, int32_t *status);
int jda_dev_image_prescan_on_device(const jda_dev_image *dimg);   /* 1: a device pre-scan produced the index */
/* copy the per-block index (n_blocks + 1 entries) and DC values (n_blocks) of a resident image back to the host
 * (either pointer may be NULL); n_blocks = mcus_x * mcus_y * blocks_per_mcu.  Synchronous. */
int jda_dev_image_read_index(jda_ctx *ctx, const jda_dev_image *dimg, uint32_t *index, int16_t *dc);
uint32_t jda_dev_image_mcus_ok(const jda_dev_image *dimg);        /* MCUs the pre-scan validated */
int jda_last_prescan_rounds(const jda_ctx *ctx);                 /* rounds the last device pre-scan on this context took (diagnostics) */
/* The marker / byte-stuffing filter (JPEGFilter, jpeg.inl:1431-1540) run on the GPU over a host buffer, result back on the
 * host: out must hold len bytes; *out_len = filtered length; restart_pos[0] = 0 and restart_pos[k] = filtered offset at
 * which the k-th RSTn marker stood (first restart_cap entries), *n_restarts = markers seen.  A stand-alone entry point
 * (jda_upload_batch still takes the scan filtered by jda_prepare on the host); exposed for callers that want the filtered
 * scan made on the GPU, and for the tests. */
int jda_filter_on_device(jda_ctx *ctx, const uint8_t *raw, int32_t len, uint8_t *out, int32_t *out_len,
                         uint32_t *restart_pos, int32_t restart_cap, int32_t *n_restarts);
void jda_dev_image_free(jda_ctx *ctx, jda_dev_image *dimg);
size_t jda_dev_image_bytes(const jda_dev_image *dimg);

/* where one image of a batch is written */
typedef struct jda_output {
    void *pixels;            /* DEVICE pointer, 16-byte aligned */
    int32_t pitch_bytes;     /* multiple of 16 */
    int32_t width_px;        /* clip: pixels written per row (<= canvas_w) */
    int32_t rows;            /* clip: rows written (<= canvas_h) */
} jda_output;

/* Build the launch plan for n resident images decoded with one (pixel_type, options) each. */
jda_batch *jda_batch_create(jda_ctx *ctx, int32_t n, jda_dev_image *const *images,
                            const jda_output *outputs, const int32_t *pixel_types,
                            const int32_t *options, int32_t *err);
/* The same with a rectangle of MCUs per image: mcu_rects[4 i ..] = {mx0, my0, mx1, my1} (half open, MCU units), or NULL for whole
 * images.  Only the tiles of the rectangle are launched -- the crop-aware decode (the reference skips the MCU rows above the crop
 * and the MCUs left and right of it, jpeg.inl:5111, 5134-5137; here they are not even visited: the per-block index lets a tile start
 * at any MCU).  The surface keeps the whole image's geometry; pixels outside the rectangle are not written. */
jda_batch *jda_batch_create_rect(jda_ctx *ctx, int32_t n, jda_dev_image *const *images,
                                 const jda_output *outputs, const int32_t *pixel_types,
                                 const int32_t *options, const int32_t *mcu_rects, int32_t *err);
void jda_batch_destroy(jda_ctx *ctx, jda_batch *batch);

/* Enqueue the decode kernels for the whole batch on the ctx stream (asynchronous). */
int jda_batch_decode(jda_ctx *ctx, jda_batch *batch);
/* launch statistics of the plan */
typedef struct jda_batch_stats {
    int64_t source_pixels;      /* sum of width*height */
    int64_t output_bytes;       /* bytes the kernels write */
    int64_t scan_bytes;         /* filtered entropy-coded bytes read */
    int64_t index_bytes;        /* per-block index + DC predictor bytes read */
    int64_t table_bytes;
    int32_t n_launches;         /* kernel launches per jda_batch_decode */
    int32_t n_workgroups;
    int64_t tiles;              /* wavefront tiles with work in the plan */
    int64_t tiles_whole_images; /* ... and what the whole images would have taken (crop-aware plans launch fewer) */
} jda_batch_stats;
int jda_batch_get_stats(const jda_batch *batch, jda_batch_stats *stats);
/* status[i] for every image of the plan: JDA_SUCCESS; JDA_DECODE_ERROR = the stream has a bad MCU (the MCUs before it are decoded,
 * what the reference delivers before it returns the error, jpeg.inl:5354-5356); JDA_INVALID_PARAMETER = a hole (images[i] == NULL) */
int jda_batch_get_status(const jda_batch *batch, int32_t *status);

int jda_sync(jda_ctx *ctx);

/* A position-dependent 64-bit checksum of each of n decoded surfaces (DEVICE pointers; rows x row_bytes[i] at pitch), computed
 * on the GPU: sum over the dwords d at linear index i of (uint64)((d ^ (i * 0x9E3779B1)) * 0x85EBCA6B mod 2^32) * (2 i + 1), mod 2^64,
 * a row's tail bytes zero-extended.  Lets a multi-GPU driver prove "every image decoded exactly once, identically" without moving
 * pixels (the reference has no such notion: its pixels go to a display as they are made).  Synchronous. */
int jda_checksum_surfaces(jda_ctx *ctx, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *checksums);
/* PCI bus id ("0000:8e:00.0") of the context's GPU, for NUMA placement of the host threads that feed it; buf >= 16 bytes */
int jda_device_pci_bus_id(jda_ctx *ctx, char *buf, int32_t len);
int jda_device_pci_bus_id_of(int32_t device, char *buf, int32_t len);      /* the same by device ordinal, without a context */

/* HIP-event timing on the ctx stream: start/stop record events on that stream; elapsed blocks
 * until stop has happened and returns milliseconds (<0 on error). */
int jda_timer_start(jda_ctx *ctx);
int jda_timer_stop(jda_ctx *ctx);
double jda_timer_elapsed_ms(jda_ctx *ctx);

/* One-call convenience used by the JPEGDEC class: prepare + upload + decode + copy back into a
 * HOST canvas of canvas_w x canvas_h pixels (pitch_bytes per row). */
int jda_decode_to_host(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type,
                       int32_t options, void *host_pixels, int32_t pitch_bytes, int32_t rows);
/* The same; *mcus_decoded (may be NULL) = MCUs decoded before the first invalid code, in scan order (all of them on
 * JDA_SUCCESS; fewer with JDA_DECODE_ERROR: the reference stops at that MCU, jpeg.inl:2137, 2237, 5354-5356 -- the
 * canvas holds the MCUs before it, zeros behind). */
int jda_decode_to_host_ex(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type,
                          int32_t options, void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded);
/* The same, decoding only the MCUs of mcu_rect = {mx0, my0, mx1, my1} (half open; NULL: everything): the canvas keeps its
 * geometry, the rows of the rectangle are written (zeros left and right of it), the others are not touched.  tiles (may be NULL):
 * [0] wavefront tiles launched, [1] tiles of the whole image. */
int jda_decode_to_host_rect(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                            void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles);

/* The same with flags.  JDA_TO_HOST_KEEP_UNDECODED (whole image only, mcu_rect == NULL): when the stream has a bad MCU, copy back
 * only the MCUs in front of it -- whole MCU rows, then the row's MCUs before the bad one -- and leave every other byte of
 * host_pixels as it was: what the reference's early return does to a caller's framebuffer (jpeg.inl:5354-5356). */
#define JDA_TO_HOST_KEEP_UNDECODED 1
int jda_decode_to_host_flags(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags);

/* The same, the copy back cut into n_bands (<= 8) bands of whole MCU rows: band_ready(user, row0, row1) is called, in order, as soon as
 * rows [row0, row1) of host_pixels have landed -- what the caller does with them (JPEGDEC::decode replays the reference's JPEGDRAW
 * callbacks, jpeg.inl:5300-5336) overlaps the rest of the copy.  *mcus_decoded is set before the first call.  The copy is NOT cut --
 * one copy, band_ready never called, the caller looks at all rows after the return -- without a callback, with n_bands <= 1, with a
 * rectangle, and with JDA_TO_HOST_KEEP_UNDECODED on a stream that has a bad MCU. */
typedef void(jda_band_callback)(void *user, int32_t row0, int32_t row1);
int jda_decode_to_host_bands(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, const int32_t *mcu_rect,
                             void *host_pixels, int32_t pitch_bytes, int32_t rows, int32_t *mcus_decoded, int32_t *tiles, int32_t flags,
                             int32_t n_bands, jda_band_callback *band_ready, void *user);

/* The whole image STRIP-MAJOR: as the JPEGDRAW strips the reference hands to its draw callback (jpeg.inl:5300-5336) -- strip_mcus MCUs
 * wide (a row's last strip: what is left), one MCU row high, in raster order, every strip's pixels contiguous with the strip's own
 * width as pitch; strip (row y, column s) starts (y * ceil(mcus_x / strip_mcus) + s) * strip_mcus * mcu_w' * mcu_h' * bpp bytes into
 * host_pixels (mcu_w', mcu_h': the MCU in output pixels).  The kernels write the surface in that layout; a consumer hands out
 * pointers instead of copying strips together.  host_bytes >= mcus_y * ceil(mcus_x / strip_mcus) * that strip size.  The copy back
 * in n_bands bands of MCU rows as jda_decode_to_host_bands (band_ready may be NULL). */
int jda_decode_to_host_strips(jda_ctx *ctx, const uint8_t *jpeg, int32_t len, int32_t pixel_type, int32_t options, int32_t strip_mcus,
                              void *host_pixels, size_t host_bytes, int32_t *mcus_decoded, int32_t n_bands, jda_band_callback *band_ready, void *user);

/* ------------------------------------------------------------------ the streamed pipeline
 * Files in, pixels resident in HBM out, batch after batch: the host parses headers and builds tables (microseconds per file);
 * the unfiltered entropy-coded bytes go to the GPU, which filters them (JPEGFilter, jpeg.inl:1431-1540), makes the per-block
 * index (equivalent to the serial pre-scan's: jda_index_equivalent) and decodes (jpeg.inl:5109-5353).  Upload + filter + pre-scan of
 * batch n+1 run on streams of their own under the decode of batch n (three batches in flight keep the GPU busy).  Images the device walk cannot take or that fail its checks
 * (progressive, corrupt, truncated, ...) are redone through the serial host pre-scan when the batch is waited for, so every
 * image ends with the status -- and the pixels -- the one-image path (jda_decode_to_host) gives it.
 *   jda_pipeline_create   max_images per batch; depth = batches in flight (1..8); host_threads <= 0: up to 8
 *   jda_pipeline_submit   enqueue one batch; the JPEG buffers and the output surfaces (DEVICE pointers, as jda_output) must
 *                         stay valid until the batch has been waited for.  *ticket identifies the batch.
 *   jda_pipeline_wait     block until the batch is decoded; status[i] = JDA_SUCCESS or the image's error (may be NULL).
 *                         A batch must be waited for before `depth` further batches are submitted.  Returns JDA_SUCCESS unless
 *                         the pipeline itself failed (a bad image is reported in status[], it does not fail its batch). */
typedef struct jda_pipeline jda_pipeline;
typedef struct jda_pipeline_stats {
    int64_t images, device_images, host_path_images, failed_images;
    int64_t source_pixels, compressed_bytes, h2d_bytes;
    int32_t launches, spec_rounds_max;
} jda_pipeline_stats;
jda_pipeline *jda_pipeline_create(jda_ctx *ctx, int32_t max_images, int32_t depth, int32_t host_threads, int32_t *err);
void jda_pipeline_destroy(jda_pipeline *p);
int jda_pipeline_submit(jda_pipeline *p, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                        const int32_t *pixel_types, const int32_t *options, int32_t *ticket);
/* The same with flags.  JDA_SUBMIT_PINNED_INPUT: every jpegs[i] lies in page-locked host memory (jda_host_alloc, or any buffer
 * made known with jda_host_register) -- the copy engine then reads the files' entropy-coded bytes where they are, and no host core
 * copies them into the pipeline's own page-locked mirror first (the host's largest share of a batch; what lets a rank with two or
 * three cores feed its GPU).  Files that lie next to one another in memory (gaps up to 64 KB: a loader's arena, a ring of receive
 * buffers) travel as one copy command; a command that would carry less than 128 KB is not worth what it costs the submitting
 * thread, so isolated small files still go through the mirror. */
#define JDA_SUBMIT_PINNED_INPUT 1
int jda_pipeline_submit_ex(jda_pipeline *p, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                           const int32_t *pixel_types, const int32_t *options, int32_t flags, int32_t *ticket);
int jda_pipeline_wait(jda_pipeline *p, int32_t ticket, int32_t *status);
int jda_pipeline_get_stats(const jda_pipeline *p, jda_pipeline_stats *out);   /* totals over the batches waited for */
/* diagnostics: after jda_pipeline_wait(ticket), before `depth` more batches are submitted -- the per-block index (n_blocks + 1
 * entries) and DC values (n_blocks) the device made for image i, and its filtered scan length (any pointer may be NULL) */
int jda_pipeline_read_index(jda_pipeline *p, int32_t ticket, int32_t i, uint32_t *index, int16_t *dc, uint32_t *filtered_len);

/* ------------------------------------------------------------------ the node: one host process, every GPU
 * Images are independent (the reference zeroes its whole state per image, src/JPEGDEC.cpp:66): a node shards a LIST of files by
 * image.  jda_node owns one context + one streamed pipeline per device and deals a submitted list out in contiguous blocks --
 * device k of K takes images [first, first + count) of jda_node_shard (sizes differ by at most one: the rule the multi-process
 * bench uses).  Every device has ONE PERSISTENT host thread, made with the node and pinned to the CPUs of its GPU's NUMA node
 * (devices on one NUMA node share its CPUs; jda_node_placement tells): it creates the device's context and pipeline -- whose
 * workers inherit the placement -- and runs the device's half of every submit / wait / checksum call, so the calling thread makes
 * no HIP call and keeps its current device.  Pixels never cross between GPUs: outputs[i].pixels must be a
 * DEVICE pointer on the device that owns image i (allocate with jda_malloc(jda_node_context(node, k), ..)).  What comes back
 * is status[i] per image and, for a proof that every image was decoded once and identically wherever it landed, per-image
 * checksums made where the pixels are (jda_node_checksums = jda_checksum_surfaces per device).
 *   jda_node_create   devices == NULL or n_devices <= 0: every visible device (0 .. jda_device_count() - 1); an ordinal may be named more
 *                     than once (each entry is a pipeline, a context and a host thread of its own on that device); max_images_per_device
 *                     bounds a device's block; depth / host_threads_per_device as jda_pipeline_create.  Fails with
 *                     JDA_ERROR_NO_DEVICE when there is no GPU: there is no CPU decode path.
 *   jda_node_submit   n <= devices * max_images_per_device images; buffers and surfaces stay valid until the list is waited for
 *   jda_node_wait     blocks until every device has decoded its block; status may be NULL */
typedef struct jda_node jda_node;
jda_node *jda_node_create(const int32_t *devices, int32_t n_devices, int32_t max_images_per_device, int32_t depth,
                          int32_t host_threads_per_device, int32_t *err);
void jda_node_destroy(jda_node *node);
int32_t jda_node_device_count(const jda_node *node);
int32_t jda_node_device(const jda_node *node, int32_t k);          /* HIP device ordinal of the node's k-th device */
jda_ctx *jda_node_context(jda_node *node, int32_t k);              /* its context (owned by the node) */
void jda_node_shard(const jda_node *node, int32_t n, int32_t k, int32_t *first, int32_t *count);
void jda_node_shard_of(int32_t n_devices, int32_t n, int32_t k, int32_t *first, int32_t *count);     /* the same rule without a node */
int jda_node_submit(jda_node *node, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                    const int32_t *pixel_types, const int32_t *options, int32_t *ticket);
int jda_node_submit_ex(jda_node *node, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                       const int32_t *pixel_types, const int32_t *options, int32_t flags, int32_t *ticket);      /* flags: JDA_SUBMIT_* */
int jda_node_wait(jda_node *node, int32_t ticket, int32_t *status);
/* where device k's host thread runs: its GPU's NUMA node (-1: unknown) and how many CPUs it is pinned to (0: not pinned) */
int jda_node_placement(const jda_node *node, int32_t k, int32_t *numa_node, int32_t *cpus_pinned);
int jda_node_checksums(jda_node *node, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *checksums);
int jda_node_get_stats(const jda_node *node, jda_pipeline_stats *out);   /* sums over the devices' pipelines */

const char *jda_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JPEGDEC_AMD_H */
