// include/JPEGDEC.h -- the reference's public surface (bitbank2/JPEGDEC src/JPEGDEC.h), kept as the
// API contract for the decode path, implemented on the MI355X-native C-ABI of jpegdec_amd.h.
//
// Names, argument meaning, constants and the return / error convention are the reference's
// (src/JPEGDEC.h:68-75 options, :102-111 pixel types, :119-126 errors, :143-156 JPEGDRAW and the
// draw callback, :249-287 class JPEGDEC); the state behind them is not: the reference keeps an
// 18 KB JPEGIMAGE inside the object and streams the file 2 KiB at a time, this class keeps a
// small handle and hands the whole image to the GPU.
//
// open*/decode return 1 = success, 0 = failure; getLastError() gives the code.
#ifndef JPEGDEC_AMD_JPEGDEC_H
#define JPEGDEC_AMD_JPEGDEC_H

#include <stdint.h>

// ---- decoder options (reference src/JPEGDEC.h:68-75)
#define JPEG_AUTO_ROTATE 1        /* defined but never read by the reference either */
#define JPEG_SCALE_HALF 2
#define JPEG_SCALE_QUARTER 4
#define JPEG_SCALE_EIGHTH 8
#define JPEG_LE_PIXELS 16         /* defined but never read by the reference either */
#define JPEG_EXIF_THUMBNAIL 32
#define JPEG_LUMA_ONLY 64
#define JPEG_USES_DMA 128

#define MAX_BUFFERED_PIXELS 2048  /* size of the strip a draw callback receives, in uint16 units */

enum { JPEG_MODE_BASELINE = 0, JPEG_MODE_PROGRESSIVE, JPEG_MODE_INVALID };

enum {
    RGB565_LITTLE_ENDIAN = 0,
    RGB565_BIG_ENDIAN,
    RGB8888,
    EIGHT_BIT_GRAYSCALE,
    FOUR_BIT_DITHERED,
    TWO_BIT_DITHERED,
    ONE_BIT_DITHERED,
    INVALID_PIXEL_TYPE
};

enum {
    JPEG_SUCCESS = 0,
    JPEG_INVALID_PARAMETER,
    JPEG_DECODE_ERROR,
    JPEG_UNSUPPORTED_FEATURE,
    JPEG_INVALID_FILE,
    JPEG_ERROR_MEMORY,
    JPEG_ERROR_NO_DEVICE,        /* extension: no usable MI355X / HIP runtime (there is no CPU fallback) */
    JPEG_ERROR_HIP               /* extension: a HIP call failed */
};

typedef struct jpeg_file_tag {
    int32_t iPos;
    int32_t iSize;
    uint8_t *pData;
    void *fHandle;
} JPEGFILE;

typedef struct jpeg_draw_tag {
    int x, y;                 // upper left corner of this block of pixels
    int iWidth, iHeight;      // size of this block
    int iWidthUsed;           // columns that lie inside the image
    int iBpp;                 // 8, 16 or 32
    uint16_t *pPixels;        // strip of iWidth x (MCU height) pixels, pitch = iWidth pixels
    void *pUser;
} JPEGDRAW;

typedef int32_t(JPEG_READ_CALLBACK)(JPEGFILE *pFile, uint8_t *pBuf, int32_t iLen);
typedef int32_t(JPEG_SEEK_CALLBACK)(JPEGFILE *pFile, int32_t iPosition);
typedef int(JPEG_DRAW_CALLBACK)(JPEGDRAW *pDraw);
typedef void *(JPEG_OPEN_CALLBACK)(const char *szFilename, int32_t *pFileSize);
typedef void(JPEG_CLOSE_CALLBACK)(void *pHandle);

#ifdef __cplusplus

struct jpegdec_amd_state;   // private

class JPEGDEC {
  public:
    JPEGDEC();
    ~JPEGDEC();
    // The reference's class is a plain struct around its JPEGIMAGE (src/JPEGDEC.h:286): objects are copied and assigned freely.
    // Here a copy clones the open image (file-sourced data included; the close callback stays with the original), a move takes it.
    JPEGDEC(const JPEGDEC &);
    JPEGDEC &operator=(const JPEGDEC &);
    JPEGDEC(JPEGDEC &&);
    JPEGDEC &operator=(JPEGDEC &&) noexcept;

    int openRAM(uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw);
    int openFLASH(const uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw);
    int open(const char *szFilename, JPEG_OPEN_CALLBACK *pfnOpen, JPEG_CLOSE_CALLBACK *pfnClose,
             JPEG_READ_CALLBACK *pfnRead, JPEG_SEEK_CALLBACK *pfnSeek, JPEG_DRAW_CALLBACK *pfnDraw);
    int open(const char *szFilename, JPEG_DRAW_CALLBACK *pfnDraw);
    int open(void *fHandle, int iDataSize, JPEG_CLOSE_CALLBACK *pfnClose, JPEG_READ_CALLBACK *pfnRead,
             JPEG_SEEK_CALLBACK *pfnSeek, JPEG_DRAW_CALLBACK *pfnDraw);
    void setFramebuffer(void *pFramebuffer);
    void setDevice(int iDevice);       /* not in the reference: the GPU this object decodes on (default $JPEGDEC_AMD_DEVICE, else 0) */
    void setCropArea(int x, int y, int w, int h);
    void getCropArea(int *x, int *y, int *w, int *h);
    void close();
    int decode(int x, int y, int iOptions);
    int decodeDither(uint8_t *pDither, int iOptions);
    int decodeDither(int x, int y, uint8_t *pDither, int iOptions);
    int getOrientation();
    int getWidth();
    int getHeight();
    int getBpp();
    void setUserPointer(void *p);
    int getSubSample();
    int getJPEGType();
    int hasThumb();
    int getThumbWidth();
    int getThumbHeight();
    int getLastError();
    void setPixelType(int iType);
    int getPixelType();
    void setMaxOutputSize(int iMaxMCUs);

  private:
    jpegdec_amd_state *_jpeg;
    friend struct jpegdec_amd_c_api;      // (the C flavour below is served by objects of this class)
};

#endif // __cplusplus

// ---- the C flavour of the API (reference src/JPEGDEC.h:288-309, bodies src/jpeg.inl:564-739).  The reference
// offers it to C translation units that include jpeg.inl; here the functions live in libjpegdec_amd.so and are
// callable from C and C++ alike.  JPEGIMAGE is caller-allocated as in the reference and holds the open image as plain
// data (there: the 18 KB decoder state; here: source pointer, parsed header, settings).  As in the reference, a JPEGIMAGE
// needs no initialisation before JPEG_open*, any number of them may be open, and a RAM / FLASH source needs no JPEG_close
// (src/JPEGDEC.cpp:232-236: a no-op there).  As with the reference's struct (src/JPEGDEC.h:199-239: plain state), a struct copy
// (assignment, memcpy, a growing array that moves its elements) of an open JPEGIMAGE is an open JPEGIMAGE of the same image with
// the same settings, independent of the original from then on.  A file-sourced image owns the file's bytes until JPEG_close, as
// the reference's owns its open file; copies of it share them as the reference's copies share the FILE: closing one of them (or
// re-opening the handle that read the file) closes them all -- the others answer like closed handles, and closing them again is harmless.
typedef struct jpeg_image_tag {
    uint32_t magic[2];              /* set by JPEG_open*: lets an uninitialised (stack) JPEGIMAGE be told from an open one */
    struct jpeg_image_tag *file_owner;  /* JPEG_openFile: the handle that read the file (re-opening THAT handle gives the bytes back) */
    void *file_data;                /* JPEG_openFile: the file's bytes, freed by JPEG_close */
    uint64_t file_check;            /* the serial number the library's list of live file buffers holds for file_data: bytes are freed only while
                                       (file_data, file_check) is on that list, so closing one copy closes every copy of the handle */
    uint64_t state[40];             /* the open image (opaque plain data) */
} JPEGIMAGE;

#ifdef __cplusplus
extern "C" {
#endif
int JPEG_openRAM(JPEGIMAGE *pJPEG, uint8_t *pData, int iDataSize, JPEG_DRAW_CALLBACK *pfnDraw);
int JPEG_openFile(JPEGIMAGE *pJPEG, const char *szFilename, JPEG_DRAW_CALLBACK *pfnDraw);
void JPEG_setFramebuffer(JPEGIMAGE *pJPEG, void *pFramebuffer);
void JPEG_setDevice(JPEGIMAGE *pJPEG, int iDevice);   /* not in the reference: see JPEGDEC::setDevice */
void JPEG_setCropArea(JPEGIMAGE *pJPEG, int x, int y, int w, int h);
void JPEG_getCropArea(JPEGIMAGE *pJPEG, int *x, int *y, int *w, int *h);
int JPEG_getWidth(JPEGIMAGE *pJPEG);
int JPEG_getHeight(JPEGIMAGE *pJPEG);
int JPEG_decode(JPEGIMAGE *pJPEG, int x, int y, int iOptions);
int JPEG_decodeDither(JPEGIMAGE *pJPEG, uint8_t *pDither, int iOptions);
void JPEG_close(JPEGIMAGE *pJPEG);
int JPEG_getLastError(JPEGIMAGE *pJPEG);
int JPEG_getOrientation(JPEGIMAGE *pJPEG);
int JPEG_getBpp(JPEGIMAGE *pJPEG);
int JPEG_getSubSample(JPEGIMAGE *pJPEG);
int JPEG_hasThumb(JPEGIMAGE *pJPEG);
int JPEG_getThumbWidth(JPEGIMAGE *pJPEG);
int JPEG_getThumbHeight(JPEGIMAGE *pJPEG);
void JPEG_setPixelType(JPEGIMAGE *pJPEG, int iType);
void JPEG_setMaxOutputSize(JPEGIMAGE *pJPEG, int iMaxMCUs);
#ifdef __cplusplus
}
#endif
#endif
