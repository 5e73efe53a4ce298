/* A C translation unit using the reference's C-flavour API (src/JPEGDEC.h:288-309) against libjpegdec_amd.so,
 * the way linux/examples/c_cmdline/main.c uses the reference.  Usage: c_user file.jpg [out.rgba]
 * Prints "WxH subsample bpp" after open; with an output file it decodes RGB8888 through draw callbacks into
 * a W x H canvas and writes it (needs a GPU).  Exit code = getLastError(). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "JPEGDEC.h"

static uint8_t *g_canvas;
static int g_w, g_h;

static int draw(JPEGDRAW *d)
{
    for (int r = 0; r < d->iHeight; r++) {
        int y = d->y + r;
        if (y >= g_h) break;
        int n = d->iWidthUsed;
        if (d->x + n > g_w) n = g_w - d->x;
        if (n > 0) memcpy(g_canvas + ((size_t)y * g_w + d->x) * 4, (uint8_t *)d->pPixels + (size_t)r * d->iWidth * 4, (size_t)n * 4);
    }
    return 1;
}

int main(int argc, char **argv)
{
    JPEGIMAGE jpg;                       /* uninitialised on purpose: the open call must cope */
    if (argc < 2) return 100;
    if (!JPEG_openFile(&jpg, argv[1], draw)) { printf("open failed %d\n", JPEG_getLastError(&jpg)); return 101; }
    g_w = JPEG_getWidth(&jpg); g_h = JPEG_getHeight(&jpg);
    printf("%dx%d %d %d\n", g_w, g_h, JPEG_getSubSample(&jpg), JPEG_getBpp(&jpg));
    if (argc < 3) { JPEG_close(&jpg); return 0; }
    g_canvas = (uint8_t *)calloc((size_t)g_w * g_h, 4);
    JPEG_setPixelType(&jpg, RGB8888);
    int ok = JPEG_decode(&jpg, 0, 0, 0);
    int err = JPEG_getLastError(&jpg);
    if (ok) {
        FILE *f = fopen(argv[2], "wb");
        fwrite(g_canvas, 4, (size_t)g_w * g_h, f);
        fclose(f);
    }
    JPEG_close(&jpg);
    free(g_canvas);
    return ok ? 0 : (err ? err : 102);
}
