// Object semantics at the drop-in boundary (reference src/JPEGDEC.h:249-309): the class is copied and assigned like the plain struct
// the reference's is; a C JPEGIMAGE needs no initialisation and, for RAM / FLASH sources, no JPEG_close (src/JPEGDEC.cpp:232-236).
// Usage: semantics_user file.jpg -- opens only (no GPU needed); prints "ok" and exits 0, or the number of the check that failed.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

#include "JPEGDEC.h"

static int draw(JPEGDRAW *) { return 1; }
#define CHECK(n, cond) do { if (!(cond)) { printf("check %d failed\n", n); return n; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 100;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 101;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> jpeg((size_t)len);
    if (fread(jpeg.data(), 1, (size_t)len, f) != (size_t)len) return 102;
    fclose(f);

    // ---- the class: copies and moves
    JPEGDEC a;
    CHECK(1, a.openRAM(jpeg.data(), (int)len, draw) == 1);
    const int w = a.getWidth(), h = a.getHeight();
    CHECK(2, w > 0 && h > 0);
    a.setPixelType(RGB8888);
    JPEGDEC b(a);                                   // copy: the open image with what was set on it
    CHECK(3, b.getWidth() == w && b.getHeight() == h && b.getPixelType() == RGB8888);
    JPEGDEC c;
    c = a;                                          // assignment
    CHECK(4, c.getWidth() == w && c.getSubSample() == a.getSubSample());
    a.close();                                      // the copies outlive the original
    CHECK(5, b.getWidth() == w);
    JPEGDEC d(std::move(b));                        // move: takes the image, leaves a closed object that can be opened again
    CHECK(6, d.getWidth() == w && b.getWidth() == 0);
    CHECK(7, b.openRAM(jpeg.data(), (int)len, draw) == 1 && b.getWidth() == w);
    std::vector<JPEGDEC> pool(3, d);                // containers of decoders, as with the reference's struct
    CHECK(8, pool[2].getHeight() == h);

    // ---- the C flavour: no initialisation, no close for RAM sources, any number of handles (the state is the caller's struct)
    enum { N = 2000 };
    static JPEGIMAGE imgs[N];
    memset(imgs, 0xA5, sizeof(imgs));               // stack-garbage look-alike: an open must cope
    CHECK(9, JPEG_getWidth(&imgs[0]) == 0 && JPEG_getLastError(&imgs[0]) == JPEG_INVALID_PARAMETER);       // garbage is not a handle
    for (int i = 0; i < N; i++) {
        CHECK(10, JPEG_openRAM(&imgs[i], jpeg.data(), (int)len, draw) == 1);
        CHECK(11, JPEG_getWidth(&imgs[i]) == w);
    }
    for (int i = 0; i < N; i++) CHECK(12, JPEG_getWidth(&imgs[i]) == w && JPEG_getHeight(&imgs[i]) == h);    // every one of them stays open (the reference: src/JPEGDEC.h:199-239 is the caller's memory)
    JPEG_setPixelType(&imgs[7], RGB8888);
    JPEG_setCropArea(&imgs[7], 16, 16, 64, 32);
    int cx = 0, cy = 0, cw = 0, ch = 0;
    JPEG_getCropArea(&imgs[7], &cx, &cy, &cw, &ch);
    CHECK(13, cx == 16 && cy == 16 && cw > 0 && ch > 0);                              // what is set on a handle stays with it ..
    JPEG_getCropArea(&imgs[8], &cx, &cy, &cw, &ch);
    CHECK(14, cx == 0 && cy == 0 && cw == w && ch == h);                              // .. and with no other
    for (int k = 0; k < 1000; k++) CHECK(15, JPEG_openRAM(&imgs[5], jpeg.data(), (int)len, draw) == 1);    // re-opening a handle
    JPEGIMAGE copy = imgs[3];                                                         // a struct copy is not a handle
    CHECK(16, JPEG_getWidth(&copy) == 0 && JPEG_getWidth(&imgs[3]) == w);
    JPEG_close(&imgs[N - 1]);
    CHECK(17, JPEG_getWidth(&imgs[N - 1]) == 0);
    for (int k = 0; k < 50; k++) {                                                   // file sources: opened, re-opened without a close (the bytes are given back), closed
        JPEGIMAGE fi;
        memset(&fi, 0x5A, sizeof(fi));
        CHECK(18, JPEG_openFile(&fi, argv[1], draw) == 1 && JPEG_getWidth(&fi) == w);
        CHECK(19, JPEG_openFile(&fi, argv[1], draw) == 1 && JPEG_getHeight(&fi) == h);
        JPEG_close(&fi);
        CHECK(20, JPEG_getWidth(&fi) == 0);
    }
    CHECK(21, JPEG_openFile(&imgs[0], "/nonexistent/file.jpg", draw) == 0);
    printf("ok\n");
    return 0;
}
