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

    // ---- the C flavour: no initialisation, no close for RAM sources, slots recycled
    enum { N = 200 };
    static JPEGIMAGE imgs[N];
    memset(imgs, 0xA5, sizeof(imgs));               // stack-garbage look-alike: an open must cope
    for (int i = 0; i < N; i++) {
        CHECK(10, JPEG_openRAM(&imgs[i], jpeg.data(), (int)len, draw) == 1);
        CHECK(11, JPEG_getWidth(&imgs[i]) == w);
    }
    CHECK(12, JPEG_getWidth(&imgs[N - 1]) == w);                                     // the recent ones are live
    CHECK(13, JPEG_getWidth(&imgs[0]) == 0 && JPEG_getLastError(&imgs[0]) == JPEG_INVALID_PARAMETER);   // the oldest slot was recycled: stale, not wrong
    CHECK(14, JPEG_openRAM(&imgs[0], jpeg.data(), (int)len, draw) == 1 && JPEG_getWidth(&imgs[0]) == w);   // and can be opened again
    for (int k = 0; k < 1000; k++) CHECK(15, JPEG_openRAM(&imgs[5], jpeg.data(), (int)len, draw) == 1);    // re-opening one handle reuses its slot
    CHECK(16, JPEG_getWidth(&imgs[N - 1]) == w);
    JPEG_close(&imgs[N - 1]);
    CHECK(17, JPEG_getWidth(&imgs[N - 1]) == 0);
    printf("ok\n");
    return 0;
}
