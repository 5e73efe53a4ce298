// Object semantics at the drop-in boundary (reference src/JPEGDEC.h:249-309): the class is copied and assigned like the plain struct
// the reference's is; a C JPEGIMAGE needs no initialisation and, for RAM / FLASH sources, no JPEG_close (src/JPEGDEC.cpp:232-236).
// Usage: semantics_user file.jpg -- opens only (no GPU needed); prints "ok" and exits 0, or the number of the check that failed.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
    JPEG_setPixelType(&imgs[3], RGB565_BIG_ENDIAN);
    JPEGIMAGE copy = imgs[3];                                                         // a struct copy is an open handle of the same image (src/JPEGDEC.h:199-239: plain state) ..
    CHECK(16, JPEG_getWidth(&copy) == w && JPEG_getWidth(&imgs[3]) == w);
    JPEG_setCropArea(&copy, 32, 32, 64, 64);                                          // .. independent of the original from then on
    JPEG_getCropArea(&imgs[3], &cx, &cy, &cw, &ch);
    CHECK(22, cx == 0 && cy == 0 && cw == w && ch == h);
    JPEG_getCropArea(&copy, &cx, &cy, &cw, &ch);
    CHECK(23, cx == 32 && cy == 32);
    {                                                                                 // a growing array moves its elements: they stay open
        std::vector<JPEGIMAGE> grow;
        for (int i = 0; i < 300; i++) {
            JPEGIMAGE one;
            memset(&one, 0x77, sizeof(one));
            CHECK(24, JPEG_openRAM(&one, jpeg.data(), (int)len, draw) == 1);
            grow.push_back(one);
        }
        for (size_t i = 0; i < grow.size(); i++) CHECK(25, JPEG_getWidth(&grow[i]) == w && JPEG_getHeight(&grow[i]) == h);
    }
    JPEG_close(&imgs[N - 1]);
    CHECK(17, JPEG_getWidth(&imgs[N - 1]) == 0);
    for (int k = 0; k < 50; k++) {                                                   // file sources: opened, re-opened without a close (the bytes are given back), closed
        JPEGIMAGE fi;
        memset(&fi, 0x5A, sizeof(fi));
        CHECK(18, JPEG_openFile(&fi, argv[1], draw) == 1 && JPEG_getWidth(&fi) == w);
        CHECK(19, JPEG_openFile(&fi, argv[1], draw) == 1 && JPEG_getHeight(&fi) == h);
        JPEGIMAGE moved = fi;                                                         // the struct moved elsewhere: closed where it is now
        memset(&fi, 0, sizeof(fi));
        CHECK(26, JPEG_getWidth(&moved) == w);
        JPEG_close(&moved);
        CHECK(20, JPEG_getWidth(&moved) == 0);
    }
    for (int k = 0; k < 200; k++) {                                                  // copy, close the COPY, then use / reopen / close the original -- no memset in between:
        JPEGIMAGE a;                                                                  // (the same stack slot every turn, never initialised: what it holds is the last turn's closed handle)
        if (k == 0) memset(&a, 0x21, sizeof(a));
        CHECK(30, JPEG_openFile(&a, argv[1], draw) == 1);
        JPEGIMAGE b = a;
        JPEG_close(&b);                                                               // gives the bytes back, once
        CHECK(31, JPEG_getWidth(&a) == 0 && JPEG_decode(&a, 0, 0, 0) == 0);           // the original knows: a closed handle, not a pointer into freed memory
        if (k & 1) JPEG_close(&a);                                                    // closing it again frees nothing ..
        if (k & 2) JPEG_close(&b);
        if (k & 4) { CHECK(32, JPEG_openFile(&a, argv[1], draw) == 1 && JPEG_getWidth(&a) == w); JPEG_close(&a); }   // .. and so does opening it again
    }
    {                                                                                 // the original closed first, the copy afterwards
        JPEGIMAGE a, b;
        CHECK(33, JPEG_openFile(&a, argv[1], draw) == 1);
        b = a;
        JPEG_close(&a);
        CHECK(34, JPEG_getWidth(&b) == 0);
        JPEG_close(&b);
        CHECK(35, JPEG_openFile(&b, argv[1], draw) == 1 && JPEG_getWidth(&b) == w);   // a copy that is opened again reads its own bytes
        a = b;
        CHECK(36, JPEG_openRAM(&a, jpeg.data(), (int)len, draw) == 1);                 // a COPY opened again leaves the file's bytes to the handle that read them
        CHECK(37, JPEG_getWidth(&b) == w);
        JPEG_close(&b);
    }
    CHECK(21, JPEG_openFile(&imgs[0], "/nonexistent/file.jpg", draw) == 0);
    {                                                                                 // a file that is read and is not a JPEG: the failed open keeps nothing (no close follows it)
        char bad[] = "/tmp/jda_semantics_bad_XXXXXX";
        const int fd = mkstemp(bad);
        CHECK(27, fd >= 0);
        std::vector<uint8_t> junk(4096, 0x11);
        FILE *bf = fdopen(fd, "wb");
        fwrite(junk.data(), 1, junk.size(), bf);
        fclose(bf);
        for (int k = 0; k < 200; k++) {
            JPEGIMAGE fi;
            memset(&fi, 0x3C, sizeof(fi));
            CHECK(28, JPEG_openFile(&fi, bad, draw) == 0 && fi.file_data == NULL && JPEG_getLastError(&fi) != JPEG_SUCCESS);
        }
        remove(bad);
    }
    printf("ok\n");
    return 0;
}
