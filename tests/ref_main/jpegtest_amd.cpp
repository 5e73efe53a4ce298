// tests/ref_main/jpegtest_amd.cpp -- the reference's own test program, restated against the product.
//
// What MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp checks (tests 1, 2, 4-10 at :74-260 and the two fuzz loops at :262-300),
// written again on top of include/JPEGDEC.h + libjpegdec_amd.so.  The reference compiles its test vectors in as C arrays;
// here the same bytes are read from tests/golden/ref/*.jpg (argv[1]).  Test 3 (:139-164) compares the CPU time of a
// luma-only decode with a colour decode and has no meaning on this path; it is left out and the total says 11, not 12.
// Exit code 0 = every test passed.  TEST INFRASTRUCTURE: built by `make jpegtest`, run by tests/test_gpu_ref_fixtures.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "JPEGDEC.h"

static JPEGDEC jpg;
static int g_x1, g_y1, g_x2, g_y2;
static uint16_t *g_last_pixels;
static int g_dma_failed;
static int g_total, g_pass, g_fail;

static std::vector<uint8_t> load(const std::string &dir, const char *name)
{
    std::vector<uint8_t> v;
    FILE *f = fopen((dir + "/" + name + ".jpg").c_str(), "rb");
    if (!f) { printf("cannot open fixture %s\n", name); exit(2); }
    fseek(f, 0, SEEK_END);
    v.resize((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    if (fread(v.data(), 1, v.size(), f) != v.size()) { printf("short read %s\n", name); exit(2); }
    fclose(f);
    return v;
}

static void verdict(const char *what, bool ok, const char *why = "")
{
    printf("%-44s %s %s\n", what, ok ? "- PASSED" : "- FAILED", ok ? "" : why);
    g_total++;
    if (ok) g_pass++; else g_fail++;
}

// main.cpp:53-67: the callback records the extent of what was drawn and whether the pixel pointer alternates
static int draw(JPEGDRAW *d)
{
    if (d->pPixels == g_last_pixels) g_dma_failed = 1;
    g_last_pixels = d->pPixels;
    if (d->x < g_x1) g_x1 = d->x;
    if (d->y < g_y1) g_y1 = d->y;
    if (d->x + d->iWidthUsed - 1 > g_x2) g_x2 = d->x + d->iWidthUsed - 1;
    if (d->y + d->iHeight - 1 > g_y2) g_y2 = d->y + d->iHeight - 1;
    return 1;
}

static void reset_extent() { g_x1 = g_y1 = 1000000; g_x2 = g_y2 = 0; }

// tests 4-8 (:166-216): a corrupt file must not crash; whatever open/decode return is fine
static void survive(const char *what, std::vector<uint8_t> &v)
{
    if (jpg.openFLASH(v.data(), (int)v.size(), draw)) {
        jpg.decode(0, 0, 0);
        jpg.close();
    }
    verdict(what, true);
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/ref";
    std::vector<uint8_t> tulips = load(dir, "tulips"), thumb = load(dir, "thumb_test");
    std::vector<uint8_t> c1 = load(dir, "corrupt1"), c2 = load(dir, "corrupt2"), c3 = load(dir, "corrupt3"), c4 = load(dir, "corrupt4"), c5 = load(dir, "corrupt5");

    // test 1 (:74-105): a full-size decode draws exactly width x height
    reset_extent();
    if (jpg.openFLASH(tulips.data(), (int)tulips.size(), draw)) {
        if (jpg.decode(0, 0, 0)) {
            const int w = 1 + g_x2 - g_x1, h = 1 + g_y2 - g_y1;
            verdict("JPEG full image decode", w == jpg.getWidth() && h == jpg.getHeight(), "drawn extent != image size");
        } else verdict("JPEG full image decode", false, "decode failed");
        jpg.close();
    } else verdict("JPEG full image decode", false, "open failed");

    // test 2 (:107-137): a cropped decode draws exactly the (MCU-adjusted) crop rectangle
    if (jpg.openFLASH(tulips.data(), (int)tulips.size(), draw)) {
        int cx, cy, cw, ch;
        jpg.setCropArea(50, 50, 125, 170);
        jpg.getCropArea(&cx, &cy, &cw, &ch);
        reset_extent();
        if (jpg.decode(0, 0, 0)) {
            const int w = 1 + g_x2 - g_x1, h = 1 + g_y2 - g_y1;
            verdict("JPEG cropped image decode", w == cw && h == ch, "drawn extent != crop rectangle");
        } else verdict("JPEG cropped image decode", false, "decode failed");
        jpg.close();
    } else verdict("JPEG cropped image decode", false, "open failed");

    survive("JPEG purposely corrupt image 1", c1);
    survive("JPEG purposely corrupt image 2", c2);
    survive("JPEG purposely corrupt image 3", c3);
    survive("JPEG purposely corrupt image 4", c4);
    survive("JPEG purposely corrupt image 5", c5);

    // test 9 (:218-234): JPEG_USES_DMA hands out alternating halves of the pixel buffer
    g_dma_failed = 0; g_last_pixels = NULL;
    if (jpg.openFLASH(tulips.data(), (int)tulips.size(), draw)) {
        jpg.decode(0, 0, JPEG_USES_DMA);
        jpg.close();
    }
    verdict("JPEG DMA ping-pong buffer", !g_dma_failed, "two consecutive callbacks got the same buffer");

    // test 10 (:236-260): the EXIF thumbnail is found and decoding it leaves a 320x240 image in the object
    if (jpg.openFLASH(thumb.data(), (int)thumb.size(), draw)) {
        if (jpg.hasThumb()) {
            jpg.decode(0, 0, JPEG_EXIF_THUMBNAIL);
            jpg.close();
            verdict("JPEG EXIF Thumbnail", jpg.getWidth() == 320 && jpg.getHeight() == 240, "thumbnail not decoded");
        } else verdict("JPEG EXIF Thumbnail", false, "thumbnail not detected");
    } else verdict("JPEG EXIF Thumbnail", false, "open failed");

    // fuzz 1 (:262-280): every one of the first 2000 bytes inverted in turn -- header and the start of the scan
    std::vector<uint8_t> fz(tulips.size());
    const int n_seq = getenv("JPEGTEST_FUZZ_BYTES") ? atoi(getenv("JPEGTEST_FUZZ_BYTES")) : 2000;
    for (int i = 0; i < n_seq && i < (int)tulips.size(); i++) {
        memcpy(fz.data(), tulips.data(), tulips.size());
        fz[(size_t)i] = (uint8_t)~fz[(size_t)i];
        if (jpg.openFLASH(fz.data(), (int)fz.size(), draw)) { jpg.decode(0, 0, 0); jpg.close(); }
    }
    verdict("Single Byte Sequential Corruption Test", true);

    // fuzz 2 (:282-298): two random bytes overwritten, 1000 times.  (The reference opens the pristine array by mistake;
    // this one opens the corrupted copy, which is what its comment says it means to do.)
    srand(1);
    for (int i = 0; i < 1000; i++) {
        memcpy(fz.data(), tulips.data(), tulips.size());
        fz[(size_t)rand() % fz.size()] = (uint8_t)rand();
        fz[(size_t)rand() % fz.size()] = (uint8_t)rand();
        if (jpg.openFLASH(fz.data(), (int)fz.size(), draw)) { jpg.decode(0, 0, 0); jpg.close(); }
    }
    verdict("Multi-Byte Random Corruption Test", true);

    printf("Total tests: %d, %d passed, %d failed\n", g_total, g_pass, g_fail);
    return g_fail ? 1 : 0;
}
