// tests/fuzz/frontend_fuzz.cpp -- header / scan mutations through the host front end under ASan + UBSan.
// TEST INFRASTRUCTURE (built by `make frontfuzz` with -fsanitize=address,undefined from jda_frontend.cpp alone: the front end
// has no GPU code).  The reference's fuzz idea (MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp:262-300) aimed at the part of this
// path that reads untrusted bytes on the host: marker walk, DHT/DQT/SOF/SOS/EXIF parsing, LUT construction, scan filter,
// serial pre-scan, draw plan.  usage: frontend_fuzz file.jpg [iterations] [seed]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "jpegdec_amd.h"

static uint32_t rng_state = 1;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void run(const std::vector<uint8_t> &b, long *ok, long *rejected)
{
    jda_image_info info;
    (void)jda_parse(b.data(), (int32_t)b.size(), &info);
    int32_t err = 0;
    jda_image *img = jda_prepare_ex(b.data(), (int32_t)b.size(), ((rnd() & 1) ? JDA_PREPARE_DEVICE_PRESCAN : 0) | ((rnd() & 1) ? JDA_PREPARE_PARALLEL_PRESCAN : 0), &err);
    if (!img) { (*rejected)++; return; }
    (*ok)++;
    int32_t rects[8 * 64];
    int32_t bpp, ow, oh, cw, ch;
    const jda_image_info *I = jda_image_get_info(img);
    for (int pt = 0; pt < 4; pt++) {
        if (jda_output_geometry(I, pt, (int32_t)(rnd() & 0x4e), &bpp, &ow, &oh, &cw, &ch) != JDA_SUCCESS) continue;
        int32_t crop[4] = { (int32_t)(rnd() % 700), (int32_t)(rnd() % 500), (int32_t)(rnd() % 700), (int32_t)(rnd() % 500) };
        jda_crop_round(I, &crop[0], &crop[1], &crop[2], &crop[3]);
        (void)jda_draw_plan_ex(I, pt, 0, (int32_t)(rnd() % 40), (int32_t)(rnd() & 1), (rnd() & 1) ? crop : NULL, rects, 64);
    }
    uint32_t n = 0;
    (void)jda_image_scan(img, &n);
    (void)jda_image_truncation_events(img);
    jda_image_free(img);
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> base;
    fseek(f, 0, SEEK_END); base.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
    if (fread(base.data(), 1, base.size(), f) != base.size()) return 2;
    fclose(f);
    const long iters = argc > 2 ? atol(argv[2]) : 2000;
    rng_state = argc > 3 ? (uint32_t)atol(argv[3]) : 1u;
    size_t sos = 0;
    for (size_t i = 0; i + 1 < base.size(); i++) if (base[i] == 0xff && base[i + 1] == 0xda) { sos = i; break; }
    const size_t header = sos ? sos + 14 : (base.size() < 1024 ? base.size() : 1024);
    long ok = 0, rejected = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> b = base;
        const int kind = (int)(rnd() % 6);
        const int n = 1 + (int)(rnd() % 3);
        for (int k = 0; k < n; k++) {
            const size_t at = kind < 4 ? rnd() % header : rnd() % b.size();      // mostly the header: DHT / DQT / SOF / SOS / APPn
            switch (kind) {
            case 0: b[at] = (uint8_t)rnd(); break;
            case 1: b[at] = (uint8_t)~b[at]; break;
            case 2: b[at] = 0xff; break;
            case 3: b[at] = (uint8_t)(b[at] + 1 + rnd() % 3); break;
            default: b[at] = (uint8_t)rnd(); break;
            }
        }
        if (rnd() % 16 == 0) b.resize(256 + rnd() % (b.size() - 256 + 1));     // truncated file
        run(b, &ok, &rejected);
    }
    printf("frontend_fuzz: %ld mutations, %ld prepared, %ld rejected\n", iters, ok, rejected);
    return 0;
}
