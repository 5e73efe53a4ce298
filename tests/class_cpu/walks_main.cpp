// tests/class_cpu/walks_main.cpp -- TEST INFRASTRUCTURE ONLY.  Runs recorded walks (tests/test_class_walks_cpu.py writes them as text)
// through the class's host logic under AddressSanitizer / UBSan.  Results are not compared here (the non-sanitizer build does that).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

extern "C" {
int ref_decode_cb2(const uint8_t *data, int len, int pixel_type, int options, int max_mcus, int xoff, int yoff, const int *crop,
                   uint8_t *canvas, int pitch_bytes, int rows, int used_only, int *log, int max_log, int stop_after,
                   int *n_calls, int *dma_reuse, int *last_error, int *after);
int ref_decode_fb_crop(const uint8_t *data, int len, int pixel_type, int options, const int *crop, void *fb, int *last_error);
int ref_run_script(const uint8_t *data, int len, const int *ops, int n_ops, uint8_t *canvas, int pitch_bytes, int rows, int *out, int max_out);
int ref_get_info(const uint8_t *data, int len, int *info);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    const int n_img = atoi(argv[2]);
    std::vector<std::vector<uint8_t> > img((size_t)n_img);
    for (int k = 0; k < n_img; k++) {
        FILE *f = fopen((dir + "/img" + std::to_string(k) + ".jpg").c_str(), "rb");
        if (!f) return 3;
        fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
        img[(size_t)k].resize((size_t)n);
        if (fread(img[(size_t)k].data(), 1, (size_t)n, f) != (size_t)n) return 3;
        fclose(f);
    }
    FILE *w = fopen((dir + "/walks.txt").c_str(), "r");
    if (!w) return 4;
    std::vector<uint8_t> canvas((size_t)1200 * 8192 * 2);
    std::vector<int> log(6 * 65536);
    char kind;
    int done = 0;
    while (fscanf(w, " %c", &kind) == 1) {
        if (kind == 'W') {
            int im, fb, pt, opt, mm, xo, yo, c[4];
            if (fscanf(w, "%d %d %d %d %d %d %d %d %d %d %d", &im, &fb, &pt, &opt, &mm, &xo, &yo, &c[0], &c[1], &c[2], &c[3]) != 11) return 5;
            const std::vector<uint8_t> &j = img[(size_t)im];
            const int *crop = c[0] < 0 ? NULL : c;
            int info[10], err = 0, calls = 0, dma = 0, after[2];
            if (!ref_get_info(j.data(), (int)j.size(), info)) { done++; continue; }
            if (fb) {
                // as oracle/loader.py sizes it: MCU-padded rows + one MCU + 8, width * bpp + 64
                const size_t bytes = ((size_t)info[1] + 64 + 8) * ((size_t)info[0] * 4 + 64);
                std::vector<uint8_t> buf(bytes, 0);
                (void)ref_decode_fb_crop(j.data(), (int)j.size(), pt, opt, crop, buf.data(), &err);
            } else {
                const int rows = 4200, pitch = (info[0] + xo + 2048) * 4 + 64;
                if ((size_t)rows * pitch > canvas.size()) canvas.resize((size_t)rows * pitch);
                memset(canvas.data(), 0, (size_t)rows * pitch);
                (void)ref_decode_cb2(j.data(), (int)j.size(), pt, opt, mm, xo, yo, crop, canvas.data(), pitch, rows, 1, log.data(), 65536, 0, &calls, &dma, &err, after);
            }
        } else if (kind == 'S') {
            int im, n_ops;
            if (fscanf(w, "%d %d", &im, &n_ops) != 2) return 5;
            std::vector<int> ops((size_t)n_ops * 5);
            for (size_t i = 0; i < ops.size(); i++) if (fscanf(w, "%d", &ops[i]) != 1) return 5;
            const std::vector<uint8_t> &j = img[(size_t)im];
            int out[4096];
            if ((size_t)1200 * 8192 > canvas.size()) canvas.resize((size_t)1200 * 8192);
            (void)ref_run_script(j.data(), (int)j.size(), ops.data(), n_ops, canvas.data(), 8192, 1200, out, 4096);
        } else return 6;
        done++;
    }
    fclose(w);
    printf("%d walks done\n", done);
    return 0;
}
