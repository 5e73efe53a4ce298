// tests/fuzz/chunk_equiv.cpp -- the chunk-parallel host pre-scan against the serial one at chunk sizes the thread count of a test machine
// never produces.  TEST INFRASTRUCTURE (built by `make chunkequiv` from jda_frontend.cpp alone with -DJDA_TEST_CHUNK_BYTES_HOOK, which
// exposes the chunk size as a variable: the product library has no such switch -- there the size follows from the scan and the threads).
// For every file given: chunks of 512 .. 16 K bytes, the file as it is and corrupted copies (bytes of the entropy-coded data changed):
// the same verdict as the serial pre-scan and, where both index the whole image, the same index in the sense of jda_index_equivalent,
// DC values, truncation count and continuation entries.  usage: chunk_equiv iterations seed file.jpg..
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "jpegdec_amd.h"

extern uint32_t jda_test_chunk_bytes;            // jda_frontend.cpp under JDA_TEST_CHUNK_BYTES_HOOK (0: the product's rule)
extern uint32_t jda_test_chunk_taken;            // counts the images host_prescan_chunks indexed
extern std::atomic<int> jda_test_throw_countdown;   // the n-th allocation point a pre-scan worker passes throws std::bad_alloc (0: never)
extern "C" int jda_host_prescan_threads(void);   // the caller + the helper threads (1: a machine with fewer than four CPUs -- nothing runs in chunks there)

static uint32_t rng_state = 1;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static bool same(const std::vector<uint8_t> &b, int32_t flags, const char *what)
{
    int32_t ea = 0, eb = 0;
    jda_image *a = jda_prepare_ex(b.data(), (int32_t)b.size(), flags | JDA_PREPARE_SERIAL_PRESCAN, &ea);
    jda_image *p = jda_prepare_ex(b.data(), (int32_t)b.size(), flags | JDA_PREPARE_PARALLEL_PRESCAN, &eb);
    bool ok = true;
    if (!a || !p) ok = !a && !p && ea == eb;
    else {
        const jda_image_info *I = jda_image_get_info(a);
        const uint32_t nb = (uint32_t)(I->mcus_x * I->mcus_y * I->blocks_per_mcu);
        uint32_t na = 0, np = 0, ca = 0, cp = 0;
        const uint32_t *ia = jda_image_block_index(a, &na), *ip = jda_image_block_index(p, &np);
        const uint32_t *fa = NULL, *fp = NULL;
        const uint32_t *xa = jda_image_block_cont(a, &fa, &ca), *xp = jda_image_block_cont(p, &fp, &cp);
        ok = na == np && jda_image_truncation_events(a) == jda_image_truncation_events(p) && ca == cp &&
             !memcmp(jda_image_block_dc(a), jda_image_block_dc(p), (size_t)nb * 2) && !memcmp(fa, fp, ((size_t)nb + 1) * 4) && (ca == 0 || !memcmp(xa, xp, (size_t)ca * 4));
        if (ok) ok = na == (uint32_t)(I->mcus_x * I->mcus_y) ? jda_index_equivalent(ia, ip, nb) != 0 : !memcmp(ia, ip, ((size_t)nb + 1) * 4);
    }
    if (!ok) fprintf(stderr, "chunk_equiv: %s differs (chunk bytes %u, flags %d)\n", what, jda_test_chunk_bytes, flags);
    if (a) jda_image_free(a);
    if (p) jda_image_free(p);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const long iters = atol(argv[1]);
    rng_state = (uint32_t)atol(argv[2]);
    static const uint32_t sizes[] = { 512, 1024, 2048, 4096, 8192, 16384 };
    long checked = 0;
    for (int fi = 3; fi < argc; fi++) {
        FILE *f = fopen(argv[fi], "rb");
        if (!f) return 2;
        std::vector<uint8_t> base;
        fseek(f, 0, SEEK_END); base.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
        if (fread(base.data(), 1, base.size(), f) != base.size()) return 2;
        fclose(f);
        size_t sos = 0;
        for (size_t i = 0; i + 1 < base.size(); i++) if (base[i] == 0xff && base[i + 1] == 0xda) { sos = i + 14; break; }
        if (!sos || sos + 16 >= base.size()) return 2;
        for (uint32_t cs : sizes) {
            jda_test_chunk_bytes = cs;
            for (int32_t flags : { 0, (int32_t)JDA_PREPARE_CONT_ALWAYS }) { if (!same(base, flags, argv[fi])) return 1; checked++; }
            for (long it = 0; it < iters; it++) {
                std::vector<uint8_t> b = base;
                const int n = 1 + (int)(rnd() % 3);
                for (int k = 0; k < n; k++) {
                    const size_t at = sos + rnd() % (b.size() - sos - 2);
                    switch (rnd() % 3) {
                    case 0: b[at] ^= (uint8_t)(1u << (rnd() & 7)); break;
                    case 1: b[at] = (uint8_t)rnd(); break;
                    default: b[at] = 0xff; b[at + 1] = 0x00; break;          // (a stuffed FF: the filtered scan keeps its length, the symbols change)
                    }
                }
                if (!same(b, (rnd() & 1) ? JDA_PREPARE_CONT_ALWAYS : 0, argv[fi])) return 1;
                checked++;
            }
        }
    }
    // an allocation that fails inside a job of the helper pool -- on the caller's thread or on a helper's, early or late in the walk:
    // the job is closed, nobody is left inside it, and the image comes out as the serial pre-scan makes it (RstPool::run's try / catch)
    long thrown = 0;
    for (int fi = 3; fi < argc && fi < 9; fi++) {
        FILE *f = fopen(argv[fi], "rb");
        if (!f) return 2;
        std::vector<uint8_t> base;
        fseek(f, 0, SEEK_END); base.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
        if (fread(base.data(), 1, base.size(), f) != base.size()) return 2;
        fclose(f);
        for (uint32_t cs : { 1024u, 4096u }) {
            jda_test_chunk_bytes = cs;
            for (int at : { 1, 2, 7, 40, 300, 2000, 9000 }) {
                jda_test_throw_countdown.store(at);
                const bool ok = same(base, 0, argv[fi]);
                if (jda_test_throw_countdown.load() <= 0) thrown++;      // (the point was reached: somebody threw)
                jda_test_throw_countdown.store(0);
                if (!ok) { fprintf(stderr, "chunk_equiv: after a thrown allocation (point %d)\n", at); return 1; }
                if (!same(base, 0, argv[fi])) return 1;                  // .. and the pool works again afterwards
                checked += 2;
            }
        }
    }
    if (jda_host_prescan_threads() > 1 && thrown == 0) { fprintf(stderr, "chunk_equiv: no allocation point was ever reached\n"); return 1; }
    printf("chunk_equiv: %ld allocation failures injected and survived\n", thrown);
    printf("chunk_equiv: %ld comparisons, %u images indexed by the chunk-parallel pre-scan on %d threads\n", checked, jda_test_chunk_taken, jda_host_prescan_threads());
    return 0;
}
