// tests/node_stub/node_stub_user.cpp -- TEST INFRASTRUCTURE: jda_node over eight pretend devices (stub_pipeline.cpp): the shard rule,
// one persistent thread per device (never the caller's), statuses in list order, depth, a device that refuses its block.
// Prints "ok" and exits 0, or the number of the check that failed.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "jpegdec_amd.h"

extern "C" int stub_wrong_thread_calls(void);
extern "C" int stub_calls_on_main(void);
extern "C" int stub_device_thread_hash(int k);
extern "C" int stub_pipeline_threads(int k);
#define CHECK(n, cond) do { if (!(cond)) { printf("check %d failed\n", n); return n; } } while (0)

int main()
{
    int32_t err = -1;
    jda_node *nd = jda_node_create(NULL, 0, 16, 2, 0, &err);
    CHECK(1, nd && err == JDA_SUCCESS && jda_node_device_count(nd) == 8);
    for (int k = 0; k < 8; k++) {
        int32_t numa = 7, cpus = 7;
        CHECK(2, jda_node_device(nd, k) == k && jda_node_context(nd, k) != NULL && jda_node_placement(nd, k, &numa, &cpus) == JDA_SUCCESS);
        CHECK(3, numa == -1 && cpus == 0);                          // (the pretend bus ids are in no sysfs: nothing is pinned)
    }
    int thread_of[8];
    for (int k = 0; k < 8; k++) thread_of[k] = stub_device_thread_hash(k);
    for (int k = 0; k < 8; k++) for (int j = 0; j < k; j++) CHECK(4, thread_of[k] != thread_of[j]);      // a thread per device
    const int N = 100;                                              // 100 files over 8 devices: 13 13 13 13 12 12 12 12
    std::vector<std::vector<uint8_t> > files((size_t)N, std::vector<uint8_t>(8, 0));
    std::vector<const uint8_t *> ptrs((size_t)N);
    std::vector<int32_t> lens((size_t)N, 8), pts((size_t)N, JDA_RGB8888), opts((size_t)N, 0), status((size_t)N, -1);
    std::vector<uint64_t> surf((size_t)N, 0), sums((size_t)N, 0);
    std::vector<jda_output> outs((size_t)N);
    std::vector<int32_t> rowb((size_t)N, 8);
    for (int round = 0; round < 5; round++) {
        for (int i = 0; i < N; i++) {
            const uint32_t id = (uint32_t)(round * 1000 + i);
            files[(size_t)i][0] = (i % 17 == 3) ? 0xEE : 0x11;       // some files "fail to decode"
            memcpy(files[(size_t)i].data() + 2, &id, 4);
            ptrs[(size_t)i] = files[(size_t)i].data();
            outs[(size_t)i].pixels = &surf[(size_t)i]; outs[(size_t)i].pitch_bytes = 16; outs[(size_t)i].width_px = 2; outs[(size_t)i].rows = 1;
        }
        int32_t t = -1;
        CHECK(10, jda_node_submit_ex(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), round & 1 ? JDA_SUBMIT_PINNED_INPUT : 0, &t) == JDA_SUCCESS && t == round);
        CHECK(11, jda_node_wait(nd, t, status.data()) == JDA_SUCCESS);
        CHECK(12, jda_node_checksums(nd, N, outs.data(), rowb.data(), sums.data()) == JDA_SUCCESS);
        for (int k = 0; k < 8; k++) {
            int32_t first = 0, count = 0;
            jda_node_shard(nd, N, k, &first, &count);
            CHECK(13, count == (k < 4 ? 13 : 12) && first == (k < 4 ? 13 * k : 52 + 12 * (k - 4)));
            for (int i = first; i < first + count; i++) {
                CHECK(14, (uint32_t)sums[(size_t)i] == (uint32_t)(round * 1000 + i));                         // every file once, in its place ..
                CHECK(15, (uint32_t)(sums[(size_t)i] >> 32) == ((uint32_t)k | ((uint32_t)(round & 1) << 8)));   // .. on the device the rule gives it, with the caller's flags
                CHECK(16, status[(size_t)i] == (i % 17 == 3 ? JDA_DECODE_ERROR : JDA_SUCCESS));                // a bad file is its own status, where it stood in the list
            }
            CHECK(17, stub_device_thread_hash(k) == thread_of[k]);  // the same thread serves the device round after round
        }
    }
    CHECK(20, stub_wrong_thread_calls() == 0);                      // every call of a device came from its own thread ..
    CHECK(21, stub_calls_on_main() == 0);                           // .. and none from the caller's
    // depth 2: a third list without a wait is refused, nothing breaks
    int32_t t0 = -1, t1 = -1, t2 = -1;
    CHECK(30, jda_node_submit(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t0) == JDA_SUCCESS);
    CHECK(31, jda_node_submit(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t1) == JDA_SUCCESS);
    CHECK(32, jda_node_submit(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t2) == JDA_INVALID_PARAMETER);
    CHECK(33, jda_node_wait(nd, t0, NULL) == JDA_SUCCESS && jda_node_wait(nd, t1, status.data()) == JDA_SUCCESS);
    CHECK(34, jda_node_wait(nd, t1, NULL) == JDA_INVALID_PARAMETER);
    // a device that refuses its block: the error comes back, the other devices' blocks are waited out, the next list goes through
    files[60][0] = 0xFD;
    CHECK(40, jda_node_submit(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t2) == JDA_ERROR_MEMORY);
    files[60][0] = 0x11;
    CHECK(41, jda_node_submit(nd, N, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t2) == JDA_SUCCESS && jda_node_wait(nd, t2, status.data()) == JDA_SUCCESS);
    CHECK(42, jda_node_submit(nd, 8 * 16 + 1, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t2) == JDA_INVALID_PARAMETER);      // more than the node holds
    // a short list: devices without a share are left alone
    CHECK(43, jda_node_submit(nd, 3, ptrs.data(), lens.data(), outs.data(), pts.data(), opts.data(), &t2) == JDA_SUCCESS && jda_node_wait(nd, t2, status.data()) == JDA_SUCCESS);
    jda_pipeline_stats st;
    CHECK(44, jda_node_get_stats(nd, &st) == JDA_SUCCESS && st.images > 0);
    jda_node_destroy(nd);
    CHECK(50, stub_wrong_thread_calls() == 0 && stub_calls_on_main() == 0);
    // a subset of devices, explicit host threads
    const int32_t devs[3] = { 5, 2, 7 };
    nd = jda_node_create(devs, 3, 4, 1, 3, &err);
    CHECK(51, nd && jda_node_device_count(nd) == 3 && jda_node_device(nd, 1) == 2 && stub_pipeline_threads(2) == 3);
    jda_node_destroy(nd);
    const int32_t twice[2] = { 1, 1 };                 // a device named twice: two entries, each with its own pipeline on that device
    nd = jda_node_create(twice, 2, 4, 1, 0, &err);
    CHECK(52, nd && jda_node_device_count(nd) == 2 && jda_node_device(nd, 0) == 1 && jda_node_device(nd, 1) == 1);
    jda_node_destroy(nd);
    const int32_t beyond[1] = { 99 };
    CHECK(53, jda_node_create(beyond, 1, 4, 1, 0, &err) == NULL && err == JDA_INVALID_PARAMETER);
    printf("ok\n");
    return 0;
}
