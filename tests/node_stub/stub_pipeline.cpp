// tests/node_stub/stub_pipeline.cpp -- TEST INFRASTRUCTURE: a stand-in for the device half of the C-ABI (contexts, pipelines,
// checksums) with EIGHT pretend devices, so that jda_node.cpp -- host code above the public C-ABI only -- can be exercised without
// a GPU: who is called on which thread, which block of the list each device gets, how statuses and errors come back.
// "Decoding" writes a word derived from the file and the device into the output surface (host memory here).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "jpegdec_amd.h"

struct jda_ctx { int32_t device; std::thread::id owner; };
struct StubBatch { int32_t n; std::vector<int32_t> status; };
struct jda_pipeline { jda_ctx *ctx; int32_t max_images, depth, threads, next; std::vector<StubBatch> slots; std::thread::id owner; jda_pipeline_stats st; };

static std::mutex g_mu;
static int g_wrong_thread = 0;              // calls that did not come from the thread that created their context
static int g_calls_on_main = 0;
static std::thread::id g_main = std::this_thread::get_id();
static std::thread::id g_dev_thread[16];
static std::atomic<int32_t> g_pipeline_threads[16];      // (two entries of a node may sit on one device)

extern "C" {
int stub_wrong_thread_calls(void) { return g_wrong_thread; }
int stub_calls_on_main(void) { return g_calls_on_main; }
int stub_device_thread_hash(int k) { return (int)(std::hash<std::thread::id>()(g_dev_thread[k]) & 0x7fffffff); }
int stub_pipeline_threads(int k) { return g_pipeline_threads[k]; }

static void note(jda_ctx *c)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (std::this_thread::get_id() != c->owner) g_wrong_thread++;
    if (std::this_thread::get_id() == g_main) g_calls_on_main++;
}

int jda_device_count(void) { return 8; }
int jda_device_pci_bus_id_of(int32_t device, char *buf, int32_t len) { if (len < 16) return JDA_INVALID_PARAMETER; snprintf(buf, (size_t)len, "ffff:%02x:00.0", device); return JDA_SUCCESS; }
jda_ctx *jda_create(int32_t device, int32_t *err)
{
    if (device < 0 || device >= 8) { if (err) *err = JDA_ERROR_NO_DEVICE; return NULL; }
    jda_ctx *c = new jda_ctx;
    c->device = device; c->owner = std::this_thread::get_id();
    { std::lock_guard<std::mutex> lk(g_mu); g_dev_thread[device] = c->owner; if (c->owner == g_main) g_calls_on_main++; }
    if (err) *err = JDA_SUCCESS;
    return c;
}
void jda_destroy(jda_ctx *c) { if (c) { note(c); delete c; } }
jda_pipeline *jda_pipeline_create(jda_ctx *ctx, int32_t max_images, int32_t depth, int32_t host_threads, int32_t *err)
{
    note(ctx);
    jda_pipeline *p = new jda_pipeline;
    p->ctx = ctx; p->max_images = max_images; p->depth = depth; p->threads = host_threads; p->next = 0; p->slots.resize((size_t)depth);
    for (StubBatch &b : p->slots) b.n = -1;
    memset(&p->st, 0, sizeof(p->st));
    g_pipeline_threads[ctx->device] = host_threads;
    if (err) *err = JDA_SUCCESS;
    return p;
}
void jda_pipeline_destroy(jda_pipeline *p) { if (p) { note(p->ctx); delete p; } }
int jda_pipeline_submit_ex(jda_pipeline *p, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                           const int32_t *pixel_types, const int32_t *options, int32_t flags, int32_t *ticket)
{
    note(p->ctx);
    (void)pixel_types; (void)options;
    if (n <= 0 || n > p->max_images) return JDA_INVALID_PARAMETER;
    StubBatch &B = p->slots[(size_t)(p->next % p->depth)];
    if (B.n >= 0) return JDA_INVALID_PARAMETER;
    for (int32_t i = 0; i < n; i++) if (lens[i] >= 2 && jpegs[i][0] == 0xFD) return JDA_ERROR_MEMORY;       // a file this device chokes on: the whole block is refused
    B.n = n; B.status.assign((size_t)n, JDA_SUCCESS);
    for (int32_t i = 0; i < n; i++) {
        uint32_t id = 0;
        if (lens[i] >= 6) memcpy(&id, jpegs[i] + 2, 4);
        if (lens[i] >= 1 && jpegs[i][0] == 0xEE) B.status[(size_t)i] = JDA_DECODE_ERROR;
        uint32_t w[2] = { id, (uint32_t)p->ctx->device | ((uint32_t)flags << 8) };
        memcpy(outputs[i].pixels, w, 8);                             // "pixels": which file, decoded where, submitted how
    }
    p->st.images += n; p->st.device_images += n;
    *ticket = p->next++;
    return JDA_SUCCESS;
}
int jda_pipeline_wait(jda_pipeline *p, int32_t ticket, int32_t *status)
{
    note(p->ctx);
    StubBatch &B = p->slots[(size_t)(ticket % p->depth)];
    if (B.n < 0) return JDA_INVALID_PARAMETER;
    if (status) memcpy(status, B.status.data(), (size_t)B.n * sizeof(int32_t));
    for (int32_t s : B.status) if (s != JDA_SUCCESS) p->st.failed_images++;
    B.n = -1;
    return JDA_SUCCESS;
}
int jda_pipeline_get_stats(const jda_pipeline *p, jda_pipeline_stats *out) { *out = p->st; return JDA_SUCCESS; }
int jda_checksum_surfaces(jda_ctx *ctx, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *out)
{
    note(ctx);
    (void)row_bytes;
    for (int32_t i = 0; i < n; i++) { uint32_t w[2]; memcpy(w, surfaces[i].pixels, 8); out[i] = ((uint64_t)w[1] << 32) | w[0]; }
    return JDA_SUCCESS;
}
}
