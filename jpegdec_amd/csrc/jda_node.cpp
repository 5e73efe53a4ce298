// jda_node -- one host process, every GPU of the node (include/jpegdec_amd.h, "the node").
//
// Images are independent units of work -- the reference zeroes its whole decoder state per image (src/JPEGDEC.cpp:66) -- so a node
// shards a LIST of files by image and nothing else: jda_node owns one context and one streamed pipeline (jda_pipeline: device
// filter + pre-scan + decode) per device, deals a submitted list out in contiguous blocks (the rule bench.py's ranks use:
// jda_node_shard), and a submit runs the devices' host halves on a thread each.  No byte of pixel data crosses between GPUs: a
// device's block is decoded into surfaces on that device (allocate them on jda_node_context(k)); what the caller gathers is status
// words and, for a proof that every image was decoded exactly once and identically wherever it landed, per-image checksums made
// where the pixels are (jda_node_checksums).  Host code above the public C-ABI only: no kernel lives here.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <thread>
#include <vector>

#include "jpegdec_amd.h"

struct jda_node {
    struct Dev { int32_t device; jda_ctx *ctx; jda_pipeline *pipe; };
    std::vector<Dev> devs;
    int32_t max_images, depth;
    struct Slot { bool in_flight; int32_t ticket, n; std::vector<int32_t> dev_ticket, dev_rc; };
    std::vector<Slot> slots;
    int32_t next_ticket;
};

extern "C" {

void jda_node_shard_of(int32_t n_devices, int32_t n, int32_t k, int32_t *first, int32_t *count)
{
    if (n_devices <= 0 || k < 0 || k >= n_devices || n < 0) { if (first) *first = 0; if (count) *count = 0; return; }
    const int32_t base = n / n_devices, extra = n % n_devices;
    if (first) *first = k * base + (k < extra ? k : extra);
    if (count) *count = base + (k < extra ? 1 : 0);
}

jda_node *jda_node_create(const int32_t *devices, int32_t n_devices, int32_t max_images_per_device, int32_t depth, int32_t host_threads_per_device, int32_t *err)
{
    int32_t dummy;
    if (!err) err = &dummy;
    *err = JDA_SUCCESS;
    const int visible = jda_device_count();
    if (visible <= 0) { *err = JDA_ERROR_NO_DEVICE; return NULL; }            // there is no CPU decode path
    if (n_devices <= 0) { n_devices = visible; devices = NULL; }
    if (max_images_per_device <= 0 || depth < 1 || depth > 4) { *err = JDA_INVALID_PARAMETER; return NULL; }
    jda_node *nd = new (std::nothrow) jda_node;
    if (!nd) { *err = JDA_ERROR_MEMORY; return NULL; }
    nd->max_images = max_images_per_device; nd->depth = depth; nd->next_ticket = 0;
    for (int32_t k = 0; k < n_devices; k++) {
        jda_node::Dev d;
        d.device = devices ? devices[k] : k; d.ctx = NULL; d.pipe = NULL;
        for (const jda_node::Dev &o : nd->devs) if (o.device == d.device) *err = JDA_INVALID_PARAMETER;      // a device twice
        if (*err == JDA_SUCCESS) d.ctx = jda_create(d.device, err);
        if (d.ctx) d.pipe = jda_pipeline_create(d.ctx, max_images_per_device, depth, host_threads_per_device, err);
        nd->devs.push_back(d);
        if (!d.pipe) { if (*err == JDA_SUCCESS) *err = JDA_ERROR_NO_DEVICE; jda_node_destroy(nd); return NULL; }
    }
    nd->slots.resize((size_t)depth);
    for (jda_node::Slot &s : nd->slots) { s.in_flight = false; s.ticket = -1; s.n = 0; }
    return nd;
}

void jda_node_destroy(jda_node *nd)
{
    if (!nd) return;
    for (jda_node::Dev &d : nd->devs) {
        if (d.pipe) jda_pipeline_destroy(d.pipe);
        if (d.ctx) jda_destroy(d.ctx);
    }
    delete nd;
}

int32_t jda_node_device_count(const jda_node *nd) { return nd ? (int32_t)nd->devs.size() : 0; }
jda_ctx *jda_node_context(jda_node *nd, int32_t k) { return (nd && k >= 0 && k < (int32_t)nd->devs.size()) ? nd->devs[(size_t)k].ctx : NULL; }
int32_t jda_node_device(const jda_node *nd, int32_t k) { return (nd && k >= 0 && k < (int32_t)nd->devs.size()) ? nd->devs[(size_t)k].device : -1; }
void jda_node_shard(const jda_node *nd, int32_t n, int32_t k, int32_t *first, int32_t *count) { jda_node_shard_of(jda_node_device_count(nd), n, k, first, count); }

int jda_node_submit(jda_node *nd, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                    const int32_t *pixel_types, const int32_t *options, int32_t *ticket)
{
    if (!nd) return JDA_ERROR_NO_DEVICE;
    const int32_t nk = (int32_t)nd->devs.size();
    if (n <= 0 || !jpegs || !lens || !outputs || !pixel_types || !options || !ticket) return JDA_INVALID_PARAMETER;
    if ((int64_t)n > (int64_t)nd->max_images * nk) return JDA_INVALID_PARAMETER;
    jda_node::Slot &S = nd->slots[(size_t)(nd->next_ticket % nd->depth)];
    if (S.in_flight) return JDA_INVALID_PARAMETER;                           // wait for the batch `depth` submits ago first
    S.dev_ticket.assign((size_t)nk, -1); S.dev_rc.assign((size_t)nk, JDA_SUCCESS); S.n = n;
    // the devices' host halves (header parse, tables, the copy into page-locked memory, the launches) side by side
    auto one = [&](int32_t k) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, n, k, &first, &count);
        if (count == 0) return;
        S.dev_rc[(size_t)k] = jda_pipeline_submit(nd->devs[(size_t)k].pipe, count, jpegs + first, lens + first, outputs + first, pixel_types + first,
                                                  options + first, &S.dev_ticket[(size_t)k]);
    };
    std::vector<std::thread> th;
    for (int32_t k = 1; k < nk; k++) th.emplace_back(one, k);
    one(0);
    for (std::thread &t : th) t.join();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) if (S.dev_rc[(size_t)k] != JDA_SUCCESS && rc == JDA_SUCCESS) rc = S.dev_rc[(size_t)k];
    if (rc != JDA_SUCCESS) {                                                 // a device refused its block: nothing of this list stays in flight
        for (int32_t k = 0; k < nk; k++) if (S.dev_rc[(size_t)k] == JDA_SUCCESS && S.dev_ticket[(size_t)k] >= 0) (void)jda_pipeline_wait(nd->devs[(size_t)k].pipe, S.dev_ticket[(size_t)k], NULL);
        return rc;
    }
    S.in_flight = true; S.ticket = nd->next_ticket;
    *ticket = nd->next_ticket++;
    return JDA_SUCCESS;
}

int jda_node_wait(jda_node *nd, int32_t ticket, int32_t *status)
{
    if (!nd) return JDA_ERROR_NO_DEVICE;
    if (ticket < 0 || ticket >= nd->next_ticket) return JDA_INVALID_PARAMETER;
    jda_node::Slot &S = nd->slots[(size_t)(ticket % nd->depth)];
    if (!S.in_flight || S.ticket != ticket) return JDA_INVALID_PARAMETER;
    const int32_t nk = (int32_t)nd->devs.size();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, S.n, k, &first, &count);
        if (count == 0) continue;
        const int r = jda_pipeline_wait(nd->devs[(size_t)k].pipe, S.dev_ticket[(size_t)k], status ? status + first : NULL);
        if (r != JDA_SUCCESS && rc == JDA_SUCCESS) rc = r;
    }
    S.in_flight = false;
    return rc;
}

int jda_node_checksums(jda_node *nd, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *checksums)
{
    if (!nd) return JDA_ERROR_NO_DEVICE;
    if (n <= 0 || !surfaces || !row_bytes || !checksums) return JDA_INVALID_PARAMETER;
    const int32_t nk = (int32_t)nd->devs.size();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, n, k, &first, &count);
        if (count == 0) continue;
        const int r = jda_checksum_surfaces(nd->devs[(size_t)k].ctx, count, surfaces + first, row_bytes + first, checksums + first);
        if (r != JDA_SUCCESS && rc == JDA_SUCCESS) rc = r;
    }
    return rc;
}

int jda_node_get_stats(const jda_node *nd, jda_pipeline_stats *out)
{
    if (!nd || !out) return JDA_INVALID_PARAMETER;
    memset(out, 0, sizeof(*out));
    for (const jda_node::Dev &d : nd->devs) {
        jda_pipeline_stats s;
        if (jda_pipeline_get_stats(d.pipe, &s) != JDA_SUCCESS) continue;
        out->images += s.images; out->device_images += s.device_images; out->host_path_images += s.host_path_images; out->failed_images += s.failed_images;
        out->source_pixels += s.source_pixels; out->compressed_bytes += s.compressed_bytes; out->h2d_bytes += s.h2d_bytes; out->launches += s.launches;
        if (s.spec_rounds_max > out->spec_rounds_max) out->spec_rounds_max = s.spec_rounds_max;
    }
    return JDA_SUCCESS;
}

} // extern "C"
