// jda_node -- one host process, every GPU of the node (include/jpegdec_amd.h, "the node").
//
// Images are independent units of work -- the reference zeroes its whole decoder state per image (src/JPEGDEC.cpp:66) -- so a node
// shards a LIST of files by image and nothing else: jda_node owns one context and one streamed pipeline (jda_pipeline: device
// filter + pre-scan + decode) per device, deals a submitted list out in contiguous blocks (the rule bench.py's ranks use:
// jda_node_shard), and a submit runs the devices' host halves on a thread each.  No byte of pixel data crosses between GPUs: a
// device's block is decoded into surfaces on that device (allocate them on jda_node_context(k)); what the caller gathers is status
// words and, for a proof that every image was decoded exactly once and identically wherever it landed, per-image checksums made
// where the pixels are (jda_node_checksums).  Host code above the public C-ABI only: no kernel lives here.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "jpegdec_amd.h"

// One persistent host thread per device, created with the node and pinned to the CPUs of its GPU's NUMA node (devices that share
// a NUMA node deal its CPUs out among themselves).  The thread makes the device's context and pipeline -- so the pipeline's own
// workers inherit the placement and its page-locked buffers are first touched there -- and from then on runs that device's host
// half of every call (submit, wait, checksums): a submit wakes K threads instead of creating them, and the calling thread never
// makes a HIP call of the node's, so its own current device stays what it was.
namespace {
struct DevThread {
    int32_t device = -1;
    jda_ctx *ctx = NULL;
    jda_pipeline *pipe = NULL;
    int32_t numa_node = -1, cpus_pinned = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    void loop()
    {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this]() { return has_job || stop; });
                if (!has_job && stop) return;
                j.swap(job); has_job = false;
            }
            j();
            { std::lock_guard<std::mutex> lk(m); done = true; }
            cv_done.notify_all();
        }
    }
    void post(std::function<void()> j)
    {
        { std::lock_guard<std::mutex> lk(m); job.swap(j); has_job = true; done = false; }
        cv.notify_all();
    }
    void join_job() { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [this]() { return done; }); }
    void run(std::function<void()> j) { post(std::move(j)); join_job(); }
};

// "0-15,32-47" -> CPU numbers
std::vector<int> parse_cpulist(const char *s)
{
    std::vector<int> v;
    while (*s) {
        char *e;
        long a = strtol(s, &e, 10);
        if (e == s) break;
        long b = a;
        if (*e == '-') { s = e + 1; b = strtol(s, &e, 10); }
        for (long c = a; c <= b && c < 4096; c++) v.push_back((int)c);
        s = *e == ',' ? e + 1 : e;
        if (*e != ',' ) break;
    }
    return v;
}
int numa_node_of_pci(const char *bus_id)
{
    char path[128], low[32];
    size_t i = 0;
    for (; bus_id[i] && i < sizeof(low) - 1; i++) low[i] = (char)((bus_id[i] >= 'A' && bus_id[i] <= 'Z') ? bus_id[i] + 32 : bus_id[i]);
    low[i] = 0;
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", low);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int n = -1;
    if (fscanf(f, "%d", &n) != 1) n = -1;
    fclose(f);
    return n;
}
// pin the calling thread to share `j` of `m` of the CPUs of NUMA node `node` that the process may use; returns how many CPUs (0: left alone)
int pin_to_numa_share(int node, int j, int m)
{
    if (node < 0 || m <= 0) return 0;
    char path[96], buf[4096];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const bool got = fgets(buf, sizeof(buf), f) != NULL;
    fclose(f);
    if (!got) return 0;
    cpu_set_t allowed, mine;
    CPU_ZERO(&allowed); CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
    std::vector<int> cpus;
    for (int c : parse_cpulist(buf)) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) cpus.push_back(c);
    if ((int)cpus.size() < m) return 0;                              // fewer usable CPUs than devices on the node: leave the scheduler alone
    int n = 0;
    for (size_t i = (size_t)j; i < cpus.size(); i += (size_t)m) { CPU_SET(cpus[i], &mine); n++; }
    return sched_setaffinity(0, sizeof(mine), &mine) == 0 ? n : 0;
}
} // namespace

struct jda_node {
    std::vector<DevThread *> devs;
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
    if (max_images_per_device <= 0 || depth < 1 || depth > 8) { *err = JDA_INVALID_PARAMETER; return NULL; }
    // (a device may be named more than once: every entry gets a context, a pipeline and a host thread of its own -- several feeders of
    // one GPU, or a node's control flow rehearsed on the one GPU a box has)
    for (int32_t k = 0; k < n_devices; k++) { const int32_t o = devices ? devices[k] : k; if (o < 0 || o >= visible) { *err = JDA_INVALID_PARAMETER; return NULL; } }
    jda_node *nd = new (std::nothrow) jda_node;
    if (!nd) { *err = JDA_ERROR_MEMORY; return NULL; }
    nd->max_images = max_images_per_device; nd->depth = depth; nd->next_ticket = 0;
    // where the devices sit: the threads of devices on one NUMA node share its CPUs
    std::vector<int> numa((size_t)n_devices, -1);
    for (int32_t k = 0; k < n_devices; k++) {
        char bus[32] = { 0 };
        if (jda_device_pci_bus_id_of(devices ? devices[k] : k, bus, (int32_t)sizeof(bus)) == JDA_SUCCESS) numa[(size_t)k] = numa_node_of_pci(bus);
    }
    std::vector<int32_t> rc((size_t)n_devices, JDA_SUCCESS);
    for (int32_t k = 0; k < n_devices; k++) {
        DevThread *d = new (std::nothrow) DevThread;
        if (!d) { *err = JDA_ERROR_MEMORY; break; }
        d->device = devices ? devices[k] : k; d->numa_node = numa[(size_t)k];
        int j = 0, m = 0;
        for (int32_t o = 0; o < n_devices; o++) if (numa[(size_t)o] == d->numa_node) { if (o < k) j++; m++; }
        nd->devs.push_back(d);
        d->th = std::thread([d]() { d->loop(); });
        int32_t *rck = &rc[(size_t)k];
        d->post([d, j, m, max_images_per_device, depth, host_threads_per_device, rck]() {
            d->cpus_pinned = pin_to_numa_share(d->numa_node, j, m);
            int32_t e = JDA_SUCCESS;
            d->ctx = jda_create(d->device, &e);
            int32_t threads = host_threads_per_device;
            if (threads <= 0 && d->cpus_pinned > 0) threads = d->cpus_pinned < 8 ? d->cpus_pinned : 8;      // (the pipeline's own default: up to 8 of the CPUs it may use)
            if (d->ctx) d->pipe = jda_pipeline_create(d->ctx, max_images_per_device, depth, threads, &e);
            *rck = d->pipe ? JDA_SUCCESS : (e != JDA_SUCCESS ? e : JDA_ERROR_NO_DEVICE);
        });
    }
    for (DevThread *d : nd->devs) d->join_job();
    for (int32_t k = 0; k < (int32_t)nd->devs.size() && *err == JDA_SUCCESS; k++) *err = rc[(size_t)k];
    if (*err != JDA_SUCCESS || (int32_t)nd->devs.size() != n_devices) { if (*err == JDA_SUCCESS) *err = JDA_ERROR_MEMORY; jda_node_destroy(nd); return NULL; }
    nd->slots.resize((size_t)depth);
    for (jda_node::Slot &s : nd->slots) { s.in_flight = false; s.ticket = -1; s.n = 0; }
    return nd;
}

void jda_node_destroy(jda_node *nd)
{
    if (!nd) return;
    for (DevThread *d : nd->devs) {
        d->run([d]() {                                                   // on the device's own thread, like everything else of it
            if (d->pipe) jda_pipeline_destroy(d->pipe);
            if (d->ctx) jda_destroy(d->ctx);
            d->pipe = NULL; d->ctx = NULL;
        });
        { std::lock_guard<std::mutex> lk(d->m); d->stop = true; }
        d->cv.notify_all();
        if (d->th.joinable()) d->th.join();
        delete d;
    }
    delete nd;
}

int32_t jda_node_device_count(const jda_node *nd) { return nd ? (int32_t)nd->devs.size() : 0; }
jda_ctx *jda_node_context(jda_node *nd, int32_t k) { return (nd && k >= 0 && k < (int32_t)nd->devs.size()) ? nd->devs[(size_t)k]->ctx : NULL; }
int32_t jda_node_device(const jda_node *nd, int32_t k) { return (nd && k >= 0 && k < (int32_t)nd->devs.size()) ? nd->devs[(size_t)k]->device : -1; }
int jda_node_placement(const jda_node *nd, int32_t k, int32_t *numa_node, int32_t *cpus_pinned)
{
    if (!nd || k < 0 || k >= (int32_t)nd->devs.size()) return JDA_INVALID_PARAMETER;
    if (numa_node) *numa_node = nd->devs[(size_t)k]->numa_node;
    if (cpus_pinned) *cpus_pinned = nd->devs[(size_t)k]->cpus_pinned;
    return JDA_SUCCESS;
}
void jda_node_shard(const jda_node *nd, int32_t n, int32_t k, int32_t *first, int32_t *count) { jda_node_shard_of(jda_node_device_count(nd), n, k, first, count); }

int jda_node_submit(jda_node *nd, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                    const int32_t *pixel_types, const int32_t *options, int32_t *ticket)
{
    return jda_node_submit_ex(nd, n, jpegs, lens, outputs, pixel_types, options, 0, ticket);
}

int jda_node_submit_ex(jda_node *nd, int32_t n, const uint8_t *const *jpegs, const int32_t *lens, const jda_output *outputs,
                       const int32_t *pixel_types, const int32_t *options, int32_t flags, int32_t *ticket)
{
    if (!nd) return JDA_ERROR_NO_DEVICE;
    const int32_t nk = (int32_t)nd->devs.size();
    if (n <= 0 || !jpegs || !lens || !outputs || !pixel_types || !options || !ticket) return JDA_INVALID_PARAMETER;
    if ((int64_t)n > (int64_t)nd->max_images * nk) return JDA_INVALID_PARAMETER;
    jda_node::Slot &S = nd->slots[(size_t)(nd->next_ticket % nd->depth)];
    if (S.in_flight) return JDA_INVALID_PARAMETER;                           // wait for the batch `depth` submits ago first
    S.dev_ticket.assign((size_t)nk, -1); S.dev_rc.assign((size_t)nk, JDA_SUCCESS); S.n = n;
    // the devices' host halves (header parse, tables, what has to be copied into page-locked memory, the launches) side by side,
    // each on its device's thread
    for (int32_t k = 0; k < nk; k++) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, n, k, &first, &count);
        if (count == 0) continue;
        DevThread *d = nd->devs[(size_t)k];
        int32_t *rc = &S.dev_rc[(size_t)k], *tk = &S.dev_ticket[(size_t)k];
        d->post([=]() { *rc = jda_pipeline_submit_ex(d->pipe, count, jpegs + first, lens + first, outputs + first, pixel_types + first, options + first, flags, tk); });
    }
    for (DevThread *d : nd->devs) d->join_job();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) if (S.dev_rc[(size_t)k] != JDA_SUCCESS && rc == JDA_SUCCESS) rc = S.dev_rc[(size_t)k];
    if (rc != JDA_SUCCESS) {                                                 // a device refused its block: nothing of this list stays in flight
        for (int32_t k = 0; k < nk; k++) {
            if (S.dev_rc[(size_t)k] != JDA_SUCCESS || S.dev_ticket[(size_t)k] < 0) continue;
            DevThread *d = nd->devs[(size_t)k];
            const int32_t tk = S.dev_ticket[(size_t)k];
            d->run([d, tk]() { (void)jda_pipeline_wait(d->pipe, tk, NULL); });
        }
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
    std::vector<int32_t> rcs((size_t)nk, JDA_SUCCESS);
    for (int32_t k = 0; k < nk; k++) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, S.n, k, &first, &count);
        if (count == 0) continue;
        DevThread *d = nd->devs[(size_t)k];
        const int32_t tk = S.dev_ticket[(size_t)k];
        int32_t *rc = &rcs[(size_t)k], *st = status ? status + first : NULL;
        d->post([d, tk, rc, st]() { *rc = jda_pipeline_wait(d->pipe, tk, st); });      // (a device's redo of a bad image runs beside the other devices' waits)
    }
    for (DevThread *d : nd->devs) d->join_job();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) if (rcs[(size_t)k] != JDA_SUCCESS && rc == JDA_SUCCESS) rc = rcs[(size_t)k];
    S.in_flight = false;
    return rc;
}

int jda_node_checksums(jda_node *nd, int32_t n, const jda_output *surfaces, const int32_t *row_bytes, uint64_t *checksums)
{
    if (!nd) return JDA_ERROR_NO_DEVICE;
    if (n <= 0 || !surfaces || !row_bytes || !checksums) return JDA_INVALID_PARAMETER;
    const int32_t nk = (int32_t)nd->devs.size();
    std::vector<int32_t> rcs((size_t)nk, JDA_SUCCESS);
    for (int32_t k = 0; k < nk; k++) {
        int32_t first = 0, count = 0;
        jda_node_shard_of(nk, n, k, &first, &count);
        if (count == 0) continue;
        DevThread *d = nd->devs[(size_t)k];
        int32_t *rc = &rcs[(size_t)k];
        d->post([=]() { *rc = jda_checksum_surfaces(d->ctx, count, surfaces + first, row_bytes + first, checksums + first); });
    }
    for (DevThread *d : nd->devs) d->join_job();
    int rc = JDA_SUCCESS;
    for (int32_t k = 0; k < nk; k++) if (rcs[(size_t)k] != JDA_SUCCESS && rc == JDA_SUCCESS) rc = rcs[(size_t)k];
    return rc;
}

int jda_node_get_stats(const jda_node *nd, jda_pipeline_stats *out)
{
    if (!nd || !out) return JDA_INVALID_PARAMETER;
    memset(out, 0, sizeof(*out));
    for (const DevThread *d : nd->devs) {
        jda_pipeline_stats s;
        if (jda_pipeline_get_stats(d->pipe, &s) != JDA_SUCCESS) continue;
        out->images += s.images; out->device_images += s.device_images; out->host_path_images += s.host_path_images; out->failed_images += s.failed_images;
        out->source_pixels += s.source_pixels; out->compressed_bytes += s.compressed_bytes; out->h2d_bytes += s.h2d_bytes; out->launches += s.launches;
        if (s.spec_rounds_max > out->spec_rounds_max) out->spec_rounds_max = s.spec_rounds_max;
    }
    return JDA_SUCCESS;
}

} // extern "C"
