// What the memory system of this GPU sustains for the decode kernel's kind of traffic (tools/, not product): a pure fill with 16-byte
// stores (the colour stage's stores), a pure read, a copy; and the fill issued the way the decode kernel issues it -- a persistent
// grid of 256 x 1024 threads, every wavefront writing 160 x 16 pixel tiles of an RGB8888 surface row by row (640-byte runs).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling tools/lab/hbm_ceiling.hip && ./hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill16(uint4 *p, size_t n16)
{
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill16_nt(uint4 *p, size_t n16)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = { threadIdx.x, blockIdx.x, 3, 4 };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, (u4 *)&p[i]);
}
__global__ void read16(const uint4 *p, size_t n16, uint32_t *sink)
{
    uint32_t a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; a += v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) *sink = a;
}
__global__ void copy16(const uint4 *s, uint4 *d, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// the decode kernel's store pattern: images of W x H RGB8888, tiles of 160 x 16 pixels, a wavefront per tile, lane = 4 x 2 pixel item,
// five passes of two 16-byte stores a pitch apart (jda_p4_420_full10)
__global__ void tiles(uint8_t *out, int W, int H, int n_img, int tiles_total)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const int tpr = W / 160, tpi = tpr * (H / 16);
    const int per_wg = (tiles_total + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per_wg, t1 = min(t0 + per_wg, tiles_total);
    const uint4 v = make_uint4(lane, wave, 3, 4);
    const size_t pitch = (size_t)W * 4;
    for (int t = t0 + wave; t < t1; t += waves) {
        const int img = t / tpi, r = t - img * tpi, ty = r / tpr, tx = r - ty * tpr;
        uint8_t *tile = out + (size_t)img * pitch * H + (size_t)ty * 16 * pitch + (size_t)tx * 640;
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int i = lane + 64 * it, rp = i / 40, g = i - rp * 40;
            uint8_t *d = tile + (size_t)rp * 2 * pitch + g * 16;
            *(uint4 *)d = v;
            *(uint4 *)(d + pitch) = v;
        }
    }
}
// the same stores with arithmetic between them: every wavefront does `work` dependent multiply-adds per tile (its four wavefronts a SIMD keep
// the VALU port busy as the decode kernel's do) and either waits for a tile's write acknowledgements before it goes on (WAIT: what
// sharing vmcnt with its prefetch loads forces on the decode kernel once a tile) or never waits for them
template <bool WAIT>
__global__ void tiles_busy(uint8_t *out, int W, int H, int n_img, int tiles_total, int work, uint32_t *sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const int tpr = W / 160, tpi = tpr * (H / 16);
    const int per_wg = (tiles_total + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per_wg, t1 = min(t0 + per_wg, tiles_total);
    const size_t pitch = (size_t)W * 4;
    uint32_t acc = lane;
    for (int t = t0 + wave; t < t1; t += waves) {
        for (int k = 0; k < work; k++) acc = acc * 1664525u + 1013904223u;          // (a quarter-rate multiply and an add a step)
        const uint4 v = make_uint4(acc, wave, 3, 4);
        const int img = t / tpi, r = t - img * tpi, ty = r / tpr, tx = r - ty * tpr;
        uint8_t *tile = out + (size_t)img * pitch * H + (size_t)ty * 16 * pitch + (size_t)tx * 640;
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int i = lane + 64 * it, rp = i / 40, g = i - rp * 40;
            uint8_t *d = tile + (size_t)rp * 2 * pitch + g * 16;
            *(uint4 *)d = v;
            *(uint4 *)(d + pitch) = v;
        }
        if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main()
{
    const int W = 4096, H = 4096, N = 64;
    const size_t bytes = (size_t)W * H * 4 * N, n16 = bytes / 16;
    uint8_t *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto run = [&](const char *name, auto launch, double traffic) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.4f ms  %.2f TB/s\n", name, ms / reps, traffic / (ms / reps * 1e-3) / 1e12);
    };
    for (int grid : {256, 1024, 4096, 16384})
        for (int block : {256, 1024}) {
            char nm[96];
            snprintf(nm, sizeof nm, "fill 16 B stores  grid %d x %d", grid, block);
            run(nm, [&]() { hipLaunchKernelGGL(fill16, dim3(grid), dim3(block), 0, 0, (uint4 *)a, n16); }, (double)bytes);
        }
    run("fill 16 B nontemporal  grid 4096 x 256", [&]() { hipLaunchKernelGGL(fill16_nt, dim3(4096), dim3(256), 0, 0, (uint4 *)a, n16); }, (double)bytes);
    run("read 16 B loads  grid 4096 x 256", [&]() { hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, n16, sink); }, (double)bytes);
    run("copy  grid 4096 x 256 (read + write)", [&]() { hipLaunchKernelGGL(copy16, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, n16); }, 2.0 * bytes);
    const int tiles_total = N * (W / 160) * (H / 16);      // (25 whole tiles a row: the last 96 columns are left out)
    run("decode-kernel store pattern 256 x 1024", [&]() { hipLaunchKernelGGL(tiles, dim3(256), dim3(1024), 0, 0, a, W, H, N, tiles_total); }, (double)tiles_total * 160 * 16 * 4);
    run("decode-kernel store pattern 512 x 512", [&]() { hipLaunchKernelGGL(tiles, dim3(512), dim3(512), 0, 0, a, W, H, N, tiles_total); }, (double)tiles_total * 160 * 16 * 4);
    for (int work : {0, 300, 500, 600, 700}) {
        char nm[96];
        snprintf(nm, sizeof nm, "stores + %d steps a tile, acks waited for", work);
        run(nm, [&]() { hipLaunchKernelGGL(tiles_busy<true>, dim3(256), dim3(1024), 0, 0, a, W, H, N, tiles_total, work, sink); }, (double)tiles_total * 160 * 16 * 4);
        snprintf(nm, sizeof nm, "stores + %d steps a tile, never waited for", work);
        run(nm, [&]() { hipLaunchKernelGGL(tiles_busy<false>, dim3(256), dim3(1024), 0, 0, a, W, H, N, tiles_total, work, sink); }, (double)tiles_total * 160 * 16 * 4);
    }
    CK(hipMemsetAsync(a, 0, bytes, 0));
    hipEventRecord(e0, 0); for (int i = 0; i < 5; i++) hipMemsetAsync(a, 0, bytes, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.4f ms  %.2f TB/s\n", "hipMemsetAsync", ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    return 0;
}
