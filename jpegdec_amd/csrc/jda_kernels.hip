// jda_kernels.hip -- gfx950 (CDNA4) kernels of the decode path.  Written for wave64 only.
//
// jda_decode_tiles<MODE, FAST>: one 192-thread workgroup (3 wavefronts) decodes one tile = 192
// consecutive 8x8 blocks of one MCU row (32 MCUs of 4:2:0).  ~52 KB of LDS per workgroup, so three
// workgroups (9 wavefronts) share a CU and hide each other's LDS / HBM latency.
//   P0  Huffman LUTs (short halves), quantisers, zigzag -> LDS; the tile's slice of the filtered
//       scan -> LDS window, all with coalesced 16-byte loads (compressed bytes leave HBM once);
//   P1  thread = block: Huffman/RLE expand from the per-block index entry into the block's int16[64]
//       in LDS (coefficients never touch HBM); blocks are classified and their non-empty columns
//       appended to work lists with LDS atomics;
//   P2  thread = (block, non-empty column): dequant + IDCT column stage, in place;
//   P3  thread = (block, row): IDCT row stage + range limit, grouped by the reference's row variant;
//   P4  threads tile the tile's output rows: YCbCr -> RGB and 16-byte stores to consecutive addresses.
// No MFMA: the IDCT is shift/add integer work and the path is bound by the 4 B/pixel it writes.
#include <hip/hip_runtime.h>

#include "jda_device_core.h"
#include "jda_plan.h"

// optional phase trace (profiling aid, off unless jda_internal_set_trace() was called): per traced
// workgroup and wave, the shader clock at each phase boundary
__device__ unsigned long long *g_jda_trace = nullptr;
#define JDA_TRACE_STRIDE 64
#define JDA_TRACE(slot) do { if (trace && lane0) trace[(blockIdx.x / JDA_TRACE_STRIDE * 3 + (t >> 6)) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)

template <int MODE, bool FAST>
__global__ __launch_bounds__(JDA_WG_THREADS)
void jda_decode_tiles(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t t = threadIdx.x;
    unsigned long long *trace = (blockIdx.x % JDA_TRACE_STRIDE == 0) ? g_jda_trace : nullptr;
    const bool lane0 = (t & 63u) == 0;
    JDA_TRACE(0);

    // tile record and image descriptor are workgroup-uniform: keep them in SGPRs
    const jda_strip *tp = tiles + blockIdx.x;
    jda_strip S;
    S.image = __builtin_amdgcn_readfirstlane(tp->image);
    S.mcu_y = __builtin_amdgcn_readfirstlane(tp->mcu_y);
    S.mcu_x0 = __builtin_amdgcn_readfirstlane(tp->mcu_x0);
    S.count = __builtin_amdgcn_readfirstlane(tp->count);
    const jda_dev_desc &D = descs[S.image];
    jda_tile_ctx C = jda_tile_setup<MODE>(D, S);
    C.count = __builtin_amdgcn_readfirstlane(C.count);
    C.win_lo = __builtin_amdgcn_readfirstlane(C.win_lo);
    C.win_len = __builtin_amdgcn_readfirstlane(C.win_len);

    const jda_p1_inputs p1in = jda_p1_prefetch<MODE>(D, C, t);   // in flight while P0 stages LDS
    JDA_TRACE(1);
    jda_p0_stage<MODE>(D, C, t, lds, JDA_WIN_BYTES);
    JDA_TRACE(2);
    __syncthreads();
    JDA_TRACE(3);
    if (!(D.pad_[0] & 4)) jda_p1_entropy<MODE>(D, C, p1in, lds, JDA_WIN_BYTES);
    JDA_TRACE(4);
    __syncthreads();
    JDA_TRACE(5);
    if (D.scale_shift < 2 && !(D.pad_[0] & 4)) {
        if (!(D.pad_[0] & 2)) {
            jda_p2_columns<MODE, FAST>(D, t, lds);
            JDA_TRACE(6);
            __syncthreads();
            JDA_TRACE(7);
            jda_p3_rows<MODE>(D, t, lds);
            JDA_TRACE(8);
        }
        __syncthreads();
        JDA_TRACE(9);
    }
    if (!(D.pad_[0] & 1)) jda_p4_output<MODE>(D, S, C, t, lds);
    JDA_TRACE(10);
    if (trace) { __builtin_amdgcn_s_waitcnt(0); JDA_TRACE(11); }   // + time for this wave's stores to be acknowledged
}

template <int MODE, bool FAST>
static hipError_t launch(const jda_dev_desc *descs, const jda_strip *tiles, uint32_t n_tiles, hipStream_t stream)
{
    const int lds_bytes = jda_lds_layout<MODE>::TOTAL_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)jda_decode_tiles<MODE, FAST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((jda_decode_tiles<MODE, FAST>), dim3(n_tiles), dim3(JDA_WG_THREADS), lds_bytes, stream, descs, tiles);
    return hipGetLastError();
}

extern "C" hipError_t jda_internal_set_trace(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_jda_trace), &dev_buf, sizeof(dev_buf));
}

// Launch entry used by jda_runtime.cpp.
extern "C" hipError_t jda_launch_decode(int mode, int fast_mul, const jda_dev_desc *descs, const jda_strip *tiles,
                                        uint32_t n_tiles, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    switch (mode * 2 + (fast_mul ? 1 : 0)) {
    case JDA_MODE_GRAY * 2 + 0: return launch<JDA_MODE_GRAY, false>(descs, tiles, n_tiles, stream);
    case JDA_MODE_GRAY * 2 + 1: return launch<JDA_MODE_GRAY, true>(descs, tiles, n_tiles, stream);
    case JDA_MODE_444 * 2 + 0: return launch<JDA_MODE_444, false>(descs, tiles, n_tiles, stream);
    case JDA_MODE_444 * 2 + 1: return launch<JDA_MODE_444, true>(descs, tiles, n_tiles, stream);
    case JDA_MODE_420 * 2 + 0: return launch<JDA_MODE_420, false>(descs, tiles, n_tiles, stream);
    default: return launch<JDA_MODE_420, true>(descs, tiles, n_tiles, stream);
    }
}
