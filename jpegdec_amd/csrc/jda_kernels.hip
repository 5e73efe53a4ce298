// jda_kernels.hip -- gfx950 (CDNA4) kernels of the decode path.  Written for wave64 only.
//
// jda_decode_tiles<MODE, FAST>: a 256-thread workgroup = 4 independent wavefronts that share only
// the image's tables in LDS; each wavefront decodes one tile = up to 64 consecutive 8x8 blocks of
// one MCU row (10 MCUs of 4:2:0).  10.8 KB of LDS per wave + 6.7 KB of tables per workgroup ->
// three workgroups (12 wavefronts) per CU.  After the tables are staged there is no workgroup
// barrier: phases are separated by wave-local fences, so wavefronts drift freely.
//   P0  Huffman LUTs (short halves), quantisers, zigzag -> LDS (per workgroup); the tile's slice of
//       the filtered scan -> the wave's LDS window; coalesced 16-byte loads (compressed bytes leave
//       HBM once);
//   P1  lane = block: Huffman/RLE expand from the per-block index entry into the block's int16[64]
//       in LDS (coefficients never touch HBM); blocks are classified and their non-empty columns
//       appended to work lists with LDS atomics;
//   P2  lane = (block, non-empty column): dequant + IDCT column stage, in place;
//   P3  lane = (block, row): IDCT row stage + range limit, grouped by the reference's row variant;
//   P4  lanes tile the tile's output rows: YCbCr -> RGB and 16-byte stores to consecutive addresses.
// No MFMA: the IDCT is shift/add integer work and the path is bound by the 4 B/pixel it writes.
#include <hip/hip_runtime.h>

#include "jda_device_core.h"
#include "jda_plan.h"

// optional phase trace (profiling aid, off unless jda_internal_set_trace() was called): per traced
// workgroup and wave, the shader clock at each phase boundary
__device__ unsigned long long *g_jda_trace = nullptr;
#define JDA_TRACE_STRIDE 64
#define JDA_TRACE(slot) do { if (trace && lane == 0) trace[(blockIdx.x / JDA_TRACE_STRIDE * JDA_WAVES_PER_WG + wave) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)

// wave-local phase boundary: LDS operations of one wavefront complete in order, so ordering the
// compiler is all that is needed -- no s_barrier
#define JDA_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int MODE, bool FAST>
__global__ __launch_bounds__(64 * JDA_WAVES_PER_WG)
void jda_decode_tiles(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ tiles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef jda_lds_layout<MODE> L;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long *trace = (blockIdx.x % JDA_TRACE_STRIDE == 0) ? g_jda_trace : nullptr;
    JDA_TRACE(0);

    // tile record and image descriptor are wave-uniform: keep them in SGPRs
    const jda_strip *tp = tiles + (size_t)blockIdx.x * JDA_WAVES_PER_WG + wave;
    jda_strip S;
    S.image = __builtin_amdgcn_readfirstlane(tp->image);
    S.mcu_y = __builtin_amdgcn_readfirstlane(tp->mcu_y);
    S.mcu_x0 = __builtin_amdgcn_readfirstlane(tp->mcu_x0);
    S.count = __builtin_amdgcn_readfirstlane(tp->count);
    const jda_dev_desc &D = descs[S.image];           // the four tiles of a workgroup belong to one image
    jda_tile_ctx C = jda_tile_setup<MODE>(D, S);
    C.count = __builtin_amdgcn_readfirstlane(C.count);
    C.win_lo = __builtin_amdgcn_readfirstlane(C.win_lo);
    C.win_len = __builtin_amdgcn_readfirstlane(C.win_len);

    uint8_t *tab = lds;
    uint8_t *wl = lds + JDA_LT_BYTES + wave * L::WAVE_BYTES;
    const jda_p1_inputs p1in = jda_p1_prefetch<MODE>(D, C, lane);   // in flight while LDS is staged
    JDA_TRACE(1);
    jda_p0_tables(D, threadIdx.x, 64 * JDA_WAVES_PER_WG, tab);
    jda_p0_stage<MODE>(D, C, lane, wl, JDA_WIN_BYTES);
    JDA_TRACE(2);
    __syncthreads();                                  // the only workgroup barrier: tables are in LDS
    JDA_TRACE(3);
    if (!(D.pad_[0] & 4)) jda_p1_entropy<MODE>(D, C, p1in, tab, wl, JDA_WIN_BYTES);
    JDA_WAVE_SYNC();
    JDA_TRACE(4);
    if (D.scale_shift < 2 && !(D.pad_[0] & 6)) {
        jda_p2_columns<MODE, FAST>(D, lane, tab, wl);
        JDA_WAVE_SYNC();
        JDA_TRACE(5);
        jda_p3_rows<MODE>(D, lane, tab, wl);
        JDA_WAVE_SYNC();
        JDA_TRACE(6);
    }
    if (!(D.pad_[0] & 1)) jda_p4_output<MODE>(D, S, C, lane, wl);
    JDA_TRACE(7);
}

template <int MODE, bool FAST>
static hipError_t launch(const jda_dev_desc *descs, const jda_strip *tiles, uint32_t n_tiles, hipStream_t stream)
{
    const int lds_bytes = JDA_LT_BYTES + JDA_WAVES_PER_WG * jda_lds_layout<MODE>::WAVE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)jda_decode_tiles<MODE, FAST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((jda_decode_tiles<MODE, FAST>), dim3(n_tiles / JDA_WAVES_PER_WG), dim3(64 * JDA_WAVES_PER_WG),
                       lds_bytes, stream, descs, tiles);
    return hipGetLastError();
}

extern "C" hipError_t jda_internal_set_trace(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_jda_trace), &dev_buf, sizeof(dev_buf));
}

// Launch entry used by jda_runtime.cpp.  n_tiles is a multiple of JDA_WAVES_PER_WG (padded per image).
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
