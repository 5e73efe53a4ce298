// jda_kernels.hip -- gfx950 (CDNA4) kernels of the decode path.  Written for wave64 only.
//
// jda_decode_strips<MODE, FAST>: one 256-thread workgroup = 4 independent wavefronts; each
// wavefront decodes one strip (<= 64 consecutive MCUs of one MCU row) of one image:
//   * the image's Huffman LUTs + zigzag table (10.6 KB) are copied once per workgroup into LDS
//     with coalesced 16-byte loads; the (wave-uniform) quantisers stay in SGPRs via scalar loads;
//   * the strip's slice of the filtered scan is staged into the wave's LDS window with coalesced
//     16-byte loads (the compressed bytes are read from HBM exactly once, in full cache lines);
//   * phase A: lane = MCU.  Huffman/RLE expand straight into a lane-private 8x8 int16 block in LDS,
//     then dequant + fixed-point IDCT in registers, 8-bit samples to the lane's LDS planes.
//     Coefficients never touch HBM.
//   * phase B: the 64 lanes tile the strip's output rows (4 pixels per lane, consecutive lanes ->
//     consecutive 16-byte groups) so every store instruction writes full, contiguous cache lines.
// No MFMA: the IDCT is shift/add integer work and the path is bound by the 4 B/pixel it writes.
#include <hip/hip_runtime.h>

#include "jda_device_core.h"
#include "jda_plan.h"

template <int MODE, bool FAST>
__global__ __launch_bounds__(64 * JDA_WAVES_PER_WG)
void jda_decode_strips(const jda_dev_desc *__restrict__ descs, const jda_strip *__restrict__ strips)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef jda_lds_layout<MODE> L;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;

    // strip record: wave-uniform, keep it in SGPRs
    const jda_strip *sp = strips + (size_t)blockIdx.x * JDA_WAVES_PER_WG + wave;
    jda_strip S;
    S.image = __builtin_amdgcn_readfirstlane(sp->image);
    S.mcu_y = __builtin_amdgcn_readfirstlane(sp->mcu_y);
    S.mcu_x0 = __builtin_amdgcn_readfirstlane(sp->mcu_x0);
    S.count = __builtin_amdgcn_readfirstlane(sp->count);
    // all strips of a workgroup belong to one image (the list is padded per image)
    const jda_dev_desc &D = descs[strips[(size_t)blockIdx.x * JDA_WAVES_PER_WG].image];

    uint8_t *tables = lds;
    {
        const uint4 *src = (const uint4 *)D.tables;
        uint4 *dst = (uint4 *)tables;
        for (uint32_t i = threadIdx.x; i < JDA_TABLE_BYTES / 16; i += 64 * JDA_WAVES_PER_WG) dst[i] = src[i];
    }
    uint8_t *wave_lds = lds + JDA_TABLE_BYTES + wave * L::WAVE_BYTES;
    jda_window W = jda_strip_window(D, S, JDA_WIN_BYTES);
    W.lo = __builtin_amdgcn_readfirstlane(W.lo);
    W.len = __builtin_amdgcn_readfirstlane(W.len);
    jda_window_fill(D.scan, W.lo, W.len, wave_lds + L::WIN_OFF, lane);
    __syncthreads();                       // tables (workgroup-wide) and this wave's window are in LDS

    jda_phase_a<MODE, FAST>(D, S, lane, tables, wave_lds, W);
    // phase B reads other lanes' planes of the SAME wave: LDS operations of one wave complete in
    // order, so a wave-scope fence (compiler ordering) is all that is needed -- no s_barrier.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!(D.pad_[0] & 1)) jda_phase_b<MODE>(D, S, lane, wave_lds);
}

template <int MODE> static size_t lds_bytes()
{
    return JDA_TABLE_BYTES + (size_t)JDA_WAVES_PER_WG * jda_lds_layout<MODE>::WAVE_BYTES;
}

template <int MODE, bool FAST>
static hipError_t launch(const jda_dev_desc *descs, const jda_strip *strips, uint32_t n_strips, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)jda_decode_strips<MODE, FAST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<MODE>());
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 grid(n_strips / JDA_WAVES_PER_WG), block(64 * JDA_WAVES_PER_WG);
    hipLaunchKernelGGL((jda_decode_strips<MODE, FAST>), grid, block, lds_bytes<MODE>(), stream, descs, strips);
    return hipGetLastError();
}

// Launch entry used by jda_runtime.cpp.  n_strips is a multiple of JDA_WAVES_PER_WG.
extern "C" hipError_t jda_launch_decode(int mode, int fast_mul, const jda_dev_desc *descs, const jda_strip *strips,
                                        uint32_t n_strips, hipStream_t stream)
{
    if (n_strips == 0) return hipSuccess;
    switch (mode * 2 + (fast_mul ? 1 : 0)) {
    case JDA_MODE_GRAY * 2 + 0: return launch<JDA_MODE_GRAY, false>(descs, strips, n_strips, stream);
    case JDA_MODE_GRAY * 2 + 1: return launch<JDA_MODE_GRAY, true>(descs, strips, n_strips, stream);
    case JDA_MODE_444 * 2 + 0: return launch<JDA_MODE_444, false>(descs, strips, n_strips, stream);
    case JDA_MODE_444 * 2 + 1: return launch<JDA_MODE_444, true>(descs, strips, n_strips, stream);
    case JDA_MODE_420 * 2 + 0: return launch<JDA_MODE_420, false>(descs, strips, n_strips, stream);
    default: return launch<JDA_MODE_420, true>(descs, strips, n_strips, stream);
    }
}
