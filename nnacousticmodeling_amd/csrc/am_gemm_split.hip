// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: the optional split-f16 precision mode of the Linear layers
#include "am_device.h"

namespace am {

// ---------------------------------------------------------------------------------------------------
// Split-f16 Linear layer (optional precision mode AM_PRECISION_SPLIT_F16): the same NT GEMM on the f16 matrix pipe, with
// every f32 operand carried as two halves, x = hi + lo (hi = rn_f16(x), lo = rn_f16(x - hi)), and
//     x * w  ~  lo*wh + hi*wl + hi*wh          (three v_mfma_f32_32x32x16_f16 into ONE f32 accumulator; lo*wl ~ 2^-22 dropped)
// Products of halves are exact in f32 and the pipe accumulates in f32, so the result is f32-accurate: on the bench shape the
// error against an f64 reference is 2.1e-6 vs 3.7e-6 for the exact-f32 MFMA chain (tools/split_lab.hip), at 2.4x the speed
// (the f16 pipe is 16x the f32 one; three MFMAs per product; the kernel becomes bound by the LDS-DMA operand stream, whose
// bytes per flop are unchanged: 2 x 2 B instead of 4 B per element).
// Weights are pre-scaled by an exact power of two per layer so their lo halves stay normal f16 numbers; the epilogue
// multiplies the accumulator by the inverse.  Activations whose magnitude would not fit f16 raise an overflow flag and
// the host re-runs the pass on the exact f32 path.
// Same tiling as gemm_nt_f32_kernel: block 256 x 256, 8 waves 4 x 2, BK = 32 halves (64-B LDS rows, chunk ^ ((row>>2)&3)),
// 2 LDS stages (128 KiB), LDS-DMA issue interleaved between MFMA groups.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                        size_t n, float scale, int* __restrict__ overflow)
{
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i] * scale;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
        bad |= !(fabsf(v) <= SPLIT_F16_LIMIT);
    }
    if (bad && overflow) *overflow = 1;
}

hipError_t launch_split_f32(hipStream_t s, const float* x, void* hi, void* lo, size_t n, float scale, int* overflow)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (_Float16*)hi, (_Float16*)lo, n, scale, overflow);
    return hipGetLastError();
}

// OUT_F32 = false: hidden layer, writes act(acc*out_scale + bias) as hi/lo halves (Chi, Clo, leading dimension ldc halves)
// OUT_F32 = true : output layer, writes f32 with the EPI_LINEAR / EPI_SCALE / EPI_ACCUM modes of the exact kernel
template <bool OUT_F32>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_split_kernel(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, int lda,
    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo, int ldw, const float* __restrict__ bias,
    float* __restrict__ C, _Float16* __restrict__ Chi, _Float16* __restrict__ Clo, int ldc,
    int tiles_n, int nk, int act, int epi, float alpha, float out_scale, int* __restrict__ overflow, int row_p, int row_skip)
{
    constexpr int BM = GEMM_BM, BN = 256, BK = 32, NW = GEMM_WM * GEMM_WN;
    constexpr int MI = 2, NI = BN / GEMM_WN / 32;
    constexpr int ROWB = BK * 2;                   // 64-B LDS rows
    constexpr int RPI = 1024 / ROWB;               // rows per 1-KiB LDS-DMA piece
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + W_BYTES);
    constexpr int A_PER_WAVE = BM / RPI / NW, W_PER_WAVE = BN / RPI / NW;  // per operand half
    constexpr int LOADS = 2 * (A_PER_WAVE + W_PER_WAVE);
    static_assert(LOADS <= 2 * MI * NI, "not enough MFMA groups to interleave the LDS-DMA issue");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / GEMM_WN, wn = wave % GEMM_WN;

    size_t aoff[A_PER_WAVE], woff[W_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / 4;
        const int c = (lane % 4) ^ ((r >> 2) & 3);
        const int R = m0 + r;   // TDNN layers: windowed A rows, as in gemm_nt_f32_kernel
        aoff[j] = (row_p > 0 ? (size_t)R + (size_t)(R / row_p) * row_skip : (size_t)R) * lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / 4;
        const int c = (lane % 4) ^ ((r >> 2) & 3);
        woff[j] = (size_t)(n0 + r) * ldw + c * 8;
    }
    auto issue = [&](char* sbase, int k0, int l) {   // pieces: [Ahi][Alo][Whi][Wlo], each a run of 1-KiB wave-instructions
        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(Ahi + aoff[l] + k0), AM_LDS_PTR(sbase + (l * NW + wave) * 1024), 16, 0, 0);
        else if (l < 2 * A_PER_WAVE) { const int j = l - A_PER_WAVE; __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(Alo + aoff[j] + k0), AM_LDS_PTR(sbase + A_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
        else if (l < 2 * A_PER_WAVE + W_PER_WAVE) { const int j = l - 2 * A_PER_WAVE; __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(Whi + woff[j] + k0), AM_LDS_PTR(sbase + 2 * A_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
        else { const int j = l - 2 * A_PER_WAVE - W_PER_WAVE; __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(Wlo + woff[j] + k0), AM_LDS_PTR(sbase + 2 * A_BYTES + W_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
    const int li = lane & 31, lh = lane >> 5;
    int aidx[MI], asw[MI], widx[NI], wsw[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * 64 + mi * 32 + li; aidx[mi] = r * 4; asw[mi] = (r >> 2) & 3; }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * (BN / GEMM_WN) + ni * 32 + li; widx[ni] = r * 4; wsw[ni] = (r >> 2) & 3; }

#pragma unroll
    for (int l = 0; l < LOADS; ++l) issue(smem, 0, l);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int kn = kt + 1 < nk ? kt + 1 : nk - 1;   // the tail re-fetches the last stage into the idle buffer (never read)
        char* sbase = smem + (buf ^ 1) * STAGE_BYTES;
        const int k0 = kn * BK;
        const h8* ah4 = (const h8*)(smem + buf * STAGE_BYTES);
        const h8* al4 = (const h8*)(smem + buf * STAGE_BYTES + A_BYTES);
        const h8* wh4 = (const h8*)(smem + buf * STAGE_BYTES + 2 * A_BYTES);
        const h8* wl4 = (const h8*)(smem + buf * STAGE_BYTES + 2 * A_BYTES + W_BYTES);
#pragma unroll
        for (int g = 0; g < 2; ++g) {          // two 16-k MFMA groups per 32-k stage
            h8 ah[MI], al[MI], wh[NI], wl[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { const int idx = aidx[mi] + ((2 * g + lh) ^ asw[mi]); ah[mi] = ah4[idx]; al[mi] = al4[idx]; }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { const int idx = widx[ni] + ((2 * g + lh) ^ wsw[ni]); wh[ni] = wh4[idx]; wl[ni] = wl4[idx]; }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], wh[ni], acc[mi][ni], 0, 0, 0);   // small terms first
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], wl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], wh[ni], acc[mi][ni], 0, 0, 0);
                    const int slot = (g * MI + mi) * NI + ni;
                    if (slot < LOADS) issue(sbase, k0, slot);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int erow = m0 + wm * 64 + 4 * lh, ecol = n0 + wn * (BN / GEMM_WN) + li;
    bool bad = false;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = ecol + ni * 32;
            const float bv = bias[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = erow + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                float v = acc[mi][ni][reg] * out_scale + bv;
                const size_t o = (size_t)row * ldc + col;
                if (OUT_F32) {
                    if (epi == EPI_SCALE) v = v * alpha;
                    else if (epi == EPI_ACCUM) v = C[o] + v;
                    C[o] = v;
                } else {
                    v = am_activate(v, act);
                    const _Float16 h = (_Float16)v;
                    Chi[o] = h;
                    Clo[o] = (_Float16)(v - (float)h);
                    bad |= !(fabsf(v) <= SPLIT_F16_LIMIT);
                }
            }
        }
    if (!OUT_F32 && bad) *overflow = 1;
}

constexpr int GEMM_SPLIT_LDS = 2 * 2 * (GEMM_BM + 256) * 64;   // 128 KiB

hipError_t init_split_kernels()
{
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SPLIT_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)gemm_nt_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SPLIT_LDS);
}

hipError_t launch_gemm_split(hipStream_t s, const SplitGemmArgs& g)
{
    if (g.M_pad % GEMM_BM || g.N_pad % 256 || g.K_pad % 32 || g.M_pad <= 0 || g.K_pad <= 0) return hipErrorInvalidValue;
    const int tiles_n = g.N_pad / 256, grid = (g.M_pad / GEMM_BM) * tiles_n;
    if (g.C)
        hipLaunchKernelGGL(gemm_nt_split_kernel<true>, dim3(grid), dim3(GEMM_THREADS), GEMM_SPLIT_LDS, s, (const _Float16*)g.Ahi, (const _Float16*)g.Alo, g.lda,
                           (const _Float16*)g.Whi, (const _Float16*)g.Wlo, g.ldw, g.bias, g.C, (_Float16*)nullptr, (_Float16*)nullptr, g.ldc, tiles_n, g.K_pad / 32,
                           g.act, g.epi, g.alpha, g.out_scale, g.overflow, g.row_p, g.row_skip);
    else
        hipLaunchKernelGGL(gemm_nt_split_kernel<false>, dim3(grid), dim3(GEMM_THREADS), GEMM_SPLIT_LDS, s, (const _Float16*)g.Ahi, (const _Float16*)g.Alo, g.lda,
                           (const _Float16*)g.Whi, (const _Float16*)g.Wlo, g.ldw, g.bias, (float*)nullptr, (_Float16*)g.Chi, (_Float16*)g.Clo, g.ldc, tiles_n, g.K_pad / 32,
                           g.act, g.epi, g.alpha, g.out_scale, g.overflow, g.row_p, g.row_skip);
    return hipGetLastError();
}

}  // namespace am
