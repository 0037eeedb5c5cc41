// Feasibility lab for an fp32-accurate GEMM on the f16 MFMA pipe ("3 x f16" split: x = hi + lo, x*w ~ hi*wh + hi*wl + lo*wh,
// f32 accumulate).  Measures speed and error against an f64 reference next to the exact f32-MFMA numbers.
// NOT part of the product.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/split_lab.hip -o tools/split_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// f32 [rows, ld] -> hi/lo halves (same shape).  scale = exact power of two applied first.
__global__ void split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo, size_t n, float scale)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i] * scale;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
}

// C[M,N] (f32) = relu((Ahi+Alo)(Whi+Wlo)^T * out_scale + bias), dropping lo*lo.  Block 256 x BN, 8 waves 4x2, BK = 32 halves.
template <int BN>
__global__ __launch_bounds__(512, 2) void gemm_split_kernel(const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, int lda,
                                                            const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ C, int ldc, int tiles_n, int nk, float out_scale)
{
    constexpr int BM = 256, BK = 32, NW = 8, WN = 2;
    constexpr int MI = 2, NI = BN / WN / 32;
    constexpr int ROWB = BK * 2;                   // 64-B rows
    constexpr int RPI = 1024 / ROWB;               // 16 rows per 1-KiB LDS-DMA piece
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = 2 * (A_BYTES + W_BYTES);
    constexpr int A_PER_WAVE = BM / RPI / NW, W_PER_WAVE = BN / RPI / NW;  // per operand half
    constexpr int LOADS = 2 * (A_PER_WAVE + W_PER_WAVE);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // per-lane source pointers: 64-B rows = 4 chunks of 16 B, chunk index swizzled with (row >> 2) & 3
    size_t aoff[A_PER_WAVE], woff[W_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / 4;
        const int c = (lane % 4) ^ ((r >> 2) & 3);
        aoff[j] = (size_t)(m0 + r) * lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / 4;
        const int c = (lane % 4) ^ ((r >> 2) & 3);
        woff[j] = (size_t)(n0 + r) * ldw + c * 8;
    }
    auto issue = [&](char* sbase, int k0, int l) {
        // pieces: [Ahi x A_PER_WAVE][Alo x A_PER_WAVE][Whi x W_PER_WAVE][Wlo x W_PER_WAVE]
        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(GP(Ahi + aoff[l] + k0), LP(sbase + (l * NW + wave) * 1024), 16, 0, 0);
        else if (l < 2 * A_PER_WAVE) { const int j = l - A_PER_WAVE; __builtin_amdgcn_global_load_lds(GP(Alo + aoff[j] + k0), LP(sbase + A_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
        else if (l < 2 * A_PER_WAVE + W_PER_WAVE) { const int j = l - 2 * A_PER_WAVE; __builtin_amdgcn_global_load_lds(GP(Whi + woff[j] + k0), LP(sbase + 2 * A_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
        else { const int j = l - 2 * A_PER_WAVE - W_PER_WAVE; __builtin_amdgcn_global_load_lds(GP(Wlo + woff[j] + k0), LP(sbase + 2 * A_BYTES + W_BYTES + (j * NW + wave) * 1024), 16, 0, 0); }
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
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * (BN / WN) + ni * 32 + li; widx[ni] = r * 4; wsw[ni] = (r >> 2) & 3; }

#pragma unroll
    for (int l = 0; l < LOADS; ++l) issue(smem, 0, l);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int kn = kt + 1 < nk ? kt + 1 : nk - 1;
        char* sbase = smem + (buf ^ 1) * STAGE_BYTES;
        const int k0 = kn * BK;
        const h8* ah4 = (const h8*)(smem + buf * STAGE_BYTES);
        const h8* al4 = (const h8*)(smem + buf * STAGE_BYTES + A_BYTES);
        const h8* wh4 = (const h8*)(smem + buf * STAGE_BYTES + 2 * A_BYTES);
        const h8* wl4 = (const h8*)(smem + buf * STAGE_BYTES + 2 * A_BYTES + W_BYTES);
        int issued = 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {          // two 16-k groups per stage
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
                    if (issued < LOADS) { issue(sbase, k0, issued); ++issued; }
                }
        }
#pragma unroll
        for (int l = 0; l < LOADS; ++l) if (l >= 2 * MI * NI) issue(sbase, k0, l);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int erow = m0 + wm * 64 + 4 * lh, ecol = n0 + wn * (BN / WN) + li;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = ecol + ni * 32;
            const float bv = bias[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = erow + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                C[(size_t)row * ldc + col] = fmaxf(acc[mi][ni][reg] * out_scale + bv, 0.0f);
            }
        }
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 196608, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    printf("M=%d N=%d K=%d\n", M, N, K);
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    srand(1);
    for (auto& v : hA) v = std::max(0.0f, (rand() / (float)RAND_MAX) * 4 - 1.5f);          // ReLU-like activations
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *dA, *dW, *db, *dC;
    _Float16 *dAh, *dAl, *dWh, *dWl;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dAh, hA.size() * 2)); CK(hipMalloc(&dAl, hA.size() * 2)); CK(hipMalloc(&dWh, hW.size() * 2)); CK(hipMalloc(&dWl, hW.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    float wmax = 0; for (float v : hW) wmax = std::max(wmax, std::fabs(v));
    const float wscale = std::exp2(std::floor(std::log2(1024.0f / wmax)));                 // exact power of two: max |W'| in [512, 1024]
    hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, s, dA, dAh, dAl, hA.size(), 1.0f);
    hipLaunchKernelGGL(split_kernel, dim3(4096), dim3(256), 0, s, dW, dWh, dWl, hW.size(), wscale);
    auto kern = gemm_split_kernel<256>;
    constexpr int lds = 2 * 2 * (256 + 256) * 64;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    auto run = [&]() { hipLaunchKernelGGL(kern, dim3((M / 256) * (N / 256)), dim3(512), lds, s, dAh, dAl, K, dWh, dWl, K, db, dC, N, N / 256, K / 32, 1.0f / wscale); };
    run();
    CK(hipStreamSynchronize(s)); CK(hipGetLastError());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int> sr = {0, 1, 127, 128, 255, 256, 4097, M / 2 + 3, M - 1};
    double maxerr = 0, maxerr32 = 0;
    for (int r : sr) for (int c = 0; c < N; c += 7) {
        double ref = hb[c]; float chain = 0.0f;
        for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k]; chain = fmaf(hA[(size_t)r * K + k], hW[(size_t)c * K + k], chain); }
        ref = std::max(ref, 0.0);
        maxerr = std::max(maxerr, std::fabs(ref - hC[(size_t)r * N + c]));
        maxerr32 = std::max(maxerr32, std::fabs(ref - std::max(0.0f, chain + hb[c])));
    }
    printf("split-f16 GEMM: max |err| vs f64 on samples %.3g   (plain f32 fmaf chain on the same samples: %.3g)\n", maxerr, maxerr32);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 5; ++i) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("split-f16 256x256x32: %.3f ms  = %.1f TF/s of f32-equivalent work (%.1f TF/s of f16 MFMA)\n", ms, 2.0 * M * N * K / ms / 1e9, 6.0 * M * N * K / ms / 1e9);
    }
    return 0;
}
