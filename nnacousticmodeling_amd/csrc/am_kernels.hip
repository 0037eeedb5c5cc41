// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine.
//
// All buffers are laid out by the host layer so that the kernels need no bounds checks in their main loops:
// activation matrices have M padded to 128 rows and their leading dimension padded to the next layer's K_pad,
// weights are zero-padded to [N_pad, K_pad].  Wavefront = 64 lanes everywhere.
#include "am_internal.h"

namespace am {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AM_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float am_sigmoid(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }

__device__ __forceinline__ float am_activate(float v, int act) {
    if (act == 0) return fmaxf(v, 0.0f);
    if (act == 1) return am_sigmoid(v);
    return tanhf(v);
}

// ---------------------------------------------------------------------------------------------------
// Linear layer:  C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N])   -- "NT" GEMM, both operands K-contiguous.
//
// Exact-f32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles / SIMD, one k-pair per instruction, bitwise an fmaf chain).
// Block = 512 threads = 8 waves as 4(M) x 2(N); block tile 256 x BN (BN = 256 for hidden layers, 384 for the
// 1909-class output layer: 1920 = 5 x 384, no column padding waste); each wave owns 64 x BN/2 = 2 x NI MFMA tiles.
// Tile choice measured in tools/gemm_lab.hip on M=196608: 128x128 tiles 135.5 TF/s, 256x256 142.8, 256x384 142.1
// (the LDS-DMA stream costs MFMA issue slots in proportion to bytes per flop; a no-load ablation runs 149).
// Staging: global_load_lds_dwordx4 straight into LDS (no VGPR round trip), 2 stages.
// LDS image per operand: [rows][BK floats]; the 16-B chunk index is XOR-swizzled with the row so a ds_read_b128
// lane group (16 distinct rows, same chunk) touches 16 distinct 16-B slots of the 256-B bank row.
// LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address (guide rule 21).
// K permutation: lane (i = lane&31, h = lane>>5) reads 4 consecutive floats k = 8t + 4h + j (one ds_read_b128)
// and feeds them to MFMA steps j = 0..3; A and W use the same map, so each MFMA pairs equal k.
// Loop: [vmcnt(0)] [raw s_barrier] [issue stage kt+1, interleaved with] [MFMAs of stage kt].
// ---------------------------------------------------------------------------------------------------
constexpr int GEMM_WM = 4, GEMM_WN = 2, GEMM_THREADS = 64 * GEMM_WM * GEMM_WN;

// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// MODE: 0 relu, 1 sigmoid, 2 tanh, 3 linear, 4 scale by alpha, 5 accumulate into C.
template <int MODE, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[MI][NI], const float* __restrict__ bias, float* __restrict__ C,
                                              int ldc, int erow, int ecol, float alpha)
{
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = ecol + ni * 32;
            const float bv = bias[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = erow + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                float v = acc[mi][ni][reg] + bv;
                float* dst = C + (size_t)row * ldc + col;
                if (MODE == 0) v = fmaxf(v, 0.0f);
                else if (MODE == 1) v = am_sigmoid(v);
                else if (MODE == 2) v = tanhf(v);
                else if (MODE == 4) v = v * alpha;
                else if (MODE == 5) v = *dst + v;
                *dst = v;
            }
        }
    }
}

template <int BN, int BK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    float* __restrict__ C, int ldc, int tiles_n, int nk, int act, int epi, float alpha)
{
    constexpr int BM = GEMM_BM, NW = GEMM_WM * GEMM_WN;
    constexpr int MI = BM / GEMM_WM / 32, NI = BN / GEMM_WN / 32;
    constexpr int CH = BK / 4;                      // 16-B chunks per row
    constexpr int RPI = 64 / CH;                    // rows per 1-KiB wave-instruction
    constexpr int STAGE_BYTES = (BM + BN) * BK * 4;
    constexpr int A_PER_WAVE = BM / RPI / NW, W_PER_WAVE = BN / RPI / NW;
    constexpr int LOADS = A_PER_WAVE + W_PER_WAVE;  // LDS-DMA instructions per wave per stage
    constexpr int TS = BK / 8;                      // 8-k sub-steps per stage
    constexpr int SLOTS = TS * 4;                   // MFMA groups per stage
    constexpr int EVERY = SLOTS / LOADS > 0 ? SLOTS / LOADS : 1;
    constexpr int SW_SHIFT = (BK == 32) ? 1 : 2;    // 128-B rows: (row>>1)&7 ; 64-B rows: (row>>2)&3
    constexpr int SW_MASK = CH - 1;
    static_assert(LOADS <= SLOTS, "not enough MFMA groups to interleave the LDS-DMA issue");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run
    // of tiles, n fastest, so the blocks resident on one XCD share A panels and W in its L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / GEMM_WN, wn = wave % GEMM_WN;

    const float* ga[A_PER_WAVE];
    const float* gw[W_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / CH;
        const int c = (lane % CH) ^ ((r >> SW_SHIFT) & SW_MASK);
        ga[j] = A + (size_t)(m0 + r) * lda + c * 4;
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / CH;
        const int c = (lane % CH) ^ ((r >> SW_SHIFT) & SW_MASK);
        gw[j] = W + (size_t)(n0 + r) * ldw + c * 4;
    }
    auto issue = [&](char* sbase, int k0, int l) {   // l-th LDS-DMA piece of a stage (wave-uniform LDS address)
        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(ga[l] + k0), AM_LDS_PTR(sbase + (l * NW + wave) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(gw[l - A_PER_WAVE] + k0), AM_LDS_PTR(sbase + BM * BK * 4 + ((l - A_PER_WAVE) * NW + wave) * 1024), 16, 0, 0);
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
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * (BM / GEMM_WM) + mi * 32 + li; aidx[mi] = r * CH; asw[mi] = (r >> SW_SHIFT) & SW_MASK; }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * (BN / GEMM_WN) + ni * 32 + li; widx[ni] = r * CH; wsw[ni] = (r >> SW_SHIFT) & SW_MASK; }

#pragma unroll
    for (int l = 0; l < LOADS; ++l) issue(smem, 0, l);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage kt landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                      // ... and everyone's; everyone is done reading stage kt-1
        // prefetch stage kt+1 into the other buffer; the tail re-fetches the last stage into the idle buffer (never read)
        const int kn = kt + 1 < nk ? kt + 1 : nk - 1;
        char* sbase = smem + (buf ^ 1) * STAGE_BYTES;
        const int k0 = kn * BK;
        const f32x4* a4 = (const f32x4*)(smem + buf * STAGE_BYTES);
        const f32x4* w4 = (const f32x4*)(smem + buf * STAGE_BYTES + BM * BK * 4);
        f32x4 av[2][MI], bv[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[0][mi] = a4[aidx[mi] + (lh ^ asw[mi])];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[0][ni] = w4[widx[ni] + (lh ^ wsw[ni])];
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            if (t + 1 < TS) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) av[(t + 1) & 1][mi] = a4[aidx[mi] + ((2 * (t + 1) + lh) ^ asw[mi])];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bv[(t + 1) & 1][ni] = w4[widx[ni] + ((2 * (t + 1) + lh) ^ wsw[ni])];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][mi][j], bv[t & 1][ni][j], acc[mi][ni], 0, 0, 0);
                const int slot = t * 4 + j;
                if (slot % EVERY == 0 && slot / EVERY < LOADS) issue(sbase, k0, slot / EVERY);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail prefetch before the wave ends

    // epilogue: one uniform dispatch on the mode, then straight-line code
    const int erow = m0 + wm * (BM / GEMM_WM) + 4 * lh, ecol = n0 + wn * (BN / GEMM_WN) + li;
    if (epi == EPI_ACT) {
        if (act == 0) gemm_epilogue<0, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
        else if (act == 1) gemm_epilogue<1, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
        else gemm_epilogue<2, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    } else if (epi == EPI_LINEAR) gemm_epilogue<3, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    else if (epi == EPI_SCALE) gemm_epilogue<4, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    else gemm_epilogue<5, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
}

constexpr int GEMM_LDS_256 = 2 * (GEMM_BM + 256) * 32 * 4;  // 128 KiB
constexpr int GEMM_LDS_384 = 2 * (GEMM_BM + 384) * 16 * 4;  //  80 KiB

hipError_t init_kernels()
{
    // per device: allow the GEMM's large dynamic LDS
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_f32_kernel<256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_256);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)gemm_nt_f32_kernel<384, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_384);
}

int gemm_n_pad(int n)
{
    // 384-wide tiles when they waste fewer padded columns than 256-wide ones (1909 -> 1920 instead of 2048)
    const int p256 = (n + 255) / 256 * 256, p384 = (n + 383) / 384 * 384;
    return p384 < p256 ? p384 : p256;
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs& g)
{
    if (g.M_pad % GEMM_BM || g.K_pad % GEMM_BK || g.M_pad <= 0 || g.K_pad <= 0) return hipErrorInvalidValue;
    const int tiles_m = g.M_pad / GEMM_BM;
    if (g.N_pad % 256 == 0) {
        const int tiles_n = g.N_pad / 256;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<256, 32>), dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), GEMM_LDS_256, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 32, g.act, g.epi, g.alpha);
    } else if (g.N_pad % 384 == 0) {
        const int tiles_n = g.N_pad / 384;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<384, 16>), dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), GEMM_LDS_384, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 16, g.act, g.epi, g.alpha);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Fused splice + Kaldi feature transform + i-vector passthrough.  HBM-bound: every output element is written
// once (coalesced, 16 B per lane), every input row is read n_shifts times out of L1/L2.
// out[r, w*D + d] = (x[clamp(first + r + shift_w, 0, ext_rows-1), d] + add[w*D+d]) * scale[w*D+d]
// out[r, n_shifts*D + j] = ivec[row0 + r, j];  remaining columns up to ldo and rows [n, n_pad) are zero.
// VEC = 4 needs D % 4 == 0 and d_iv % 4 == 0 (then no float4 straddles a segment); VEC = 1 is the general form.
// ---------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void splice_ft_kernel(
    const float* __restrict__ x, long long ext_rows, long long first, const float* __restrict__ ivec, long long row0,
    long long n, long long n_pad, int D, int n_shifts, const int* __restrict__ shifts,
    const float* __restrict__ add, const float* __restrict__ scale, int d_iv, float* __restrict__ out, int ldo,
    int rows_per_block)
{
    const int Q = ldo / VEC;                 // vector columns per row
    const int Qt = Q < 256 ? Q : 256;        // threads form a [rows_per_pass][Qt] grid; surplus threads idle
    const int rows_per_pass = 256 / Qt;
    const int tid = threadIdx.x;
    const int rsub = tid / Qt;
    if (rsub >= rows_per_pass) return;
    const long long rblock = (long long)blockIdx.x * rows_per_block;
    const int spl = n_shifts * D;

    for (int q = tid - rsub * Qt; q < Q; q += Qt) {
        const int col = q * VEC;
        int w = 0, d = 0, kind;  // 0 = spliced feature, 1 = i-vector, 2 = zero pad
        int shift = 0;
        if (col < spl) { kind = 0; w = col / D; d = col - w * D; shift = shifts[w]; }
        else if (col < spl + d_iv) { kind = 1; d = col - spl; }
        else kind = 2;
        float av[VEC], sv[VEC];
        const bool has_ft = (add != nullptr);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { av[e] = (kind == 0 && has_ft) ? add[col + e] : 0.0f; sv[e] = (kind == 0 && has_ft) ? scale[col + e] : 1.0f; }

        for (int rr = rsub; rr < rows_per_block; rr += rows_per_pass) {
            const long long r = rblock + rr;
            if (r >= n_pad) break;
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = 0.0f;
            if (r < n) {
                if (kind == 0) {
                    long long src = first + r + shift;
                    src = src < 0 ? 0 : (src > ext_rows - 1 ? ext_rows - 1 : src);
                    const float* p = x + src * D + d;
                    if (VEC == 4) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                    else v[0] = p[0];
                    if (has_ft) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) { const float s1 = v[e] + av[e]; v[e] = s1 * sv[e]; }
                    }
                } else if (kind == 1) {
                    const float* p = ivec + (row0 + r) * d_iv + d;
                    if (VEC == 4) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                    else v[0] = p[0];
                }
            }
            float* o = out + r * ldo + col;
            if (VEC == 4) { f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *(f32x4*)o = t; }
            else o[0] = v[0];
        }
    }
}

hipError_t launch_splice_ft(hipStream_t s, const SpliceArgs& a)
{
    if (a.n_pad <= 0) return hipSuccess;
    const bool vec = (a.D % 4 == 0) && (a.d_iv % 4 == 0) && (a.ldo % 4 == 0);
    const int rows_per_block = 32;
    const long long blocks = (a.n_pad + rows_per_block - 1) / rows_per_block;
    if (vec)
        hipLaunchKernelGGL(splice_ft_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first,
                           a.ivec, (long long)a.row0, (long long)a.n, (long long)a.n_pad, a.D, a.n_shifts, a.shifts, a.add, a.scale, a.d_iv, a.out, a.ldo, rows_per_block);
    else
        hipLaunchKernelGGL(splice_ft_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first,
                           a.ivec, (long long)a.row0, (long long)a.n, (long long)a.n_pad, a.D, a.n_shifts, a.shifts, a.add, a.scale, a.d_iv, a.out, a.ldo, rows_per_block);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Finish: one wavefront per frame; lane l holds columns l, l+64, ... in registers (256-B coalesced dword
// accesses).  v -> v/divisor -> RPL4 (log_softmax, x + x*W + b, log-add-exp with floor lb) -> v - prior ->
// v - logsum(v), logsum = mx + log(sum exp(v - mx)) with NaN -> -1e20 (reference kw_utils.py:38-43).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NIT>
__device__ __forceinline__ float row_logsum(const float (&v)[NIT], int lane, int C)
{
    // np.max propagates NaN; fmaxf does not, but a NaN element makes the exp-sum NaN, which is all that matters.
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (it * 64 + lane < C) m = fmaxf(m, v[it]);
    m = wave_max(m);
    float sacc = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (it * 64 + lane < C) sacc += expf(v[it] - m);
    sacc = wave_sum(sacc);
    float lse = m + logf(sacc);
    if (lse != lse) lse = -1e20f;
    return lse;
}

template <int NIT>
__global__ __launch_bounds__(256) void finish_kernel(
    const float* __restrict__ in, int ldin, float* __restrict__ out, long long ldout, const int* __restrict__ row_map,
    long long n, int C, float divisor, const float* __restrict__ rpl_w, const float* __restrict__ rpl_b,
    const float* __restrict__ rpl_lb, const float* __restrict__ prior, int normalise, int accumulate)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    long long orow = row;
    if (row_map) { orow = row_map[row]; if (orow < 0) return; }

    float v[NIT];
    const float* src = in + row * ldin;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; v[it] = c < C ? src[c] : 0.0f; }

    if (divisor > 0.0f) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = v[it] / divisor;
    }
    if (rpl_w) {
        const float lse = row_logsum<NIT>(v, lane, C);  // F.log_softmax has no NaN patch-up; identical for finite rows
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < C) {
                const float x = v[it] - lse;
                const float xw = x * rpl_w[c];
                const float h = (x + xw) + rpl_b[c];
                const float lb = rpl_lb[c];
                const float mx = fmaxf(h, lb), mn = fminf(h, lb);
                v[it] = mx + logf(1.0f + expf(mn - mx));
            }
        }
    }
    if (prior) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; if (c < C) v[it] = v[it] - prior[c]; }
    }
    if (normalise) {
        const float lse = row_logsum<NIT>(v, lane, C);
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = v[it] - lse;
    }
    float* dst = out + orow * ldout;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < C) dst[c] = accumulate ? dst[c] + v[it] : v[it];
    }
}

hipError_t launch_finish(hipStream_t s, const FinishArgs& a)
{
    if (a.n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((a.n + 3) / 4);
    const int nit = (a.C + 63) / 64;
#define AM_FINISH(N) hipLaunchKernelGGL(finish_kernel<N>, dim3(blocks), dim3(256), 0, s, a.in, a.ldin, a.out, (long long)a.ldout, a.row_map, \
        (long long)a.n, a.C, a.divisor, a.rpl_w, a.rpl_b, a.rpl_lb, a.prior, a.normalise, a.accumulate)
    if (nit <= 1) AM_FINISH(1);
    else if (nit <= 2) AM_FINISH(2);
    else if (nit <= 4) AM_FINISH(4);
    else if (nit <= 8) AM_FINISH(8);
    else if (nit <= 16) AM_FINISH(16);
    else if (nit <= 30) AM_FINISH(30);
    else if (nit <= 32) AM_FINISH(32);
    else return hipErrorInvalidValue;
#undef AM_FINISH
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent path, stage 1: gather file-order frames into the time-major packed order + per-dim transform.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_ft_kernel(
    const float* __restrict__ x, const float* __restrict__ ivec, const int* __restrict__ src, long long n, long long n_pad,
    int D, const float* __restrict__ add, const float* __restrict__ scale, int d_iv, float* __restrict__ out, int ldo)
{
    const long long total = n_pad * ldo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / ldo;
        const int c = (int)(idx - r * ldo);
        float v = 0.0f;
        if (r < n) {
            const long long f = src[r];
            if (c < D) {
                v = x[f * D + c];
                if (add) { const float s1 = v + add[c]; v = s1 * scale[c]; }
            } else if (c < D + d_iv) {
                v = ivec[f * d_iv + (c - D)];
            }
        }
        out[idx] = v;
    }
}

hipError_t launch_gather_ft(hipStream_t s, const GatherArgs& a)
{
    if (a.n_pad <= 0) return hipSuccess;
    long long total = a.n_pad * a.ldo;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_ft_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.x, a.ivec, a.src, (long long)a.n, (long long)a.n_pad,
                       a.D, a.add, a.scale, a.d_iv, a.out, a.ldo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent path, stage 2: one LSTM time step for the active utterances (first n_active slots).
//   g = G[row] + Hprev[row] * Wlat^T ;  c' = tanh(a) sig(i) + sig(f) c ;  h' = sig(o) tanh(c')
// Gate-tiled layout (host permutes weights): columns [64*j + 16*k + u'] hold gate k (a,i,f,o) of unit 16*j+u'.
// Block = (16 utterances) x (16 units) x 4 gates; wave k computes gate k with v_mfma_f32_16x16x4_f32, operands
// straight from global/L2 (each weight element is used by exactly one wave of the block, no LDS staging),
// 4 consecutive k per 16-B load with the same K permutation trick as the GEMM.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_kernel(
    const float* __restrict__ G, const float* __restrict__ Hprev, const float* __restrict__ Wlat,
    float* __restrict__ Cst, float* __restrict__ Hout, int n_active, int U_pad, int first)
{
    __shared__ float sg[4][16][17];
    const int ut = blockIdx.x, rt = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int gate = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int ldg = 4 * U_pad;

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!first) {
        const float* wrow = Wlat + (size_t)(ut * 64 + gate * 16 + l15) * U_pad + 4 * lq;
        const float* arow = Hprev + (size_t)(rt * 16 + l15) * U_pad + 4 * lq;
        const int steps = U_pad / 16;
#pragma unroll 4
        for (int s = 0; s < steps; ++s) {
            const f32x4 a = *(const f32x4*)(arow + 16 * s);
            const f32x4 w = *(const f32x4*)(wrow + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w[j], acc, 0, 0, 0);
        }
    }
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = lq * 4 + reg;
        sg[gate][row][l15] = acc[reg] + G[(size_t)(rt * 16 + row) * ldg + ut * 64 + gate * 16 + l15];
    }
    __syncthreads();
    const int row = tid >> 4, u = tid & 15;
    const int slot = rt * 16 + row;
    if (slot < n_active) {
        const float a = tanhf(sg[0][row][u]);
        const float i = am_sigmoid(sg[1][row][u]);
        const float f = am_sigmoid(sg[2][row][u]);
        const float o = am_sigmoid(sg[3][row][u]);
        const size_t idx = (size_t)slot * U_pad + ut * 16 + u;
        const float cprev = first ? 0.0f : Cst[idx];
        const float ai = a * i;
        const float fc = f * cprev;
        const float c = ai + fc;
        Cst[idx] = c;
        Hout[idx] = o * tanhf(c);
    }
}

hipError_t launch_lstm_step(hipStream_t s, const LstmStepArgs& a)
{
    if (a.n_active <= 0) return hipSuccess;
    dim3 grid(a.U_pad / 16, (a.n_active + 15) / 16);
    hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, s, a.G, a.Hprev, a.Wlat, a.Cst, a.Hout, a.n_active, a.U_pad, a.first);
    return hipGetLastError();
}

}  // namespace am
