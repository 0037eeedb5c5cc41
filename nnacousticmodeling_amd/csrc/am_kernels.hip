// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine.
//
// All buffers are laid out by the host layer so that the kernels need no bounds checks in their main loops:
// activation matrices have M padded to 256 rows (GEMM_BM) and their leading dimension padded to the next layer's K_pad,
// weights are zero-padded to [N_pad, K_pad].  Wavefront = 64 lanes everywhere.
#include "am_internal.h"
#include <algorithm>
#include <cstdlib>

namespace am {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LSTM_P_TIMEOUT = 200000000;   // ticks of s_memrealtime (100 MHz): 2 s, the bound of every inter-workgroup wait
__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define AM_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr float SPLIT_F16_LIMIT = 60000.0f;   // |value| above this cannot be carried as an f16 hi half (split-f16 mode)

__device__ __forceinline__ float am_sigmoid(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }

__device__ __forceinline__ float am_activate(float v, int act) {
    if (act == 0) return fmaxf(v, 0.0f);
    if (act == 1) return am_sigmoid(v);
    return tanhf(v);
}

// Gate math of the recurrent cells.  libm's tanhf costs ~60 instructions; an LSTM step evaluates five of them per unit and
// row, which was 2 of the 23 us of a 192-row step.  tanh(x) = 1 - 2 / (exp(2|x|) + 1) on the hardware exp2 / rcp for
// |x| >= 0.25 and an odd polynomial below: max abs error 1.1e-7 over [-12, 12] (numpy's float32 tanh: 6e-8), ~1/6 of the
// instructions.  sigmoid = 1 / (1 + exp(-x)) on the same two instructions (abs error 1.2e-7; Chainer's CPU form
// tanh(x/2)/2 + 1/2 is the same function).  Every recurrent kernel uses these two, so a frame's
// posteriors do not depend on the kernel form that computed them.
__device__ __forceinline__ float am_tanh_rec(float x)
{
    const float ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);   // exp(2|x|)
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    const float t = fmaf(-2.0f, r, 1.0f);
    const float x2 = x * x;
    float p = fmaf(-17.0f / 315.0f, x2, 2.0f / 15.0f);
    p = fmaf(p, x2, -1.0f / 3.0f);
    p = p * x2;
    const float poly = fmaf(ax, p, ax);
    return copysignf(ax < 0.25f ? poly : t, x);
}
__device__ __forceinline__ float am_sigmoid_rec(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }   // 1 / (1 + exp(-x))
__device__ __forceinline__ float am_activate_rec(float v, int act) {
    if (act == 0) return fmaxf(v, 0.0f);
    if (act == 1) return am_sigmoid_rec(v);
    return am_tanh_rec(v);
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

// One block tile's products: acc = A[m0.., :] * W[n0.., :]^T over nk stages of BK.  AUX = 16 loads the A operand with sc1 (rows
// another workgroup of the SAME launch stored write-through, gemm_stack_kernel); SWAP feeds W as the MFMA's row operand, so a
// lane ends up holding 4 consecutive COLUMNS of one output row (C/D map transposed) -- every element is the same fmaf chain in k.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int BN, int BK, int AUX, bool SWAP, class Hook = NoHook>
__device__ __forceinline__ void gemm_tile_products(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int m0, int n0, int nk,
                                                   int row_p, int row_skip, char* smem, f32x16 (&acc)[GEMM_BM / GEMM_WM / 32][BN / GEMM_WN / 32],
                                                   Hook after_first_barrier = Hook())
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / GEMM_WN, wn = wave % GEMM_WN;

    const float* ga[A_PER_WAVE];
    const float* gw[W_PER_WAVE];
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / CH;
        const int c = (lane % CH) ^ ((r >> SW_SHIFT) & SW_MASK);
        // TDNN layers (row_p > 0): output row R = (frame, position) reads the window starting at input row
        // R + (R / row_p) * row_skip of the channel-last activation: the K extent spans the kernel's taps, no im2col copy
        const int R = m0 + r;
        const size_t src = row_p > 0 ? (size_t)R + (size_t)(R / row_p) * row_skip : (size_t)R;
        ga[j] = A + src * lda + c * 4;
    }
#pragma unroll
    for (int j = 0; j < W_PER_WAVE; ++j) {
        const int r = (j * NW + wave) * RPI + lane / CH;
        const int c = (lane % CH) ^ ((r >> SW_SHIFT) & SW_MASK);
        gw[j] = W + (size_t)(n0 + r) * ldw + c * 4;
    }
    auto issue = [&](char* sbase, int k0, int l) {   // l-th LDS-DMA piece of a stage (wave-uniform LDS address)
        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(ga[l] + k0), AM_LDS_PTR(sbase + (l * NW + wave) * 1024), 16, 0, AUX);
        else __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(gw[l - A_PER_WAVE] + k0), AM_LDS_PTR(sbase + BM * BK * 4 + ((l - A_PER_WAVE) * NW + wave) * 1024), 16, 0, 0);
    };

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
        if (kt == 0) after_first_barrier();                // (every wave has drained ALL its earlier memory operations here)
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
                        acc[mi][ni] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t & 1][ni][j], av[t & 1][mi][j], acc[mi][ni], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][mi][j], bv[t & 1][ni][j], acc[mi][ni], 0, 0, 0);
                const int slot = t * 4 + j;
                if (slot % EVERY == 0 && slot / EVERY < LOADS) issue(sbase, k0, slot / EVERY);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail prefetch before the wave ends
}

template <int BN>
__device__ __forceinline__ void gemm_tile_epilogue(const f32x16 (&acc)[GEMM_BM / GEMM_WM / 32][BN / GEMM_WN / 32], const float* __restrict__ bias,
                                                   float* __restrict__ C, int ldc, int m0, int n0, int act, int epi, float alpha)
{
    constexpr int MI = GEMM_BM / GEMM_WM / 32, NI = BN / GEMM_WN / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / GEMM_WN, wn = wave % GEMM_WN, li = lane & 31, lh = lane >> 5;
    // one uniform dispatch on the mode, then straight-line code
    const int erow = m0 + wm * (GEMM_BM / GEMM_WM) + 4 * lh, ecol = n0 + wn * (BN / GEMM_WN) + li;
    if (epi == EPI_ACT) {
        if (act == 0) gemm_epilogue<0, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
        else if (act == 1) gemm_epilogue<1, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
        else gemm_epilogue<2, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    } else if (epi == EPI_LINEAR) gemm_epilogue<3, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    else if (epi == EPI_SCALE) gemm_epilogue<4, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
    else gemm_epilogue<5, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha);
}

template <int BN, int BK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    float* __restrict__ C, int ldc, int tiles_n, int nk, int act, int epi, float alpha, int row_p, int row_skip, int* __restrict__ queue, int n_tiles)
{
    constexpr int BM = GEMM_BM;
    constexpr int MI = BM / GEMM_WM / 32, NI = BN / GEMM_WN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous run
    // of tiles, n fastest, so the blocks resident on one XCD share A panels and W in its L2.
    // Work-queue mode (queue != nullptr): the blocks pull tiles from a global counter instead.  Workgroups are dealt to the XCDs
    // statically, so when other kernels hold some XCDs (the persistent recurrent kernel's last row groups), the blocks dealt to
    // the free XCDs do ALL the work and the late-comers find the queue empty; which block computes a tile never changes a bit.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    for (;;) {
    if (queue) {           // the tile number travels through the first word of the (idle) stage buffers
        __syncthreads();   // the previous tile's LDS reads are done
        if (threadIdx.x == 0) *(volatile int*)smem = atomicAdd(queue, 1);
        __syncthreads();
        tile = *(volatile int*)smem;
        __syncthreads();   // everyone has read it before the first LDS-DMA piece lands there
        if (tile >= n_tiles) break;
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    f32x16 acc[MI][NI];
    gemm_tile_products<BN, BK, 0, false>(A, lda, W, ldw, m0, n0, nk, row_p, row_skip, smem, acc);
    gemm_tile_epilogue<BN>(acc, bias, C, ldc, m0, n0, act, epi, alpha);
    if (!queue) break;
    }
}

constexpr int GEMM_LDS_256 = 2 * (GEMM_BM + 256) * 32 * 4;  // 128 KiB
constexpr int GEMM_LDS_384 = 2 * (GEMM_BM + 384) * 16 * 4;  //  80 KiB
constexpr int GEMM_LDS_192 = 2 * (GEMM_BM + 192) * 32 * 4;  // 112 KiB

hipError_t init_split_kernels();
hipError_t init_lstm_kernels();
hipError_t init_rec_kernels();
hipError_t init_stack_kernel();

hipError_t init_kernels()
{
    hipError_t se = init_split_kernels();
    if (se != hipSuccess) return se;
    se = init_lstm_kernels();
    if (se != hipSuccess) return se;
    se = init_rec_kernels();
    if (se != hipSuccess) return se;
    // per device: allow the GEMM's large dynamic LDS
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_f32_kernel<256, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_256);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)gemm_nt_f32_kernel<384, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_384);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)gemm_nt_f32_kernel<192, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_192);
    if (e != hipSuccess) return e;
    return init_stack_kernel();
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
        const int tiles_n = g.N_pad / 256, n_tiles = tiles_m * tiles_n;
        const int grid = g.queue ? std::min(n_tiles, g.n_cus) : n_tiles;     // work-queue mode: one block per CU slot is enough
        hipLaunchKernelGGL((gemm_nt_f32_kernel<256, 32>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_256, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 32, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles);
    } else if (g.N_pad % 192 == 0 && !getenv("AM_GEMM_384")) {
        // 1909 classes -> 1920 = 10 x 192: half the tile of the 384-wide form (the round-1 choice, kept behind AM_GEMM_384 for A/B runs) --
        // 96 accumulator registers instead of 192, no spills, and twice as many tiles to fill the last round of a launch with
        const int tiles_n = g.N_pad / 192, n_tiles = tiles_m * tiles_n;
        const int grid = g.queue ? std::min(n_tiles, g.n_cus) : n_tiles;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<192, 32>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_192, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 32, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles);
    } else if (g.N_pad % 384 == 0) {
        const int tiles_n = g.N_pad / 384, n_tiles = tiles_m * tiles_n;
        const int grid = g.queue ? std::min(n_tiles, 2 * g.n_cus) : n_tiles;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<384, 16>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_384, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 16, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The whole Linear stack of one network in ONE launch (gemm_stack_kernel): a work queue over (layer, row tile, column tile).
//
// Why: a layer launched on its own runs in whole rounds of 256 tiles -- 57,919 rows x 1024 = 908 tiles = 3.55 rounds, run as
// 4 -- and every launch ends in a ragged tail.  Row tile r of layer l+1 needs only row tile r of layer l, so here the 256
// resident workgroups pull items from per-XCD counters, layer-major, and the tail of layer l runs beside the head of layer l+1:
// one partial round per network instead of one per layer.
//   item  = atomicAdd(queue of this XCD)   (eight queues: queue x holds the row tiles x, x + 8, ...; layer-major, column tile
//                                           fastest, so the workgroups of an XCD share A panels in its L2; a dry queue -> the next XCD's)
//   wait  : done[l-1][row tile] == column tiles of layer l-1        (one lane polls, sc1; then the workgroup's barrier)
//   tile  : the same products and epilogue as gemm_nt_f32_kernel -- which workgroup computes a tile never changes a bit
//   hand-off: hidden activations are stored write-through (sc1) and read by sc1 LDS-DMA; after every storing wave's
//             vmcnt(0) and the workgroup's barrier ONE lane adds 1 to done[l][row tile] (agent scope) -- behind the first stage
//             barrier of the NEXT tile, where that drain has happened anyway; the next item is pulled there too, a tile ahead
//             (MI355X_MICROARCH.md, inter-workgroup visibility, first table row).
// Every hidden layer has its OWN activation plane: a plane re-written inside the launch could be served stale from an XCD's
// L2 that read it two layers earlier.  No deadlock without co-residency: a queue's items are pulled in order and an item only
// waits for items of the same queue with smaller numbers, which resident workgroups hold (and a workgroup that has to wait
// signals the tile it holds back first).  Waits are bounded like the recurrent kernel's (abort flag).
// ---------------------------------------------------------------------------------------------------
// Epilogue of a hidden tile: act(acc + bias), stored write-through (sc1) for the workgroups that read it in this launch.
// With the transposed C/D map (SWAP) a lane holds 4 consecutive columns of one row, but storing those directly writes every
// 128-B line in four 32-B pieces, and a write-through store of part of a line is a fabric write of its own: measured 11 us per
// 256 x 256 tile (the whole headline pass 3.3 % slower than with plain stores).  So the block goes through LDS once and leaves
// as whole lines -- a wave's 64 x 128 block as 2 rows x 512 B per store instruction -- which costs the same as plain stores.
// Each wave stages 32 rows x 128 columns at a time in its own 17-KB slice (row stride 136 floats: conflict-free b128 writes).
constexpr int GEMM_STACK_STAGE_LD = 136;
constexpr int GEMM_STACK_LDS = (GEMM_WM * GEMM_WN) * 32 * GEMM_STACK_STAGE_LD * 4;   // 139,264 B >= the products' two stages
// MODE: 0 relu, 1 sigmoid, 2 tanh (hidden layers, stored write-through: AUX = 16); 5 accumulate into C (an ensemble's second and later
// members: C = C + (acc + bias), plain loads and stores, AUX = 0 -- as whole lines the read-modify-write is 1.8 % of a 10-member
// pass faster than gemm_epilogue's 4-byte accesses).  NI = 4 (256-wide tiles: a wave's rows are 512 B) or 3 (192-wide: 384 B).
template <int MODE, int AUX, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_handoff_lines(const f32x16 (&acc)[MI][NI], const float* __restrict__ bias, float* __restrict__ Ctile, int ldc,
                                                            int wave, int wm, int wn, int lane, char* smem)
{
    constexpr int LD = NI * 32 + 8;                 // 136 / 104 floats: = 8 (mod 16), conflict-free b128 writes down a column of rows
    static_assert(LD <= GEMM_STACK_STAGE_LD, "a wave's staging slice holds 32 rows");
    const int li = lane & 31, lh = lane >> 5;
    float* stage = (float*)smem + wave * (32 * GEMM_STACK_STAGE_LD);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Ctile, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = ni * 32 + 8 * g + 4 * lh;
                const f32x4 bv = *(const f32x4*)(bias + wn * (NI * 32) + col);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[mi][ni][4 * g + j] + bv[j];
                    if (MODE == 0) x = fmaxf(x, 0.0f);
                    else if (MODE == 1) x = am_sigmoid(x);
                    else if (MODE == 2) x = tanhf(x);
                    v[j] = x;
                }
                *(f32x4*)(stage + li * LD + col) = v;
            }
        }
        // out as whole 128-B lines: 8 lanes per line, 8 lines per store instruction
#pragma unroll
        for (int i = 0; i < 4 * NI; ++i) {
            const int sgm = 8 * i + (lane >> 3);            // 128-B segment of the 32 x NI segments of this pass
            const int row = sgm / NI, col = (sgm - row * NI) * 32 + 4 * (lane & 7);
            f32x4 v = *(const f32x4*)(stage + row * LD + col);
            const unsigned off = (unsigned)(wm * (GEMM_BM / GEMM_WM) + mi * 32 + row) * (unsigned)ldc + (unsigned)(wn * (NI * 32) + col);
            if (MODE == 5) {
                const f32x4 old = *(const f32x4*)(Ctile + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = old[j] + v[j];
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(off * 4u), 0, AUX);
        }
    }
}

// OUT_ACCUM: the instantiation for an ensemble's second and later members (output epilogue C = C + ...): a kernel of its own so that
// its extra tile code does not cost the single-network kernel registers (0.6 % on the headline when both lived in one kernel).
template <bool OUT_192, bool STAMPS, bool OUT_ACCUM>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_stack_kernel(const GemmStackArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_queue, s_item, s_ready, s_abort;
    const int tid = threadIdx.x;
    if (tid == 0) s_abort = 0;
    // tid 0's bookkeeping
    int publish = -1;                                  // counter of the hidden tile this workgroup stored last, not yet signalled
    int cur_q = blockIdx.x & 7, stolen = 0;            // the queue it pulls from (its XCD's own until that runs dry)
    int pulled = tid == 0 ? atomicAdd(a.queues + cur_q * GEMM_STACK_QSTRIDE, 1) : 0;
    // diagnostic timeline (the STAMPS instantiation only; a.stamps): per workgroup its start / end, tiles, waits, and where a tile's time goes
    unsigned long long st_start = 0, st_wait = 0, st_last = 0, st_t = 0, st_ph[4] = {0, 0, 0, 0}; int st_tiles = 0, st_stolen = 0, st_waits = 0;
    auto phase = [&](int k) { if (STAMPS && tid == 0) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (k >= 0) st_ph[k] += now - st_t; st_t = now; } };
    if (STAMPS && tid == 0) st_start = __builtin_amdgcn_s_memrealtime();
    auto rows_of = [&](int q) { return a.tiles_m > q ? (a.tiles_m - q + 7) >> 3 : 0; };   // row tiles of queue q: q, q + 8, ...
    auto signal = [&] { if (publish >= 0) { __hip_atomic_fetch_add(a.done + publish, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); publish = -1; } };
    for (;;) {
        phase(-1);
        __syncthreads();                               // the previous iteration's reads of s_* are done
        if (tid == 0) {
            while (stolen < 8 && pulled >= rows_of(cur_q) * a.cum_total) {   // this queue is dry: move on to the next XCD's
                ++stolen;
                cur_q = (cur_q + 1) & 7;
                if (stolen < 8) pulled = atomicAdd(a.queues + cur_q * GEMM_STACK_QSTRIDE, 1);
            }
            s_queue = stolen < 8 ? cur_q : -1;
            s_item = pulled;
        }
        __syncthreads();
        const int q = s_queue, item = s_item;
        if (q < 0) break;
        const int rq = rows_of(q);
        int l = 0;
#pragma unroll
        for (int j = 1; j <= GEMM_STACK_MAX_HIDDEN; ++j) if (j <= a.n_hidden && item >= rq * a.L[j].cum_n) l = j;
        const GemmStackLayer& Ly = a.L[l];
        const int local = item - rq * Ly.cum_n;
        const int r = local / Ly.tiles_n, tn = local - r * Ly.tiles_n, tm = q + 8 * r;
        if (l > 0) {
            // this item's operand rows: finished?  If not, signal the tile still held back FIRST (two workgroups each waiting for
            // the other's unsignalled tile would never end), then wait -- bounded
            const unsigned* c = a.done + (size_t)(l - 1) * a.tiles_m + tm;
            const unsigned target = (unsigned)(a.L[l - 1].tiles_n + a.test_extra);
            if (tid == 0) s_ready = ld_agent_u32(c) >= target;
            __syncthreads();
            if (!s_ready) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    signal();
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    if (STAMPS) ++st_waits;
                    int it = 0;
                    while (ld_agent_u32(c) < target) {
                        __builtin_amdgcn_s_sleep(4);
                        if ((++it & 63) == 0 && (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                                __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)LSTM_P_TIMEOUT)) {
                            __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_abort = 1;
                            break;
                        }
                    }
                    if (STAMPS) st_wait += __builtin_amdgcn_s_memrealtime() - t0;
                }
                __syncthreads();
                if (s_abort) return;
            }
        }
        // behind the first stage's barrier every wave has drained its stores of the previous tile: signal that tile there, and
        // pull the next item (its number is needed a whole tile later)
        auto hook = [&] { if (tid == 0) { phase(1); signal(); pulled = atomicAdd(a.queues + cur_q * GEMM_STACK_QSTRIDE, 1); } };
        const float* A = l > 0 ? a.L[l - 1].C : a.A0;
        const int lda = l > 0 ? a.L[l - 1].ldc : a.lda0;
        const int m0 = tm * GEMM_BM;
        if (STAMPS && tid == 0) { ++st_tiles; st_stolen += stolen > 0; st_last = __builtin_amdgcn_s_memrealtime(); }
        phase(0);   // pull + decode + dependency check
        if (l < a.n_hidden) {
            constexpr int MI = GEMM_BM / GEMM_WM / 32, NI = 256 / GEMM_WN / 32;
            f32x16 acc[MI][NI];
            const int n0 = tn * 256;
            gemm_tile_products<256, 32, 16, true>(A, lda, Ly.W, Ly.ldw, m0, n0, Ly.nk, 0, 0, smem, acc, hook);
            phase(2);   // first barrier -> last MFMA
            const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            float* Ct = Ly.C + (size_t)m0 * Ly.ldc + n0;
            __syncthreads();   // every wave has read the last stage: the stage buffers become the waves' staging slices
            if (a.act == 0) gemm_epilogue_handoff_lines<0, 16, MI, NI>(acc, Ly.bias + n0, Ct, Ly.ldc, wave, wave / GEMM_WN, wave % GEMM_WN, lane, smem);
            else if (a.act == 1) gemm_epilogue_handoff_lines<1, 16, MI, NI>(acc, Ly.bias + n0, Ct, Ly.ldc, wave, wave / GEMM_WN, wave % GEMM_WN, lane, smem);
            else gemm_epilogue_handoff_lines<2, 16, MI, NI>(acc, Ly.bias + n0, Ct, Ly.ldc, wave, wave / GEMM_WN, wave % GEMM_WN, lane, smem);
            if (tid == 0) publish = l * a.tiles_m + tm;
            phase(3);   // epilogue
        } else {
            // 192-wide tiles where the class count pads to a multiple of 384 (1909 -> 1920 = 10 x 192): the last items of the launch are
            // output tiles, and the ragged end of the launch costs about half an item per workgroup -- 0.19 ms items instead of 0.37
            constexpr int BN = OUT_192 ? 192 : 256, BK = 32;
            constexpr int MI = GEMM_BM / GEMM_WM / 32, NI = BN / GEMM_WN / 32;
            f32x16 acc[MI][NI];
            const int n0 = tn * BN;
            if (OUT_ACCUM) {   // an ensemble's later members: the read-modify-write of the running sum as whole lines
                gemm_tile_products<BN, BK, 16, true>(A, lda, Ly.W, Ly.ldw, m0, n0, Ly.nk, 0, 0, smem, acc, hook);
                phase(2);
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                __syncthreads();
                gemm_epilogue_handoff_lines<5, 0, MI, NI>(acc, Ly.bias + n0, Ly.C + (size_t)m0 * Ly.ldc + n0, Ly.ldc, wave, wave / GEMM_WN, wave % GEMM_WN, lane, smem);
            } else {
                gemm_tile_products<BN, BK, 16, false>(A, lda, Ly.W, Ly.ldw, m0, n0, Ly.nk, 0, 0, smem, acc, hook);
                phase(2);
                gemm_tile_epilogue<BN>(acc, Ly.bias, Ly.C, Ly.ldc, m0, n0, a.act, a.out_epi, a.out_alpha);   // plain stores: read by later launches only
            }
            phase(3);
        }
    }
    // every queue is dry: the last hidden tile this workgroup stored still has to be signalled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        signal();
        if (STAMPS) {
            unsigned long long* o = a.stamps + (size_t)blockIdx.x * 16;
            o[8] = st_ph[0]; o[9] = st_ph[1]; o[10] = st_ph[2]; o[11] = st_ph[3];
            o[0] = st_start; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = st_tiles; o[3] = st_wait; o[4] = st_waits; o[5] = st_stolen; o[6] = st_last;
            o[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 0..3
        }
    }
}

hipError_t init_stack_kernel()
{
    static_assert(GEMM_STACK_LDS >= GEMM_LDS_256 && GEMM_STACK_LDS >= GEMM_LDS_384, "the staging slices cover the products' stages");
    hipError_t e = hipSuccess;
#define AM_STACK_ATTR(W, S, C) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_stack_kernel<W, S, C>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_STACK_LDS)
    AM_STACK_ATTR(false, false, false); AM_STACK_ATTR(true, false, false); AM_STACK_ATTR(false, false, true); AM_STACK_ATTR(true, false, true);
    AM_STACK_ATTR(false, true, false); AM_STACK_ATTR(true, true, false);
#undef AM_STACK_ATTR
    return e;
}

hipError_t launch_gemm_stack(hipStream_t s, const GemmStackArgs& a, int out_n_pad, int n_cus)
{
    if (a.n_hidden < 1 || a.n_hidden > GEMM_STACK_MAX_HIDDEN || a.tiles_m <= 0 || a.cum_total <= 0 || n_cus <= 0) return hipErrorInvalidValue;
    const int grid = (int)std::min<long long>((long long)a.tiles_m * a.cum_total, n_cus);   // one workgroup per CU (speed only: the queue needs no co-residency)
    const bool w192 = out_n_pad % 256 != 0;
    if (w192 && out_n_pad % 192 != 0) return hipErrorInvalidValue;
#define AM_STACK_LAUNCH(W, S, C) hipLaunchKernelGGL((gemm_stack_kernel<W, S, C>), dim3(grid), dim3(GEMM_THREADS), GEMM_STACK_LDS, s, a)
    const bool accum = a.out_epi == EPI_ACCUM;
    if (a.stamps && !accum) { if (w192) AM_STACK_LAUNCH(true, true, false); else AM_STACK_LAUNCH(false, true, false); }
    else if (accum) { if (w192) AM_STACK_LAUNCH(true, false, true); else AM_STACK_LAUNCH(false, false, true); }
    else if (w192) AM_STACK_LAUNCH(true, false, false);
    else AM_STACK_LAUNCH(false, false, false);
#undef AM_STACK_LAUNCH
    return hipGetLastError();
}

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

// ---------------------------------------------------------------------------------------------------
// Fused splice + Kaldi feature transform + i-vector passthrough.  HBM-bound: every output element is written
// once (coalesced, 16 B per lane), every input row is read n_shifts times out of L1/L2.
// out[r, w*D + d] = (x[clamp(first + r + shift_w, 0, ext_rows-1), d] + add[w*D+d]) * scale[w*D+d]
// out[r, n_shifts*D + j] = ivec[row0 + r, j];  remaining columns up to ldo and rows [n, n_pad) are zero.
// VEC = 4 needs D % 4 == 0 and d_iv % 4 == 0 (then no float4 straddles a segment); VEC = 1 is the general form.
// ---------------------------------------------------------------------------------------------------
// HALVES: write the result as hi/lo f16 halves (split-f16 precision mode: hi at out_hi, lo at out_lo, same [n_pad, ldo] shape)
// instead of f32, raising *overflow when a value does not fit f16.
template <int VEC, bool HALVES>
__global__ __launch_bounds__(256) void splice_ft_kernel(
    const float* __restrict__ x, long long ext_rows, long long first, const float* __restrict__ ivec, long long row0,
    long long n, long long n_pad, int D, int n_shifts, const int* __restrict__ shifts,
    const float* __restrict__ add, const float* __restrict__ scale, int d_iv, float* __restrict__ out, int ldo,
    int rows_per_block, _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo, int* __restrict__ overflow)
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

        // the common case first: the spliced feature columns of a block of real rows, four rows' loads in flight per lane before
        // the first store (one row at a time left the kernel at 3.6 TB/s, latency-bound: MI355X_MICROARCH.md, keep >= 4 loads per lane)
        int rr0 = rsub;
        if (!HALVES && VEC == 4 && kind == 0 && rblock + rows_per_block <= n) {
            for (; rr0 + 3 * rows_per_pass < rows_per_block; rr0 += 4 * rows_per_pass) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    long long src = first + rblock + rr0 + u * rows_per_pass + shift;
                    src = src < 0 ? 0 : (src > ext_rows - 1 ? ext_rows - 1 : src);
                    t[u] = *(const f32x4*)(x + src * D + d);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (has_ft) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float s1 = t[u][e] + av[e]; t[u][e] = s1 * sv[e]; }
                    }
                    *(f32x4*)(out + (rblock + rr0 + u * rows_per_pass) * ldo + col) = t[u];
                }
            }
        }
        for (int rr = rr0; rr < rows_per_block; rr += rows_per_pass) {
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
            if (HALVES) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                _Float16 hh[VEC], ll[VEC];
                bool bad = false;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    hh[e] = (_Float16)v[e];
                    ll[e] = (_Float16)(v[e] - (float)hh[e]);
                    bad |= !(fabsf(v[e]) <= SPLIT_F16_LIMIT);
                }
                if (VEC == 4) {
                    *(h4*)(out_hi + r * ldo + col) = (h4){hh[0], hh[1], hh[2], hh[3]};
                    *(h4*)(out_lo + r * ldo + col) = (h4){ll[0], ll[1], ll[2], ll[3]};
                } else { out_hi[r * ldo + col] = hh[0]; out_lo[r * ldo + col] = ll[0]; }
                if (bad) *overflow = 1;
            } else {
                float* o = out + r * ldo + col;
                if (VEC == 4) { f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *(f32x4*)o = t; }
                else o[0] = v[0];
            }
        }
    }
}

hipError_t launch_splice_ft(hipStream_t s, const SpliceArgs& a)
{
    if (a.n_pad <= 0) return hipSuccess;
    const bool vec = (a.D % 4 == 0) && (a.d_iv % 4 == 0) && (a.ldo % 4 == 0);
    const int rows_per_block = 32;
    const long long blocks = (a.n_pad + rows_per_block - 1) / rows_per_block;
#define AM_SPLICE(V, H) hipLaunchKernelGGL((splice_ft_kernel<V, H>), dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first, \
        a.ivec, (long long)a.row0, (long long)a.n, (long long)a.n_pad, a.D, a.n_shifts, a.shifts, a.add, a.scale, a.d_iv, a.out, a.ldo, rows_per_block, \
        (_Float16*)a.out_hi, (_Float16*)a.out_lo, a.overflow)
    if (a.out_hi) { if (vec) AM_SPLICE(4, true); else AM_SPLICE(1, true); }
    else { if (vec) AM_SPLICE(4, false); else AM_SPLICE(1, false); }
#undef AM_SPLICE
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

// n_in > 1: the members of an ensemble left their logits in n_in buffers (in + k * in_stride); they are combined here in member order with
// the float operations of the one-buffer path -- v = in_0 (* first_scale: a master's weight, NNWithRPL); v = v + in_k -- or, with
// member_norm (the dev-mode mean of log-posteriors, predict_folds.py:207-212), v = (in_0 - logsum(in_0)); v = v + (in_k - logsum(in_k)).
// (MULTI is an instantiation of its own: its second row of registers costs the one-buffer path occupancy -- 530 -> 622 us on the headline's
// 196,608 x 1909 rows when both lived in one kernel)
template <int NIT, bool MULTI>
__global__ __launch_bounds__(256) void finish_kernel(
    const float* __restrict__ in, int ldin, float* __restrict__ out, long long ldout, const int* __restrict__ row_map,
    long long n, int C, float divisor, const float* __restrict__ rpl_w, const float* __restrict__ rpl_b,
    const float* __restrict__ rpl_lb, const float* __restrict__ prior, int normalise, int accumulate,
    int n_in, long long in_stride, float first_scale, int member_norm)
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
    if (MULTI && n_in > 1) {
        if (first_scale != 1.0f) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] * first_scale;
        }
        if (member_norm) {
            const float lse = row_logsum<NIT>(v, lane, C);
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] - lse;
        }
        for (int k = 1; k < n_in; ++k) {
            float w[NIT];
            const float* sk = src + (long long)k * in_stride;
#pragma unroll
            for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; w[it] = c < C ? sk[c] : 0.0f; }
            if (member_norm) {
                const float lse = row_logsum<NIT>(w, lane, C);
#pragma unroll
                for (int it = 0; it < NIT; ++it) w[it] = w[it] - lse;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] + w[it];
        }
    }

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
#define AM_FINISH_ARGS a.in, a.ldin, a.out, (long long)a.ldout, a.row_map, (long long)a.n, a.C, a.divisor, a.rpl_w, a.rpl_b, a.rpl_lb, a.prior, a.normalise, \
        a.accumulate, a.n_in, (long long)a.in_stride, a.first_scale, a.member_norm
#define AM_FINISH(N) do { if (a.n_in > 1) hipLaunchKernelGGL((finish_kernel<N, true>), dim3(blocks), dim3(256), 0, s, AM_FINISH_ARGS); \
                          else hipLaunchKernelGGL((finish_kernel<N, false>), dim3(blocks), dim3(256), 0, s, AM_FINISH_ARGS); } while (0)
    if (nit <= 1) AM_FINISH(1);
    else if (nit <= 2) AM_FINISH(2);
    else if (nit <= 4) AM_FINISH(4);
    else if (nit <= 8) AM_FINISH(8);
    else if (nit <= 16) AM_FINISH(16);
    else if (nit <= 30) AM_FINISH(30);
    else if (nit <= 32) AM_FINISH(32);
    else return hipErrorInvalidValue;
#undef AM_FINISH
#undef AM_FINISH_ARGS
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
// TDNN frontend.  The reference splices +-s frames, applies the (tiled centre-frame) transform and then VIEWS each row
// of W*D values as [C = D channels][W positions] (chainer_networks.py:39: F.reshape(x, (B, -1, 1, input_win_size))), i.e.
// channel c, position p is flat element f = c*W + p of the spliced row -- which came from frame shift f / D, dimension
// f % D.  The convolutions run here as GEMMs over channel-last activations, so this kernel writes
//   out[(n*W + p) * Cp + c] = (x[clamp(first + n + shift[f / D]), f % D] + add[f]) * scale[f],   f = c*W + p
// with zero channels c >= D (Cp = D rounded up to 32) and zero rows for n >= n_real.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splice_tdnn_kernel(
    const float* __restrict__ x, long long ext_rows, long long first, long long n_real, long long n_rows, int D, int W,
    const int* __restrict__ shifts, const float* __restrict__ add, const float* __restrict__ scale, float* __restrict__ out, int Cp)
{
    const long long total = n_rows * W * Cp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % Cp);
        const long long np = idx / Cp;
        const int p = (int)(np % W);
        const long long n = np / W;
        float v = 0.0f;
        if (c < D && n < n_real) {
            const int f = c * W + p;
            long long r = first + n + shifts[f / D];
            r = r < 0 ? 0 : (r >= ext_rows ? ext_rows - 1 : r);
            v = x[r * D + f % D];
            if (add) { const float s1 = v + add[f]; v = s1 * scale[f]; }
        }
        out[idx] = v;
    }
}

hipError_t launch_splice_tdnn(hipStream_t s, const SpliceArgs& a, int W, int Cp, long long n_rows)
{
    if (n_rows <= 0) return hipSuccess;
    long long blocks = (n_rows * W * Cp + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(splice_tdnn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first,
                       (long long)a.n, n_rows, a.D, W, a.shifts, a.add, a.scale, a.out, Cp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent path, stage 2: one WAVEFRONT step of the LSTM stack.  Launch d computes, for every layer l, time step
// t = d - l: the dependency chain of the whole stack is Tmax + L - 1 launches instead of Tmax * L (the path is
// latency-bound: ~8 us per dependent launch).  Per layer and step, for the active utterances:
//   g = [G[row] | bias + Hin[row] * Wup^T] + Hprev[row] * Wlat^T ;  c' = tanh(a) sig(i) + sig(f) c ;  h' = sig(o) tanh(c')
// Layer 0 takes its upward term G from the hoisted bulk GEMM (its input is known up front); layers >= 1 fuse the upward
// projection of the previous layer's h at the same t into the step (K = U + U).
// Gate-tiled layout (host permutes weights): columns [64*j + 16*k + u'] hold gate k (a,i,f,o) of unit 16*j+u'.
// 512 threads = 8 waves: wave = gate + 4*khalf (v_mfma_f32_16x16x4_f32; the K halves are combined through LDS).
//   * loads the epilogue needs (G / bias, previous c) are issued first, then the A chunk, then the wave's weight
//     fragments (each weight element is used by exactly one wave of the block: straight to VGPRs); vmcnt is in-order,
//     so the A chunk is waited for while the weight loads are still in flight;
//   * A goes through LDS (shared by the 8 waves; rows padded by 16 B -> conflict-free ds_read_b128) behind a raw
//     s_barrier that does not drain the weight loads;
//   * block -> (unit tile, row tile) keeps the row tiles of a unit tile on one XCD (blockIdx % 8) for L2 reuse.
// K permutation as in the GEMM: 4 consecutive k per 16-B load feed 4 MFMAs, A and W alike.
// ---------------------------------------------------------------------------------------------------
// Block = 64 utterance rows (4 MFMA row tiles) x 16 units x 4 gates; the wave's weight fragments stay in registers and
// are reused by the 4 row tiles (4x less weight traffic per row than a 16-row block: the launch is L2-bandwidth-bound
// otherwise).  K is walked in batches of LSTM_KB = 128 (64 per K half): the next batch's A chunk and weight fragments are
// prefetched into registers while the current batch is multiplied.
constexpr int LSTM_RB = 64;               // rows per block
constexpr int LSTM_KB = 128;              // k per batch (both K halves)
constexpr int LSTM_KS = LSTM_KB / 32;     // 16-k sub-steps per wave per batch
constexpr int LSTM_LDA = LSTM_KB + 8;     // LDS row stride (floats): +32 B -> ds_read_b128 of (row l15, chunk lq) is conflict-free in every 16-lane service
                                          // group (+16 B left a 2-way conflict per group: SQ_LDS_BANK_CONFLICT, profiles/r2b)
constexpr int LSTM_APT = LSTM_RB * LSTM_KB / 4 / 512;  // 16-B A chunks per thread per batch

template <int NT>
__device__ __forceinline__ void lstm_batch_mma(const float* sa, const f32x4 (&wc)[LSTM_KS], f32x4 (&acc)[4])
{
    f32x4 pb[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < LSTM_KS; ++s) {
        f32x4 a[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r) a[r] = *(const f32x4*)(sa + r * 16 * LSTM_LDA + 16 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j)   // j outer, row tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int r = 0; r < NT; ++r) pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], wc[s][j], pb[r], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] += pb[r];
}

__global__ __launch_bounds__(512, 4) void lstm_wavefront_kernel(const LstmWavefrontArgs args)
{
    __shared__ __attribute__((aligned(16))) float sA[LSTM_RB * LSTM_LDA];
    __shared__ float sg[2 * 4 * LSTM_RB * 17];
    int li = 0;
#pragma unroll
    for (int l = 1; l < LSTM_MAX_LAYERS; ++l) if (l < args.n && (int)blockIdx.x >= args.L[l].block0) li = l;
    const LstmLayerStep& Ls = args.L[li];
    const int U_pad = args.U_pad, ut_groups = args.ut_groups;
    const int bid = blockIdx.x - Ls.block0;
    const int ut = (bid & 7) + 8 * ((bid >> 3) % ut_groups);
    const int rt = (bid >> 3) / ut_groups;
    if (ut * 16 >= U_pad) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ldg = 4 * U_pad;
    const int first = Ls.first, n_active = Ls.n_active;
    // this block's live 16-row tiles: [rt*tpb, rt*tpb + ntile) (tpb <= 4 is chosen by the launcher to fill the chip)
    const int row0 = rt * Ls.tpb * 16;
    const int tiles_left = (n_active - row0 + 15) / 16;
    const int ntile = tiles_left < Ls.tpb ? tiles_left : Ls.tpb;

    // loads the epilogue needs, issued first: they ride along with the operand loads instead of adding round trips
    float gpre[2][4], cprev[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + 512 * e, erow = idx >> 4, eu = idx & 15, slot = row0 + erow;
        const bool live = erow < ntile * 16 && slot < n_active;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            gpre[e][k] = !live ? 0.0f : (Ls.G ? Ls.G[(size_t)slot * ldg + ut * 64 + k * 16 + eu] : Ls.bias[ut * 64 + k * 16 + eu]);
        cprev[e] = (!first && live) ? Ls.Cst[(size_t)slot * U_pad + ut * 16 + eu] : 0.0f;
    }

    // batches: segment 0 = upward projection of this step's input rows (layers >= 1), segment 1 = lateral term (t > 0)
    const int nb0 = Ls.Hin ? Ls.Kin_pad / LSTM_KB : 0;
    const int nb1 = first ? 0 : U_pad / LSTM_KB;
    const int nb = nb0 + nb1;
    auto batch_ptrs = [&](int b, const float*& ap, const float*& wp, int& K) {
        if (b < nb0) { K = Ls.Kin_pad; ap = Ls.Hin + (size_t)row0 * K + b * LSTM_KB; wp = Ls.Wup + b * LSTM_KB; }
        else { K = U_pad; ap = Ls.Hprev + (size_t)row0 * K + (b - nb0) * LSTM_KB; wp = Ls.Wlat + (b - nb0) * LSTM_KB; }
        wp += (size_t)(ut * 64 + gate * 16 + l15) * K + kh * (LSTM_KB / 2) + 4 * lq;
    };
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 av[LSTM_APT], wv[LSTM_KS];
    auto prefetch = [&](int b) {
        const float* ap; const float* wp; int K;
        batch_ptrs(b, ap, wp, K);
#pragma unroll
        for (int e = 0; e < LSTM_APT; ++e) {   // A chunk first: in-order vmcnt lets it be waited for before the weights
            const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);  // chunk e = row tile e
            if (e < ntile) av[e] = *(const f32x4*)(ap + (size_t)r * K + 4 * c4);
        }
#pragma unroll
        for (int s = 0; s < LSTM_KS; ++s) wv[s] = *(const f32x4*)(wp + 16 * s);
    };
    if (nb > 0) prefetch(0);
    for (int b = 0; b < nb; ++b) {
        if (b > 0) __syncthreads();           // everyone is done reading the previous batch's A tile
#pragma unroll
        for (int e = 0; e < LSTM_APT; ++e) {
            const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);
            if (e < ntile) *(f32x4*)(sA + r * LSTM_LDA + 4 * c4) = av[e];
        }
        f32x4 wc[LSTM_KS];
#pragma unroll
        for (int s = 0; s < LSTM_KS; ++s) wc[s] = wv[s];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (b + 1 < nb) prefetch(b + 1);      // next batch flies while this one is multiplied
        // every 64-k chain (this wave's half of the batch) starts from zero and is then folded into the running sum:
        // the summation order is a property of the K index alone, shared with lstm_thin_kernel, so a frame's
        // posteriors do not depend on how many utterances happen to share its launch
        const float* sa = sA + l15 * LSTM_LDA + kh * (LSTM_KB / 2) + 4 * lq;
        switch (ntile) {
        case 1: lstm_batch_mma<1>(sa, wc, acc); break;
        case 2: lstm_batch_mma<2>(sa, wc, acc); break;
        case 3: lstm_batch_mma<3>(sa, wc, acc); break;
        default: lstm_batch_mma<4>(sa, wc, acc); break;
        }
    }
    // C/D map of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + gate) * LSTM_RB + r * 16 + lq * 4 + reg) * 17 + l15] = acc[r][reg];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + 512 * e, erow = idx >> 4, eu = idx & 15, slot = row0 + erow;
        if (erow < ntile * 16 && slot < n_active) {
            float pre[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pre[k] = (sg[((0 * 4 + k) * LSTM_RB + erow) * 17 + eu] + sg[((1 * 4 + k) * LSTM_RB + erow) * 17 + eu]) + gpre[e][k];
            const float a = am_tanh_rec(pre[0]);
            const float i = am_sigmoid_rec(pre[1]);
            const float f = am_sigmoid_rec(pre[2]);
            const float o = am_sigmoid_rec(pre[3]);
            const float ai = a * i;
            const float fc = f * cprev[e];
            const float c = ai + fc;
            const size_t cidx = (size_t)slot * U_pad + ut * 16 + eu;
            Ls.Cst[cidx] = c;
            Ls.Hout[cidx] = o * am_tanh_rec(c);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same step with a DEEP register prefetch of the weights ("wreg"): used when every layer's K is a multiple of 512
// (4 batches of 128 k; 3x512: 512 for the hoisted layer, 1024 for the others).
// The streaming kernel above prefetches one 128-k batch ahead, and every batch costs a full memory round trip
// (~1.6-2 us: the weight set is re-fetched from beyond L2 every launch) against 0.2-0.9 us of MFMA work per block.
// Here a wave keeps the weight fragments of FOUR batches in flight (4 x 4 x 16 B per lane = 64 VGPRs), refilled one batch
// per iteration, and the shared activation rows two batches ahead through a double-buffered LDS tile (one barrier per
// batch); loads are issued in consumption order (A0 W0 A1 W1 W2 W3 | A2 W4 | A3 W5 ...) because a wave's loads return
// in order: with all weights issued up front, the activation tile of batch 2 queued behind the whole weight stream
// (measured: 30 us per launch at 192 rows, fully additive: 13 us A path + 9 us weight stream + 9 us MFMA).
// One block per CU (registers), 1..6 row tiles per block (at 6 the weight window is 2 batches).  Same summation order as the other two kernels.
// ---------------------------------------------------------------------------------------------------
constexpr int LSTM_WREG_ROWS = 96;        // row capacity of the LDS tiles (6 row tiles)
constexpr size_t LSTM_WREG_LDS = 2 * LSTM_WREG_ROWS * LSTM_LDA * sizeof(float);   // 101,376 B; the gate exchange buffer aliases it after the K loop
constexpr int LSTM_WREG_PAIR_ROWS = 32;   // two-blocks-per-CU form: 2 row tiles
constexpr size_t LSTM_WREG_PAIR_LDS = 2 * LSTM_WREG_PAIR_ROWS * LSTM_LDA * sizeof(float);   // 33,792 B

template <int NT, int D>
__device__ __forceinline__ void lstm_wreg_body(const LstmLayerStep& Ls, int U_pad, int ut, int row0, float* smem, int tile_rows)
{
    constexpr int ROWS = NT * 16;
    constexpr int EP = (ROWS * 16 + 511) / 512;       // epilogue items per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ldg = 4 * U_pad;
    const int first = Ls.first, n_active = Ls.n_active;
    float* sA0 = smem;
    float* sA1 = smem + tile_rows * LSTM_LDA;

    float gpre[EP][4], cprev[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e, erow = idx >> 4, eu = idx & 15, slot = row0 + erow;
        const bool live = erow < ROWS && slot < n_active;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            gpre[e][k] = !live ? 0.0f : (Ls.G ? Ls.G[(size_t)slot * ldg + ut * 64 + k * 16 + eu] : Ls.bias[ut * 64 + k * 16 + eu]);
        cprev[e] = (!first && live) ? Ls.Cst[(size_t)slot * U_pad + ut * 16 + eu] : 0.0f;
    }

    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int nb0 = Ls.Hin ? Ls.Kin_pad / LSTM_KB : 0;     // batches of the upward segment, the rest is lateral
    const int nb = nb0 + (first ? 0 : U_pad / LSTM_KB);    // a multiple of 4 (launcher)
    if (nb > 0) {
        // A chunk e of a batch = row tile e: thread -> (row = idx / 32, 16-B column idx % 32)
        f32x4 av[2][NT], wv[D][LSTM_KS];   // D = weight batches in flight
        auto load_a = [&](int b, f32x4 (&dst)[NT]) {
            b = b < nb ? b : nb - 1;       // past the end: re-fetch the last batch (never consumed), keeps the loop branch-free
            const float* ap; int K;
            if (b < nb0) { K = Ls.Kin_pad; ap = Ls.Hin + (size_t)row0 * K + b * LSTM_KB; }
            else { K = U_pad; ap = Ls.Hprev + (size_t)row0 * K + (b - nb0) * LSTM_KB; }
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);
                dst[e] = *(const f32x4*)(ap + (size_t)r * K + 4 * c4);
            }
        };
        auto load_w = [&](int b, f32x4 (&dst)[LSTM_KS]) {
            b = b < nb ? b : nb - 1;
            const float* wp; int K;
            if (b < nb0) { K = Ls.Kin_pad; wp = Ls.Wup + b * LSTM_KB; }
            else { K = U_pad; wp = Ls.Wlat + (b - nb0) * LSTM_KB; }
            wp += (size_t)(ut * 64 + gate * 16 + l15) * K + kh * (LSTM_KB / 2) + 4 * lq;
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) dst[s] = *(const f32x4*)(wp + 16 * s);
        };
        load_a(0, av[0]); load_w(0, wv[0]);
        load_a(1, av[1]); load_w(1, wv[1]);
        if constexpr (D == 4) { load_w(2, wv[2]); load_w(3, wv[3]); }
        for (int b0 = 0; b0 < nb; b0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + i;
                float* sA = (i & 1) ? sA1 : sA0;
#pragma unroll
                for (int e = 0; e < NT; ++e) {
                    const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);
                    *(f32x4*)(sA + r * LSTM_LDA + 4 * c4) = av[i & 1][e];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // tile b is complete; everyone left tile b-2 (same buffer) before its barrier b-1
                load_a(b + 2, av[i & 1]);
                const float* sa = sA + l15 * LSTM_LDA + kh * (LSTM_KB / 2) + 4 * lq;
                f32x4 pb[NT];
#pragma unroll
                for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s) {
                    f32x4 a[NT];
#pragma unroll
                    for (int r = 0; r < NT; ++r) a[r] = *(const f32x4*)(sa + r * 16 * LSTM_LDA + 16 * s);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < NT; ++r) pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], wv[i % D][s][j], pb[r], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < NT; ++r) acc[r] += pb[r];
                load_w(b + D, wv[i % D]);       // refill this slot: D batches ahead
            }
        }
        __syncthreads();                    // all fragment reads of the A tiles are done: reuse the space for the gate exchange
    }
    float* sg = smem;                       // [kh][gate][ROWS][17]
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + gate) * ROWS + r * 16 + lq * 4 + reg) * 17 + l15] = acc[r][reg];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e, erow = idx >> 4, eu = idx & 15, slot = row0 + erow;
        if (erow < ROWS && slot < n_active) {
            float pre[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pre[k] = (sg[((0 * 4 + k) * ROWS + erow) * 17 + eu] + sg[((1 * 4 + k) * ROWS + erow) * 17 + eu]) + gpre[e][k];
            const float a = am_tanh_rec(pre[0]);
            const float i = am_sigmoid_rec(pre[1]);
            const float f = am_sigmoid_rec(pre[2]);
            const float o = am_sigmoid_rec(pre[3]);
            const float ai = a * i;
            const float fc = f * cprev[e];
            const float c = ai + fc;
            const size_t cidx = (size_t)slot * U_pad + ut * 16 + eu;
            Ls.Cst[cidx] = c;
            Ls.Hout[cidx] = o * am_tanh_rec(c);
        }
    }
}

// PAIR = false: one block per CU (up to 256 VGPRs), 1..6 row tiles, 4 weight batches in flight (2 at 6 tiles: 2.6 us of
// MFMA per batch already covers a round trip, 4 would spill).  PAIR = true: two co-resident blocks per CU (128 VGPRs),
// 1..2 row tiles, 2 batches in flight: one block's barrier bubbles are filled by the other's MFMAs.
template <bool PAIR>
__global__ __launch_bounds__(512, PAIR ? 4 : 2) void lstm_wreg_kernel(const LstmWavefrontArgs args)
{
    extern __shared__ __attribute__((aligned(16))) float wreg_smem[];
    int li = 0;
#pragma unroll
    for (int l = 1; l < LSTM_MAX_LAYERS; ++l) if (l < args.n && (int)blockIdx.x >= args.L[l].block0) li = l;
    const LstmLayerStep& Ls = args.L[li];
    const int U_pad = args.U_pad, ut_groups = args.ut_groups;
    const int bid = blockIdx.x - Ls.block0;
    const int ut = (bid & 7) + 8 * ((bid >> 3) % ut_groups);
    const int rt = (bid >> 3) / ut_groups;
    if (ut * 16 >= U_pad) return;
    const int row0 = rt * Ls.tpb * 16;
    const int tiles_left = (Ls.n_active - row0 + 15) / 16;
    const int ntile = tiles_left < Ls.tpb ? tiles_left : Ls.tpb;
    if (PAIR) {
        switch (ntile) {   // uniform per block
        case 1: lstm_wreg_body<1, 2>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_PAIR_ROWS); break;
        case 2: lstm_wreg_body<2, 2>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_PAIR_ROWS); break;
        default: break;
        }
    } else {
        switch (ntile) {
        case 1: lstm_wreg_body<1, 4>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        case 2: lstm_wreg_body<2, 4>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        case 3: lstm_wreg_body<3, 4>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        case 4: lstm_wreg_body<4, 4>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        case 5: lstm_wreg_body<5, 4>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        case 6: lstm_wreg_body<6, 2>(Ls, U_pad, ut, row0, wreg_smem, LSTM_WREG_ROWS); break;
        default: break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same step for FEW active utterances (<= 16*R rows per layer; half of the launches of a TIMIT-like file run with
// <= 16 rows).  The wide kernel above walks K in 8 serial batches, one memory round trip each (1.6 us), on only
// 32 blocks per layer; with few rows the step is pure latency.  Here K is split over the 8 waves of a block instead and
// every wave fetches its whole K slice -- weight fragments AND the activation rows, straight into registers, no LDS
// staging -- in ONE round trip; a block owns 4 units x 4 gates (one 16-column MFMA tile), so a layer is U/4 blocks and
// the whole chip works on the step.  The 8 partial tiles are summed through LDS in a fixed order.
//   lane (l15, lq): column l15 = gate l15>>2 of unit 4q + (l15&3); K permutation as above (4 consecutive k per lane).
// R = 16-row tiles per block, CS = 16-k sub-steps fetched per round trip.
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(512) void lstm_thin_kernel(const LstmWavefrontArgs args)
{
    extern __shared__ float sp[];                           // [chain][r][16 rows][17]
    const LstmLayerStep& Ls = args.L[blockIdx.y];
    const int U_pad = args.U_pad;
    const int q = blockIdx.x;
    const int n_active = Ls.n_active;
    const int tiles = (n_active + 15) >> 4;
    const int ntile = tiles < R ? tiles : R;
    if (ntile <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int ldg = 4 * U_pad;
    const int gcol = (q >> 2) * 64 + 4 * (q & 3);          // + 16*gate + unit-in-quad
    const int wrow = gcol + 16 * (l15 >> 2) + (l15 & 3);
    const int first = Ls.first;

    // epilogue operands first (threads tid < 64*R own one (row, unit) each)
    const int er = tid >> 6, erow = (tid & 63) >> 2, euu = tid & 3, eslot = er * 16 + erow;
    const bool elive = tid < 64 * R && er < ntile && eslot < n_active;
    float gpre[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (elive) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            gpre[k] = Ls.G ? Ls.G[(size_t)eslot * ldg + gcol + 16 * k + euu] : Ls.bias[gcol + 16 * k + euu];
        if (!first) cprev = Ls.Cst[(size_t)eslot * U_pad + 4 * q + euu];
    }

    const int Kin = Ls.Hin ? Ls.Kin_pad : 0;
    const int Ktot = Kin + (first ? 0 : U_pad);
    const int nc = Ktot >> 6;                               // 64-k chains (Ktot is a multiple of 128)
    const int m = (nc + 7) >> 3;                            // chains per wave: wave w owns [w*m, (w+1)*m)
    for (int c0 = 0; c0 < m; c0 += 2) {                     // two chains = 8 sub-steps of 16 k per round trip
        const int ch0 = w * m + c0;
        const int nch = (ch0 >= nc) ? 0 : ((c0 + 1 < m && ch0 + 1 < nc) ? 2 : 1);
        f32x4 wv[8], av[R][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if ((s >> 2) < nch) {
                const int k0 = ch0 * 64 + s * 16;           // consecutive chains are consecutive in k
                const float* ap; const float* wp; int ld;
                if (k0 < Kin) { ld = Ls.Kin_pad; ap = Ls.Hin + k0; wp = Ls.Wup + (size_t)wrow * ld + k0; }
                else { ld = U_pad; ap = Ls.Hprev + (k0 - Kin); wp = Ls.Wlat + (size_t)wrow * ld + (k0 - Kin); }
                wv[s] = *(const f32x4*)(wp + 4 * lq);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < ntile) av[r][s] = *(const f32x4*)(ap + (size_t)(r * 16 + l15) * ld + 4 * lq);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h < nch) {
                f32x4 acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 4 * h; s < 4 * h + 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (r < ntile) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][s][j], wv[s][j], acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        sp[(((ch0 + h) * R + r) * 16 + lq * 4 + reg) * 17 + l15] = acc[r][reg];
            }
        }
    }
    __syncthreads();
    if (elive) {
        float pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the wide kernel's order: even chains folded in k order, odd chains likewise, (even + odd) + upward term
            float ev = 0.0f, od = 0.0f;
            for (int c = 0; c < nc; c += 2) {
                ev += sp[((c * R + er) * 16 + erow) * 17 + 4 * k + euu];
                od += sp[(((c + 1) * R + er) * 16 + erow) * 17 + 4 * k + euu];
            }
            pre[k] = (ev + od) + gpre[k];
        }
        const float a = am_tanh_rec(pre[0]);
        const float i = am_sigmoid_rec(pre[1]);
        const float f = am_sigmoid_rec(pre[2]);
        const float o = am_sigmoid_rec(pre[3]);
        const float ai = a * i;
        const float fc = f * cprev;
        const float cn = ai + fc;
        const size_t cidx = (size_t)eslot * U_pad + 4 * q + euu;
        Ls.Cst[cidx] = cn;
        Ls.Hout[cidx] = o * am_tanh_rec(cn);
    }
}

// ---------------------------------------------------------------------------------------------------
// The whole stack, the whole time axis, ONE launch ("persistent" form; replaces the T + L - 1 dependent launches above when
// the stack fits: U_pad 256 or 512, at most 256 workgroups).
//
// Why: a dependent launch costs ~2 us of boundary plus a cold start -- L2 contents do not survive it, so every step
// re-fetched the whole weight set (22-39 MB per launch) and paid a memory round trip before its first MFMA; at 192 rows a
// step took 26.7 us against 13.7 us of MFMA time.  Here nothing is re-fetched: a workgroup = (layer, row group, 16-unit
// tile) loads its 64 gate columns x K of weights ONCE into registers (wave = gate x K half: 16 columns x K/2 = K/8 VGPRs,
// 128 at K = 1024) and then walks the time axis.  Per step only the activation rows move: h through HBM/L2 with
// write-through (sc1) stores and L1-bypassing (sc1) loads, handed over by per-group arrival counters (one agent-scope
// atomic add per workgroup and step, one lane polling; MI355X_MICROARCH.md, inter-workgroup visibility, first row of the
// measured hand-off table).  The execution is dataflow, not lock-step:
//   step t of (layer l, group g) = [wait: layer l-1 has finished step t]   upward product   h_{l-1}[t] . Wup^T
//                                  [wait: layer l   has finished step t-1] lateral product  h_l[t-1]   . Wlat^T
//                                  gates, c, h  ->  sc1 stores  ->  vmcnt(0), barrier, counter += 1
// so the upward half of step t runs in the shadow of the other workgroups' step t-1, lower layers run ahead freely (every
// layer has its own time-major H plane, nothing is overwritten), and a step's cost approaches its MFMA time.
// Row groups are re-cut EVERY step: with `tiles` 16-row tiles active, group g of NG takes tiles [g q, g q + q), q =
// ceil(tiles / NG) -- all workgroups of a layer stay busy while utterances end.  The cell state therefore travels through
// memory like h, and like h through a TIME-MAJOR plane of its own (row step_off[t] + slot): an address is written once per
// launch, so no XCD's L2 can hold an older version of a line a slot's next owner reads (a per-slot array rewritten in place
// would rely on sc1 stores dropping the writer's line and on where the workgroups land; round 3).
// Rows past n_active[t] are never loaded: in the packed layout they alias rows of step t+1 that another
// XCD may not have written yet, and a line fetched early would stay stale in this XCD's L2.
// Same summation order as the per-step kernels: per K half, 64-k chains from zero folded in k order (upward batches, then
// lateral), (half 0 + half 1) + G/bias -- a frame's posteriors do not depend on the form that computed them.
// Every wait is bounded (2 s of the 100 MHz counter): a timeout raises `abort_flag`, every workgroup leaves, the host
// reports an error instead of hanging the GPU.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 ld_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */));
}

__device__ __forceinline__ void st_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 16 /* sc1 */);
}

// Wave 0, lane j < n polls counter first + j until it reaches cum[(first + j) * T + tcol]; then the workgroup's barrier.
// Returns false (all threads) when the pass has been aborted.
__device__ __forceinline__ bool persist_wait(const LstmPersistArgs& a, int first, int n, int tcol, int* s_flag)
{
    if (threadIdx.x < 64) {
        bool ok = true;
        if ((int)threadIdx.x < n) {
            const unsigned target = a.cum[(size_t)(first + threadIdx.x) * a.T + tcol];
            const unsigned* c = a.counters + first + threadIdx.x;
            if (ld_agent_u32(c) < target) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                int it = 0;
                while (ld_agent_u32(c) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++it & 63) == 0 && (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                            __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)LSTM_P_TIMEOUT)) { ok = false; break; }
                }
            }
        }
        if (!ok) { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_flag = 1; }
    }
    __syncthreads();
    return *s_flag == 0;
}

// A pipeline of 128-k batches of one step's products: NBU batches of the upward product (rows of `up` at row_up), then -- if
// LAT -- NBL batches of the lateral product (rows of `lat` at row_lat), against the resident weight fragments.
// Tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, sc1: no VGPR staging, so up to three further tiles are in
// flight while one is multiplied -- the timeline showed 9.7 us per 4-batch product for 6.85 us of MFMA time with one or two
// tiles in flight in registers: the step is bound by the latency of the hand-off loads, not by the matrix pipe).
// LDS image of a tile: [rows][128 floats], unpadded (a DMA instruction writes 64 x 16 B contiguously = two rows); the 16-B
// chunk index is XOR-ed with the row's low four bits ON THE SOURCE ADDRESS, so the ds_read_b128 of (row l15, chunk 4s + lq) is
// conflict-free in every 16-lane service group.  Ring of NBUF tile buffers; tile i + NBUF - 1 is requested right after the
// barrier that ends tile i - 1.  Only the first rows_valid rows are read: rows beyond alias rows of step t + 1 that another XCD
// may not have written yet (a line fetched early would stay stale in this XCD's L2); their tile rows repeat the last valid row
// and feed only output rows nobody stores.  `lat_wait` runs once, before the first lateral tile is requested; `mid` once, two
// batches before the end (the epilogue's operands are requested there).
// Summation order: the per-step kernels' (64-k chains from zero, folded per K half in batch order).
__device__ __forceinline__ void wait_vmcnt(int k)
{
    switch (k) {   // folds to one instruction: k is a constant after unrolling
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

constexpr int PERSIST_TILE_LD = LSTM_KB;                              // floats per tile row (no padding: swizzled)
constexpr size_t PERSIST_RING_BYTES = 3 * 96 * PERSIST_TILE_LD * sizeof(float);   // 147,456 B: 3 x 96 rows or 4 x 64 rows

template <int NBU, int NBL, bool LAT, int NT, class Wait, class Mid>
__device__ __forceinline__ bool persist_products(const float* up, unsigned row_up, const float* lat, unsigned row_lat, int rows_valid, int U_pad,
                                                 const f32x4 (&wu)[NBU ? NBU : 1][LSTM_KS], const f32x4 (&wl)[NBL][LSTM_KS], f32x4 (&acc)[NT],
                                                 float* ring, int tid, int l15, int lq, int kh, Wait&& lat_wait, Mid&& mid)
{
    constexpr int NBT = NBU + (LAT ? NBL : 0);
    constexpr int BUF_ROWS = NT <= 4 ? 64 : 96;
    constexpr int NBUF = NT <= 4 ? 4 : 3;
    constexpr int BUF_FLOATS = BUF_ROWS * PERSIST_TILE_LD;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    bool ok = true;
    // DMA instruction e of this wave covers tile rows 2 (wave + 8 e) and + 1: lane -> (row, stored chunk); source chunk = stored ^ (row & 15).
    // Rows past the valid ones re-read the last valid row (never a row that may not have been written yet); no lane is masked, so
    // every wave issues exactly NT DMA instructions per tile -- the counted vmcnt waits below rely on that.
    // Addressing: uniform 64-bit base (source plane + first row + batch) + one 32-bit per-lane offset per e, shared by both sources.
    const int drow = lane >> 5, dchunk = lane & 31;
    unsigned doff[NT];
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int row = 2 * (wave + 8 * e) + drow;
        const int srow = row < rows_valid ? row : rows_valid - 1;
        doff[e] = (unsigned)srow * (unsigned)U_pad + 4u * (unsigned)(dchunk ^ (row & 15));
    }
    const float* up0 = up + (size_t)row_up * U_pad;
    const float* lat0 = lat + (size_t)row_lat * U_pad;
    auto issue = [&](int b) {
        const float* src = (b < NBU ? up0 + b * LSTM_KB : lat0 + (b - NBU) * LSTM_KB);
        float* buf = ring + (b % NBUF) * BUF_FLOATS;
#pragma unroll
        for (int e = 0; e < NT; ++e)
            __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(src + doff[e]), AM_LDS_PTR(buf + (wave + 8 * e) * 256), 16, 0, 16 /* sc1 */);
    };
    if (LAT && NBU < NBUF - 1 && NBU < NBT) ok = lat_wait();          // a lateral tile is among the first requests
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) if (b < NBT) issue(b);
    // fragment addressing: row r*16 + l15, chunk (kh*16 + 4 s + lq) ^ l15
    const float* fbase = ring + l15 * PERSIST_TILE_LD + kh * (LSTM_KB / 2);
    int xs[LSTM_KS];
#pragma unroll
    for (int s = 0; s < LSTM_KS; ++s) xs[s] = 4 * (((4 * s + lq) ^ l15) & 15);
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        const int issued = (b + NBUF - 1 < NBT) ? b + NBUF - 1 : NBT;     // tiles requested so far
        wait_vmcnt(NT * (issued - (b + 1)));                               // tile b has landed (this wave's part)
        __builtin_amdgcn_s_barrier();                                      // ... everybody's; and everyone has left tile b-1
        // The request of tile b + NBUF - 1 goes out BETWEEN this batch's MFMAs, and at a different sub-step in the two waves that share a SIMD
        // (K half 0: behind sub-step 0, K half 1: behind sub-step 2): every wave of the workgroup leaves the barrier at the same moment, and
        // issued right there the NT LDS-DMA instructions of both waves (~75 cycles each) find the matrix pipe idle; staggered, one wave's
        // issue runs beside the other's MFMAs.
        const bool due = b + NBUF - 1 < NBT;
        if (due && LAT && NBU >= NBUF - 1 && b + NBUF - 1 == NBU) ok = lat_wait() && ok;
        if (b + 2 == NBT || (NBT == 1 && b == 0)) mid();
        const float* sa = fbase + (b % NBUF) * BUF_FLOATS;
        f32x4 pb[NT];
#pragma unroll
        for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (NT <= 2) {   // few rows: the step is latency-bound, fetch the whole batch's fragments before the first MFMA
            f32x4 af[LSTM_KS][NT];
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s)
#pragma unroll
                for (int r = 0; r < NT; ++r) af[s][r] = *(const f32x4*)(sa + r * 16 * PERSIST_TILE_LD + xs[s]);
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < NT; ++r)
                        pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][r][j], (b < NBU ? wu[b < NBU ? b : 0][s][j] : wl[b < NBU ? 0 : b - NBU][s][j]), pb[r], 0, 0, 0);
            if (due) issue(b + NBUF - 1);   // (few rows: the batch is short, the request goes out behind its MFMAs' issue)
        } else {         // (double-buffering the fragments across sub-steps costs 16 VGPRs the weights need, and measured no gain)
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) {
                f32x4 af[NT];
#pragma unroll
                for (int r = 0; r < NT; ++r) af[r] = *(const f32x4*)(sa + r * 16 * PERSIST_TILE_LD + xs[s]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < NT; ++r)
                        pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j], (b < NBU ? wu[b < NBU ? b : 0][s][j] : wl[b < NBU ? 0 : b - NBU][s][j]), pb[r], 0, 0, 0);
                if (due && s == (kh ? 2 : 0)) issue(b + NBUF - 1);
            }
        }
#pragma unroll
        for (int r = 0; r < NT; ++r) acc[r] += pb[r];
    }
    if (NBT > 0) __syncthreads();   // every fragment read is done: the ring may be rewritten (gate exchange, next pipeline)
    return ok;
}

// diagnostic build of the timeline (AM_LSTM_STAMPS=1): the ut == 0 workgroup of every group stamps the 100 MHz counter at
// points of every step into a buffer nothing else reads
__device__ __forceinline__ void persist_stamp(const LstmPersistArgs& a, int gidx, int ut, int t, int k)
{
    if (a.stamps && ut == 0 && threadIdx.x == 0) a.stamps[((size_t)gidx * a.T + t) * 8 + k] = __builtin_amdgcn_s_memrealtime();
}

constexpr int PERSIST_SGLD = 20;   // gate-exchange row stride: 16 units + 16 B (f32x4 reads stay aligned, b32 writes conflict-free)

template <int NBU, int NBL, int NT>
__device__ __forceinline__ bool persist_step(const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int gidx, int ut, int t, int n_act, int row0,
                                             unsigned off_t, unsigned off_p, const f32x4 (&wu)[NBU ? NBU : 1][LSTM_KS], const f32x4 (&wl)[NBL][LSTM_KS],
                                             float* smem, int tile_rows, int* s_flag)
{
    constexpr int ROWS = NT * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int U_pad = a.U_pad, ldg = 4 * U_pad;
    const int rows_valid = n_act - row0;                       // > 0
    const int slot0 = a.slot_base + row0;                       // first utterance slot of this group's rows
    const unsigned r_t = off_t + (unsigned)slot0;               // its packed row at step t (off_t = step_off[t], fetched a step ahead by the caller)
    const unsigned r_p = off_p + (unsigned)slot0;               // ... and at step t - 1 (h and c of the previous step; unused at t == 0)
    persist_stamp(a, gidx, ut, t, 0);
    if (NBU && li > 0) {   // upward product: needs step t of the layer below (which runs ahead: normally no wait at all; li == 0 with an
                           // upward product: the segment's first layer, its input plane was completed by an earlier launch)
        if (!persist_wait(a, a.L[li - 1].group0, a.L[li - 1].n_groups, t, s_flag)) return false;
    }
    persist_stamp(a, gidx, ut, t, 1);
    // epilogue of thread (row, unit quad): the four gates' upward terms (G rows of layer 0, the bias otherwise), previous cell state
    const int erow = tid >> 2, quad = tid & 3;
    const bool live = erow < ROWS && erow < rows_valid;
    f32x4 gq[4], cq = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.Cst, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.H, 0, 0x7fffffff, 0x00020000);
    // layer 0 (64 VGPRs of weights) requests its G rows two batches before the end of the product; the upper layers (128 VGPRs
    // of weights, no register to spare during the products) fetch their bias -- cache-resident -- behind the gate exchange
    auto fetch_g = [&] {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gq[k] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (live) gq[k] = NBU ? *(const f32x4*)(Ly.bias + ut * 64 + k * 16 + 4 * quad)
                                  : *(const f32x4*)(Ly.G + (size_t)(r_t + erow) * ldg + ut * 64 + k * 16 + 4 * quad);
        }
    };
    auto fetch_epilogue = [&] {
        if (!NBU) fetch_g();
        if (live && t > 0) cq = ld_sc1_b128(crs, ((r_p + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u);
    };
    auto lat_wait = [&] { return persist_wait(a, Ly.group0, Ly.n_groups, t - 1, s_flag); };   // step t-1 of every group of this layer
    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    bool ok = true;
    // layer 0: one pipeline (lateral only).  Upper layers: two pipelines -- a single one over both products needs ~20 VGPRs more
    // than the 256 there are (both weight sets stay live through the wait) and the spills land in the MFMA loop: measured 21.1 us
    // per 192-row step against 21.5 us for two pipelines, not worth the scratch traffic
    if (NBU) {
        if (t == 0) ok = persist_products<NBU, NBL, false, NT>(Ly.Hin, r_t, Ly.H, 0u, rows_valid, U_pad, wu, wl, acc, smem, tid, l15, lq, kh, lat_wait, fetch_epilogue);
        else ok = persist_products<NBU, NBL, false, NT>(Ly.Hin, r_t, Ly.H, 0u, rows_valid, U_pad, wu, wl, acc, smem, tid, l15, lq, kh, lat_wait, [] {});
        persist_stamp(a, gidx, ut, t, 2);
    }
    f32x4 wnone[1][LSTM_KS] = {};   // stands in for the absent upward weights of a lateral-only pipeline (never read)
    if (t > 0) ok = persist_products<0, NBL, true, NT>(Ly.H, 0u, Ly.H, r_p, rows_valid, U_pad, wnone, wl, acc, smem, tid, l15, lq, kh, lat_wait, fetch_epilogue) && ok;
    else if (!NBU) fetch_epilogue();
    if (!ok) return false;
    persist_stamp(a, gidx, ut, t, 4);
    float* sg = smem;   // [kh][gate][ROWS][PERSIST_SGLD], aliases the A tiles (the products end with a barrier)
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + gate) * ROWS + r * 16 + lq * 4 + reg) * PERSIST_SGLD + l15] = acc[r][reg];
    if (NBU) fetch_g();
    __syncthreads();
    if (live) {
        f32x4 pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            pre[k] = (*(const f32x4*)(sg + ((0 * 4 + k) * ROWS + erow) * PERSIST_SGLD + 4 * quad) + *(const f32x4*)(sg + ((1 * 4 + k) * ROWS + erow) * PERSIST_SGLD + 4 * quad)) + gq[k];
        f32x4 cn, hn;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float ga = am_tanh_rec(pre[0][u]);
            const float gi = am_sigmoid_rec(pre[1][u]);
            const float gf = am_sigmoid_rec(pre[2][u]);
            const float go = am_sigmoid_rec(pre[3][u]);
            const float ai = ga * gi;
            const float fc = gf * cq[u];
            const float c = ai + fc;
            cn[u] = c;
            hn[u] = go * am_tanh_rec(c);
        }
        st_sc1_b128(crs, ((r_t + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u, cn);
        st_sc1_b128(hrs, ((r_t + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u, hn);
    }
    persist_stamp(a, gidx, ut, t, 5);
    // publish: every storing wave drains its stores, the workgroup meets, ONE lane signals for all of them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.counters + gidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    persist_stamp(a, gidx, ut, t, 6);
    return true;
}

// The step of an upper layer of a 512-unit stack (four 128-k batches per product) on 256-k tiles: ONE pipeline over both products, two tile
// buffers of 64 rows x 1 KB, and every memory round trip of the step's chain requested before it is needed.
// The timeline of the 128-k form showed a FIXED ~7.4 us per step on top of the MFMA time, whatever the row count (16 rows: 7.4 us a step; 192
// rows: 21 us against 13.7 us of matrix work): the counter poll at the top, the cold start of each product's tile stream, the lateral poll, the
// gate exchange and the publish -- all in series.  Here:
//   T0 = h_below[t][0:256)  T1 = h_below[t][256:512)  T2 = h[t-1][0:256)  T3 = h[t-1][256:512);  tile i lives in buffer (i + 1) & 1
//   top   : [T0 already travelling: requested by step t-1, see below | else upward wait, request T0]  lateral counters peeked  request T1
//   b = 0 : T0 landed, barrier, batches U0 U1
//   b = 1 : T1 landed, lateral wait (skips the poll when the peek already saw the arrivals; ends in the barrier), cell state requested,
//           step t+1's upward counters peeked, T2 requested between the MFMAs, batches U2 U3
//   b = 2 : T2 landed, barrier, T3 requested between the MFMAs, batches L0 L1
//   b = 3 : T3 landed, barrier, if the peek saw step t+1 of the layer below: T0 OF STEP t+1 requested (buffer 1 is free, the gate exchange
//           uses buffer 0), batches L2 L3
//   barrier, gates, stores, publish
// so a step that finds its upward input ready (the normal case: lower layers run ahead) starts with its first tile in LDS and never polls.
// A DMA instruction moves one whole tile row (64 lanes x 16 B): the clamp to the last valid row is per instruction and the per-lane part of
// the address is the swizzled chunk alone.  Summation order unchanged: 64-k chains from zero per K half, folded in batch order U0..U3, L0..L3.
struct PersistNext { int nt, row0, n_act; unsigned off; };   // a group's share of a step: tiles (<= 0: none), first row, active rows, step_off[t]

__device__ __forceinline__ bool persist_wait_seen(const LstmPersistArgs& a, int first, int n, unsigned seen, unsigned target, int* s_flag)
{
    if (threadIdx.x < 64) {
        bool ok = true;
        if ((int)threadIdx.x < n && seen < target) {
            const unsigned* c = a.counters + first + threadIdx.x;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int it = 0;
            while (ld_agent_u32(c) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++it & 63) == 0 && (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                        __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)LSTM_P_TIMEOUT)) { ok = false; break; }
            }
        }
        if (!ok) { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_flag = 1; }
    }
    __syncthreads();
    return *s_flag == 0;
}

template <int NT>
__device__ __forceinline__ bool persist_step_wide(const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int gidx, int ut, int t, int n_act, int row0,
                                                  unsigned off_t, unsigned off_p, const PersistNext nx, int& have,
                                                  const f32x4 (&wu)[4][LSTM_KS], const f32x4 (&wl)[4][LSTM_KS], float* smem, int* s_flag)
{
    constexpr int ROWS = NT * 16;
    constexpr int TLD = 2 * LSTM_KB;                 // floats per tile row
    constexpr bool THIN = NT <= 2;                   // 32-row tiles: every tile of the step has a buffer of its own
    constexpr int Q = 32 * TLD;                      // a 32 KB quarter of the tile space, in floats
    // tile -> offset.  Thick: T0, T2 in the upper half, T1, T3 in the lower.  Thin: T0, T1 in the upper half's quarters, T2, T3 in the lower half's.
    // Either way a step's first tile(s) live in the upper half and the gate exchange in the lower, so the step before can request them.
    constexpr int OFF0 = 2 * Q, OFF1 = THIN ? 3 * Q : 0, OFF2 = THIN ? 0 : 2 * Q, OFF3 = THIN ? Q : 0;
    constexpr int NI = 2 * NT;                       // DMA instructions per wave and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int U_pad = a.U_pad;
    const int rows_valid = n_act - row0;                       // > 0
    const int slot0 = a.slot_base + row0;
    const unsigned r_t = off_t + (unsigned)slot0;
    const unsigned r_p = off_p + (unsigned)slot0;              // (t == 0: unused)
    const bool lat = t > 0;
    const bool ext = li == 0;                                   // a segment's first layer: its upward input is a plane an earlier launch has completed
    const LstmPersistLayer& Lb = a.L[ext ? 0 : li - 1];
    // tile row of instruction e: wave + 8 e, so row & 15 takes two values per wave
    // (buffer form of the LDS-DMA: the per-lane part of the address is one of two 32-bit offsets, everything else is scalar -- no 64-bit vector
    // address per instruction, which the register budget of this kernel does not have)
    const int loff0 = 16 * (lane ^ wave);             // tile row wave + 8 e: row & 15 is wave or wave + 8, the odd instructions' offset is loff0 ^ 128
    const __amdgpu_buffer_rsrc_t uprs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.Hin, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.H, 0, 0x7fffffff, 0x00020000);
    auto dma_row = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte0, int rows_v, float* buf, int e) {
        const int row = wave + 8 * e;
        const int srow = row < rows_v ? row : rows_v - 1;
        int lo = loff0;
        asm volatile("" : "+v"(lo));                   // (recomputed per instruction rather than held: the kernel has no register to spare)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, AM_LDS_PTR(buf + row * TLD), 16, (e & 1) ? (lo ^ 128) : lo, (int)(byte0 + (unsigned)srow * (unsigned)U_pad * 4u), 0, 16 /* sc1 */);
    };
    auto dma_tile = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte0, float* buf) {
#pragma unroll
        for (int e = 0; e < NI; ++e) dma_row(rs, byte0, rows_valid, buf, e);
    };
    const unsigned up0 = r_t * (unsigned)U_pad * 4u;     // byte offsets into the planes (< 2 GB: am_lstm_upload's bound on the persistent form)
    const unsigned lat0 = r_p * (unsigned)U_pad * 4u;
    constexpr unsigned TB = TLD * 4u;                    // bytes of a tile row: the second 256-k tile of a product starts there
    persist_stamp(a, gidx, ut, t, 0);
    if (have < 1) {
        if (!ext && !persist_wait(a, Lb.group0, Lb.n_groups, t, s_flag)) return false;
        dma_tile(uprs, up0, smem + OFF0);
    }
    const bool t1_here = have < 2;
    if (t1_here) dma_tile(uprs, up0 + TB, smem + OFF1);
    persist_stamp(a, gidx, ut, t, 1);
    // epilogue thread = (row, 4 units) with three or four row tiles (16-B loads and stores); with one or two, (row, ONE unit): 256 / 512 threads share
    // the gate arithmetic instead of 64 / 128 -- the gate phase is VALU time with the matrix pipe idle (0.65 -> 0.42 us, the publish behind it
    // 0.74 -> 0.35).  Two units per thread at three or four row tiles, all eight waves storing 8 B per lane, was measured too: gate phase unchanged,
    // the drain of the write-through stores 0.49 -> 1.2 us.
    // (all derived from an opaque copy of the thread index at each use: neither they nor the byte offsets built from them are held through the MFMA phases)
    auto tid_now = [&] { int x = tid; asm volatile("" : "+v"(x)); return x; };
    auto row_of = [&](int x) { return THIN ? x >> 4 : x >> 2; };
    auto is_live = [&](int er) { return er < ROWS && er < rows_valid; };
    f32x4 cq = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};   // (thin: cq[0] only)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.Cst, 0, 0x7fffffff, 0x00020000);
    auto load_c = [&] {   // the previous cell state of this thread's unit(s)
        const int x = tid_now(), er = row_of(x);
        if (!is_live(er)) return;
        if (THIN) cq[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, (int)(((r_p + er) * (unsigned)U_pad + (unsigned)(ut * 16 + (x & 15))) * 4u), 0, 16 /* sc1 */));
        else cq = ld_sc1_b128(crs, ((r_p + er) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * (x & 3))) * 4u);
    };
    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // fragment of (row r*16 + l15, sub-step s): float index l15*TLD + kh*64 + 4*(((4 s + lq) ^ l15) & 15) = la ^ (16 s) -- one register, not one per sub-step
    const unsigned la = (unsigned)(l15 * TLD + kh * (LSTM_KB / 2) + 4 * ((lq ^ l15) & 15));
    // Counters looked at ahead of their wait: wave 0 sends counter and target of group min(lane, n - 1) to LDS by DMA (no register is held through
    // the MFMA phases; the 16 KB above the tile space are free), and compares when the wait comes.
    unsigned* pk = (unsigned*)(smem + 4 * Q);   // [0] lateral counters [64] their targets [128] upward counters of step t + 1 [192] their targets
    auto peek = [&](int first, int n, int tcol, unsigned* dst) {
        if (wave == 0) {
            const int j = first + (lane < n ? lane : n - 1);
            __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(a.counters + j), AM_LDS_PTR(dst), 4, 0, 16 /* sc1 */);
            __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(a.cum + (size_t)j * a.T + tcol), AM_LDS_PTR(dst + 64), 4, 0, 0);
        }
    };
    auto peek_lateral = [&] { if (lat) peek(Ly.group0, Ly.n_groups, t - 1, pk); };
    const bool want_next = lat && nx.nt > 0 && (!THIN || nx.nt <= 2);   // (rows only ever decrease; the thin form's counted wait assumes a thin successor)
    int pre = 0;
    have = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b >= 2 && !lat) break;
        // tile b has landed (this wave's part)
        if (b == 0) { if (t1_here) wait_vmcnt(NI); else wait_vmcnt(0); }
        else if (b == 2 && THIN) wait_vmcnt(NI);                   // T3 was requested with T2 and still travels
        else if (b == 3 && THIN && pre != 0) { if (nx.nt == 1) wait_vmcnt(4); else wait_vmcnt(8); }   // ... and so do step t + 1's two tiles, requested at b == 2
        else wait_vmcnt(0);
        if (b == 1 && lat) {                                       // step t-1 of every group of this layer; the wait ends in the barrier
            unsigned seenL = 0, tgtL = 0;
            if (tid < 64) { seenL = pk[lane]; tgtL = pk[64 + lane]; }
            if (!persist_wait_seen(a, Ly.group0, Ly.n_groups, seenL, tgtL, s_flag)) return false;
            // few rows: registers to spare and short batches -- the previous cell state is requested here (many rows: under the last tile's MFMAs)
            if (THIN) load_c();
            if (want_next && !ext) peek(Lb.group0, Lb.n_groups, t + 1, pk + 128);   // (both before the tiles: b == 2's counted wait must cover them)
            if (THIN) { dma_tile(hrs, lat0, smem + OFF2); dma_tile(hrs, lat0 + TB, smem + OFF3); }   // few rows: the step is a chain of latencies, both lateral tiles leave at once
        } else {
            if (b == 2 && tid < 64) {                              // has the layer below finished step t + 1?
                const bool all = want_next && (ext || __builtin_amdgcn_ballot_w64(pk[128 + lane] < pk[192 + lane]) == 0ull);
                if (tid == 0) s_flag[1] = (want_next && all) ? 1 : 0;
            }
            __builtin_amdgcn_s_barrier();
            if (b == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pre = __builtin_amdgcn_readfirstlane(s_flag[1]); }
            if (!THIN && b == 3) load_c();
        }
        if (b == 2) persist_stamp(a, gidx, ut, t, 2);
        const bool due = !THIN && lat && (b == 1 || b == 2);       // the buffer of tile b + 1 is free: everybody is past tile b - 1
        const unsigned nsrc = b == 1 ? lat0 : lat0 + TB;
        float* nbuf = smem + (b == 1 ? OFF2 : OFF3);
        // the first tile of step t + 1 -- both tiles if that step is a thin one.  Its geometry is step t + 1's, not a constant of this instantiation.
        const bool due_next = b == (THIN ? 2 : 3) && pre != 0;     // (few rows: a tile earlier -- its buffers are free, and the publish must not wait for it)
        const unsigned up_next = (nx.off + (unsigned)(a.slot_base + nx.row0)) * (unsigned)U_pad * 4u;
        const int rows_next = nx.n_act - nx.row0;
        unsigned lb = la + (unsigned)(b == 0 ? OFF0 : b == 1 ? OFF1 : b == 2 ? OFF2 : OFF3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f32x4 pb[NT];
#pragma unroll
            for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const int wi = 2 * (b & 1) + q;
            if (NT <= 2) {
                f32x4 af[LSTM_KS][NT];
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s)
#pragma unroll
                    for (int r = 0; r < NT; ++r) af[s][r] = *(const f32x4*)(smem + (lb ^ (16u * s)) + r * 16 * TLD + q * LSTM_KB);
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < NT; ++r)
                            pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][r][j], (b < 2 ? wu[wi][s][j] : wl[wi][s][j]), pb[r], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s) {
                    f32x4 af[NT];
                    asm volatile("" : "+v"(lb));
                    const float* pa = smem + (lb ^ (16u * s)) + q * LSTM_KB;
#pragma unroll
                    for (int r = 0; r < NT; ++r) af[r] = *(const f32x4*)(pa + r * 16 * TLD);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < NT; ++r)
                            pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j], (b < 2 ? wu[wi][s][j] : wl[wi][s][j]), pb[r], 0, 0, 0);
                    // the next tile's request, spread over this wave's sub-steps of one 128-k batch -- K half 0 under the first, K half 1 under the second
                    if (due && q == kh) {
#pragma unroll
                        for (int e = 0; e < NI; ++e) if ((e & 3) == s) dma_row(hrs, nsrc, rows_valid, nbuf, e);
                    }
                }
            }
            if (b == 0 && q == 0) peek_lateral();
            if (due_next && (THIN ? q == 0 : q == kh)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < 2 * nx.nt) dma_row(uprs, up_next, rows_next, smem + 2 * Q, e);
                if (nx.nt <= 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < 2 * nx.nt) dma_row(uprs, up_next + TB, rows_next, smem + 3 * Q, e);
                }
            }
#pragma unroll
            for (int r = 0; r < NT; ++r) acc[r] += pb[r];
        }
        if (due_next) have = nx.nt <= 2 ? 2 : 1;
    }
    __syncthreads();
    persist_stamp(a, gidx, ut, t, 4);
    float* sg = smem;   // the lower half (at most 40 KB of it): the upper half may be receiving step t + 1's first tiles
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + gate) * ROWS + r * 16 + lq * 4 + reg) * PERSIST_SGLD + l15] = acc[r][reg];
    const int etid = tid_now(), erow = row_of(etid), quad = etid & 3, eu = etid & 15;
    const bool live = is_live(erow);
    if constexpr (THIN) {
        float g1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g1[k] = live ? Ly.bias[ut * 64 + k * 16 + eu] : 0.0f;
        __syncthreads();
        if (live) {
            float pre[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pre[k] = (sg[((0 * 4 + k) * ROWS + erow) * PERSIST_SGLD + eu] + sg[((1 * 4 + k) * ROWS + erow) * PERSIST_SGLD + eu]) + g1[k];
            const float ga = am_tanh_rec(pre[0]);
            const float gi = am_sigmoid_rec(pre[1]);
            const float gf = am_sigmoid_rec(pre[2]);
            const float go = am_sigmoid_rec(pre[3]);
            const float ai = ga * gi;
            const float fc = gf * cq[0];
            const float c = ai + fc;
            const float h = go * am_tanh_rec(c);
            const unsigned off = ((r_t + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + eu)) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), crs, (int)off, 0, 16 /* sc1 */);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), hrs, (int)off, 0, 16 /* sc1 */);
        }
    } else {
    f32x4 gq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gq[k] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (live) gq[k] = *(const f32x4*)(Ly.bias + ut * 64 + k * 16 + 4 * quad);
    }
    __syncthreads();
    if (live) {
        f32x4 pre4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            pre4[k] = (*(const f32x4*)(sg + ((0 * 4 + k) * ROWS + erow) * PERSIST_SGLD + 4 * quad) + *(const f32x4*)(sg + ((1 * 4 + k) * ROWS + erow) * PERSIST_SGLD + 4 * quad)) + gq[k];
        f32x4 cn, hn;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float ga = am_tanh_rec(pre4[0][u]);
            const float gi = am_sigmoid_rec(pre4[1][u]);
            const float gf = am_sigmoid_rec(pre4[2][u]);
            const float go = am_sigmoid_rec(pre4[3][u]);
            const float ai = ga * gi;
            const float fc = gf * cq[u];
            const float c = ai + fc;
            cn[u] = c;
            hn[u] = go * am_tanh_rec(c);
        }
        st_sc1_b128(crs, ((r_t + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u, cn);
        st_sc1_b128(hrs, ((r_t + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u, hn);
    }
    }
    persist_stamp(a, gidx, ut, t, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.counters + gidx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    persist_stamp(a, gidx, ut, t, 6);
    return true;
}

// upper layers of a 512-unit stack take the 256-k-tile form (AM_PERSIST_WIDE=0 at build time keeps the 128-k form for A/B timing)
#ifndef AM_PERSIST_WIDE
#define AM_PERSIST_WIDE 1
#endif
template <int NBU, int NBL, int NT>
__device__ __forceinline__ bool persist_step_any(const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int gidx, int ut, int t, int n_act, int row0,
                                                 unsigned off_t, unsigned off_p, const PersistNext nx, int& have,
                                                 const f32x4 (&wu)[NBU ? NBU : 1][LSTM_KS], const f32x4 (&wl)[NBL][LSTM_KS],
                                                 float* smem, int tile_rows, int* s_flag)
{
    if constexpr (AM_PERSIST_WIDE && NBU == 4 && NBL == 4 && NT <= 4) return persist_step_wide<NT>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_p, nx, have, wu, wl, smem, s_flag);
    else return persist_step<NBU, NBL, NT>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_p, wu, wl, smem, tile_rows, s_flag);
}

template <int NBU, int NBL, int NTMAX>
__device__ __forceinline__ void persist_body(const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int gidx, int ut, float* smem, int* s_flag)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int U_pad = a.U_pad;
    const int g = gidx - Ly.group0, NG = Ly.n_groups;
    // the resident weights: this wave's 16 gate columns x its half of every 128-k batch
    f32x4 wl[NBL][LSTM_KS], wu[NBU ? NBU : 1][LSTM_KS];
    {
        const size_t wrow = (size_t)(ut * 64 + gate * 16 + l15) * U_pad + kh * (LSTM_KB / 2) + 4 * lq;
#pragma unroll
        for (int b = 0; b < NBL; ++b)
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) wl[b][s] = *(const f32x4*)(Ly.Wlat + wrow + b * LSTM_KB + 16 * s);
        if (NBU) {
#pragma unroll
            for (int b = 0; b < (NBU ? NBU : 1); ++b)
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s) wu[b][s] = *(const f32x4*)(Ly.Wup + wrow + b * LSTM_KB + 16 * s);
        }
    }
    auto local_rows = [&](int n) { n -= a.slot_base; return n < 0 ? 0 : (n > a.slot_cap ? a.slot_cap : n); };   // this launch's share of the active slots
    // this launch's steps [t_begin, t_end): a launch that starts at t_begin > 0 finds h, c and the arrival counters as the launch for the steps
    // before it left them (its first waits are already satisfied)
    const int t_end = a.t_end > 0 ? a.t_end : a.T;
    // the group's share of a step with n active rows: q = ceil(tiles / NG) tiles from tile g q on (q <= NTMAX by the host plan)
    auto share = [&](int n_raw, unsigned off) {
        const int n = local_rows(n_raw);
        const int tiles = (n + 15) >> 4;
        const int q = (tiles + NG - 1) / NG;
        int nt = tiles - g * q;
        if (nt > q) nt = q;
        return PersistNext{tiles > 0 ? nt : 0, g * q * 16, n, off};
    };
    // n_active and step_off travel two steps ahead of their use (scalar loads; consumed the iteration after they were requested): the ~1 us between
    // a step's publish and the next step's first instruction that the timeline showed was these loads' latency
    int raw_n = a.n_active[a.t_begin];
    unsigned raw_off = (unsigned)a.step_off[a.t_begin];
    PersistNext nx = share(raw_n, raw_off);
    raw_n = a.t_begin + 1 < t_end ? a.n_active[a.t_begin + 1] : 0;
    raw_off = a.t_begin + 1 < t_end ? (unsigned)a.step_off[a.t_begin + 1] : 0u;
    unsigned off_p = a.t_begin > 0 ? (unsigned)a.step_off[a.t_begin - 1] : 0u;
    int have = 0;                                       // tiles of the step's upward product the step before has already requested (256-k form)
    for (int t = a.t_begin; t < t_end; ++t) {
        const PersistNext cur = nx;
        nx = share(raw_n, raw_off);
        raw_n = t + 2 < t_end ? a.n_active[t + 2] : 0;
        raw_off = t + 2 < t_end ? (unsigned)a.step_off[t + 2] : 0u;
        const int n_act = cur.n_act, nt = cur.nt, row0 = cur.row0;
        const unsigned off_t = cur.off, off_prev = off_p;
        off_p = off_t;
        if (n_act <= 0) break;
        if (nt <= 0) continue;                          // nothing for this group at this step (the host's arrival table knows)
        bool ok = true;
        switch (nt) {
        case 1: ok = persist_step_any<NBU, NBL, 1>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, nx, have, wu, wl, smem, NTMAX * 16, s_flag); break;
        case 2: ok = persist_step_any<NBU, NBL, 2>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, nx, have, wu, wl, smem, NTMAX * 16, s_flag); break;
        case 3: ok = persist_step_any<NBU, NBL, 3>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, nx, have, wu, wl, smem, NTMAX * 16, s_flag); break;
        case 4: ok = persist_step_any<NBU, NBL, 4>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, nx, have, wu, wl, smem, NTMAX * 16, s_flag); break;
        case 5: if (NTMAX >= 5) ok = persist_step<NBU, NBL, NTMAX >= 5 ? 5 : 1>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, wu, wl, smem, NTMAX * 16, s_flag); break;
        case 6: if (NTMAX >= 6) ok = persist_step<NBU, NBL, NTMAX >= 6 ? 6 : 1>(a, Ly, li, gidx, ut, t, n_act, row0, off_t, off_prev, wu, wl, smem, NTMAX * 16, s_flag); break;
        default: break;
        }
        if (!ok) return;
    }
}

constexpr int lstm_persist_ntmax(int nb, bool upward) { return (upward ? 2 * nb : nb) >= 8 ? 4 : 6; }   // by the register budget: K/8 VGPRs of weights
constexpr size_t LSTM_PERSIST_LDS = PERSIST_RING_BYTES;   // the tile ring; the gate exchange aliases it

template <int NB>
__global__ __launch_bounds__(512, 2) void lstm_persist_kernel(const LstmPersistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float persist_smem[];
    __shared__ int s_flag[2];   // [0] the pass has been aborted; [1] scratch of the 256-k step form
    if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    // role: blocks b and b + n_groups_total share a group's neighbour tiles; b % n_groups_total = group, so with 8 groups the
    // 32 workgroups of a group sit on one XCD under round-robin placement (speed only)
    const int gidx = blockIdx.x % a.n_groups_total, ut = blockIdx.x / a.n_groups_total;
    int li = 0;
#pragma unroll
    for (int l = 1; l < LSTM_MAX_LAYERS; ++l) if (l < a.n_layers && gidx >= a.L[l].group0) li = l;
    const LstmPersistLayer& Ly = a.L[li];
    if (li > 0 || a.ext_upward) persist_body<NB, NB, lstm_persist_ntmax(NB, true)>(a, Ly, li, gidx, ut, persist_smem, s_flag);
    else persist_body<0, NB, lstm_persist_ntmax(NB, false)>(a, Ly, li, gidx, ut, persist_smem, s_flag);
}



// The persistent kernel needs all its workgroups resident at once.  A build of it that spills even one register needs scratch
// memory, and the FIRST launch of a scratch-using kernel in a process does not get them all resident (measured with a
// 2-VGPR spill: the first pass of every process ran into the 2-s bound and fell back; later launches were fine).  So the
// form is only offered when the compiled kernel has no private segment -- a toolchain that starts spilling loses speed, not
// two seconds per process.
static bool lstm_persist_scratch_free(int U_pad)
{
    static int state[2] = {-1, -1};
    int& st = state[U_pad == 512 ? 0 : 1];
    if (st < 0) {
        hipFuncAttributes at{};
        const hipError_t e = U_pad == 512 ? hipFuncGetAttributes(&at, (const void*)lstm_persist_kernel<4>) : hipFuncGetAttributes(&at, (const void*)lstm_persist_kernel<2>);
        st = (e == hipSuccess && at.localSizeBytes == 0) ? 1 : 0;
        if (!st) fprintf(stderr, "libamfwd: the persistent recurrent kernel of this build uses scratch memory (%zu B per lane); staying on the per-step kernels\n", (size_t)at.localSizeBytes);
    }
    return st == 1;
}

// Workgroups of lstm_persist_kernel the device holds at once: the occupancy the runtime reports for this kernel with its
// dynamic LDS (147 KB: one per CU) x the CUs the device really has (a partitioned or harvested part has fewer than 256).
int lstm_persist_resident_wgs(int U_pad, int n_cus)
{
    if (U_pad != 256 && U_pad != 512) return 0;
    int per_cu = 0;
    const hipError_t e = U_pad == 512 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lstm_persist_kernel<4>, 512, LSTM_PERSIST_LDS)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lstm_persist_kernel<2>, 512, LSTM_PERSIST_LDS);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu > 0 ? n_cus : 0;   // the groups are laid out one workgroup per CU even where two would fit
}

int lstm_persist_capacity(int U_pad, int n_layers, int max_wgs, bool ext_upward)
{
    if ((U_pad != 256 && U_pad != 512) || n_layers < 1 || n_layers > LSTM_MAX_LAYERS) return 0;
    if (!lstm_persist_scratch_free(U_pad)) return 0;
    const int nb = U_pad / LSTM_KB, budget = max_wgs / (U_pad / 16);
    int tiles = 0;
    for (int t0 = 1; t0 <= 4096; ++t0) {
        int total = 0;
        for (int l = 0; l < n_layers; ++l) { const int ntmax = lstm_persist_ntmax(nb, l > 0 || ext_upward); total += (t0 + ntmax - 1) / ntmax; }
        if (total > budget) break;
        tiles = t0;
    }
    return tiles * 16;
}

bool plan_lstm_persist(LstmPersistArgs& a, int n_slots, const int* n_active, unsigned* cum_out)
{
    if (a.n_layers < 1 || a.n_layers > LSTM_MAX_LAYERS || a.T <= 0 || n_slots <= 0) return false;
    if (a.U_pad != 256 && a.U_pad != 512) return false;
    const int nb = a.U_pad / LSTM_KB, tpg = a.U_pad / 16, budget = a.max_wgs / tpg;   // groups the chip can hold at one workgroup per CU
    const int tiles0 = (n_slots + 15) / 16;
    int ng[LSTM_MAX_LAYERS], total = 0;
    for (int l = 0; l < a.n_layers; ++l) {
        const int ntmax = lstm_persist_ntmax(nb, l > 0 || a.ext_upward != 0);
        ng[l] = (tiles0 + ntmax - 1) / ntmax;
        total += ng[l];
    }
    if (total > budget) return false;
    // spare workgroups go, one group at a time, to the layer whose costliest group is the heaviest (k-batches x tiles)
    for (;;) {
        if (total >= budget) break;
        int best = -1; long best_w = 0;
        for (int l = 0; l < a.n_layers; ++l) {
            if (ng[l] >= tiles0) continue;
            const long wk = (long)((l > 0 || a.ext_upward) ? 2 * nb : nb) * ((tiles0 + ng[l] - 1) / ng[l]);
            if (wk > best_w) { best_w = wk; best = l; }
        }
        if (best < 0) break;
        ++ng[best]; ++total;
    }
    int g0 = 0;
    for (int l = 0; l < a.n_layers; ++l) { a.L[l].n_groups = ng[l]; a.L[l].group0 = g0; g0 += ng[l]; }
    a.n_groups_total = total;
    if (cum_out) {
        for (int l = 0; l < a.n_layers; ++l)
            for (int g = 0; g < ng[l]; ++g) {
                unsigned* row = cum_out + (size_t)(a.L[l].group0 + g) * a.T;
                unsigned c = 0;
                for (int t = 0; t < a.T; ++t) {
                    int na = n_active[t] - a.slot_base;
                    na = na < 0 ? 0 : (na > a.slot_cap ? a.slot_cap : na);
                    const int tiles = (na + 15) / 16, q = (tiles + ng[l] - 1) / ng[l];
                    if (tiles > 0 && g * q < tiles) c += (unsigned)tpg;
                    row[t] = c;
                }
            }
    }
    return true;
}

size_t lstm_persist_cum_size(const LstmPersistArgs& a) { return (size_t)a.n_groups_total * a.T; }

hipError_t launch_lstm_persist(hipStream_t s, const LstmPersistArgs& a)
{
    const int grid = a.n_groups_total * (a.U_pad / 16);
    if (grid <= 0 || grid > a.max_wgs) return hipErrorInvalidValue;   // every workgroup must be resident: never more than the device holds
    if (a.t_begin < 0 || a.t_begin >= a.T || a.t_end < 0 || a.t_end > a.T || (a.t_end > 0 && a.t_end <= a.t_begin)) return hipErrorInvalidValue;
    if (a.U_pad == 512) hipLaunchKernelGGL(lstm_persist_kernel<4>, dim3(grid), dim3(512), LSTM_PERSIST_LDS, s, a);
    else if (a.U_pad == 256) hipLaunchKernelGGL(lstm_persist_kernel<2>, dim3(grid), dim3(512), LSTM_PERSIST_LDS, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

int lstm_persist_mode()
{
    const char* v = getenv("AM_LSTM_PERSIST");   // 0 keeps the per-step kernels (A/B measurements, cross-form parity tests); read at every upload
    return v ? atoi(v) : 1;
}

constexpr size_t LSTM_THIN_LDS_MAX = 72 * 1024;   // two blocks per CU stay co-resident
static int lstm_wreg_mode()
{
    static int mode = -1;   // AM_LSTM_WREG=0 keeps the streaming kernel (A/B measurements)
    if (mode < 0) { const char* v = getenv("AM_LSTM_WREG"); mode = v ? atoi(v) : 1; }
    return mode;
}

hipError_t init_lstm_kernels()
{
    hipError_t we = hipFuncSetAttribute((const void*)lstm_wreg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_WREG_LDS);
    if (we != hipSuccess) return we;
    we = hipFuncSetAttribute((const void*)lstm_persist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_PERSIST_LDS);
    if (we != hipSuccess) return we;
    we = hipFuncSetAttribute((const void*)lstm_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_PERSIST_LDS);
    if (we != hipSuccess) return we;

    we = hipFuncSetAttribute((const void*)lstm_wreg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_WREG_PAIR_LDS);
    if (we != hipSuccess) return we;
    hipError_t e = hipFuncSetAttribute((const void*)lstm_thin_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_THIN_LDS_MAX);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)lstm_thin_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_THIN_LDS_MAX);
}

static int lstm_thin_mode()
{
    static int mode = -1;   // AM_LSTM_THIN=0 disables the few-rows kernel (A/B measurements)
    if (mode < 0) { const char* v = getenv("AM_LSTM_THIN"); mode = v ? atoi(v) : 1; }
    return mode;
}

static int lstm_blocks_for(int n_active, int U_pad, int tpb)
{
    const int n_ut = U_pad / 16, ut_groups = (n_ut + 7) / 8, n_tiles = (n_active + 15) / 16;
    return 8 * ut_groups * ((n_tiles + tpb - 1) / tpb);
}

hipError_t launch_lstm_wavefront(hipStream_t s, LstmWavefrontArgs& a)
{
    if (a.n <= 0 || a.n > LSTM_MAX_LAYERS || a.U_pad % LSTM_KB) return hipErrorInvalidValue;
    a.ut_groups = (a.U_pad / 16 + 7) / 8;
    for (int l = 0; l < a.n; ++l)
        if (a.L[l].Hin && a.L[l].Kin_pad % LSTM_KB) return hipErrorInvalidValue;
    int max_tiles = 0;
    for (int l = 0; l < a.n; ++l) max_tiles = std::max(max_tiles, (a.L[l].n_active + 15) / 16);
    if (max_tiles == 0) return hipSuccess;
    // measured per launch (3x512, graph replay): <= 16 rows 9.4 us thin vs 15 us wide; <= 32 rows 13.5 vs 15.5;
    // at 33..64 rows the thin form (4 row tiles, A re-read by every block) loses, 18.3 vs 17 us
    int nc_max = 0;   // 64-k chains of the longest K in this launch: the thin kernel keeps one partial tile per chain in LDS
    for (int l = 0; l < a.n; ++l)
        nc_max = std::max(nc_max, ((a.L[l].Hin ? a.L[l].Kin_pad : 0) + (a.L[l].first ? 0 : a.U_pad)) / 64);
    const size_t thin_lds = (size_t)std::max(nc_max, 2) * max_tiles * 16 * 17 * sizeof(float);
    if (lstm_thin_mode() && max_tiles <= 2 && thin_lds <= LSTM_THIN_LDS_MAX) {
        const dim3 grid(a.U_pad / 4, a.n);
        if (max_tiles == 1) hipLaunchKernelGGL((lstm_thin_kernel<1>), grid, dim3(512), thin_lds, s, a);
        else hipLaunchKernelGGL((lstm_thin_kernel<2>), grid, dim3(512), thin_lds, s, a);
        return hipGetLastError();
    }
    // Row tiles per block (1..4), chosen per layer.  More tiles per block reuse each weight fragment more, fewer keep
    // every CU busy when few utterances are still active.  Two blocks are co-resident per CU and share its matrix pipes,
    // so a launch takes about (blocks per CU) x (tile-times of the costliest block) plus a fixed cost per block (operand
    // fetch, barriers, gate epilogue; 0.75 tile-times measured); a tile-time is proportional to the layer's K (the
    // layer fed by the hoisted GEMM multiplies only its lateral half).  Layers with the same K share a choice.
    // Measured on 3x512 / 192 utterances: 384 blocks of 3 tiles 33 us; 96 blocks of 4 (half K) + 2 x 192 blocks of 2: see DESIGN.md.
    int Kl[LSTM_MAX_LAYERS], Kmax = 0;
    for (int l = 0; l < a.n; ++l) {
        Kl[l] = (a.L[l].Hin ? a.L[l].Kin_pad : 0) + (a.L[l].first ? 0 : a.U_pad);
        Kmax = std::max(Kmax, Kl[l]);
    }
    // W-resident kernel: every layer's K fits 8 batches.  One block per CU, so a launch takes (rounds of 256 blocks) x
    // (the costliest block's tile-times + ~1.5 tile-times of fixed cost: the weight round trip, the first A tile, the
    // gate epilogue).  Layers whose K fits 4 batches may take up to 6 row tiles per block (half the weight registers).
    bool wreg = lstm_wreg_mode() != 0;
    for (int l = 0; l < a.n; ++l) {
        if (Kl[l] % (4 * LSTM_KB)) wreg = false;
    }
    if (wreg) {
        int best_f = 4, best_p = 4; float best = 1e30f; bool best_pair = false;
        const int pair_mode = lstm_wreg_mode();   // 1: choose by the cost model, 2: never pair, 3: always pair (A/B measurements)
        for (int pair = 0; pair < 2; ++pair) {
            if ((pair && pair_mode == 2) || (!pair && pair_mode == 3)) continue;
            const int tmax = pair ? 2 : 6;   // a 3-tile instantiation of the 128-VGPR form spills
            for (int cf = tmax; cf >= 1; --cf)
                for (int cp = tmax; cp >= 1; --cp) {
                    int blocks = 0; float work = 0.0f; bool any_part = false, any_full = false;
                    for (int l = 0; l < a.n; ++l) {
                        const bool part = Kl[l] < Kmax;
                        any_part |= part; any_full |= !part;
                        const int cand = part ? cp : cf;
                        blocks += lstm_blocks_for(a.L[l].n_active, a.U_pad, cand);
                        work = std::max(work, cand * (float)Kl[l] / (8 * LSTM_KB));
                    }
                    if ((!any_part && cp != tmax) || (!any_full && cf != tmax)) continue;
                    // one block per CU: rounds of 256 blocks, each paying its fixed cost in full; two per CU: rounds of 512,
                    // a CU's two blocks share the matrix pipe (2 x work) and hide each other's fixed cost
                    const int rounds = (blocks + (pair ? 511 : 255)) / (pair ? 512 : 256);
                    const bool two = pair && blocks > 256;
                    const float cost = pair ? rounds * ((two ? 2.0f : 1.0f) * work + 1.0f) : rounds * (work + 1.5f);
                    if (cost < best - 1e-6f) { best = cost; best_f = cf; best_p = cp; best_pair = pair != 0; }
                }
        }
        int total = 0;
        for (int l = 0; l < a.n; ++l) {
            const int tpb = Kl[l] < Kmax ? best_p : best_f;
            const int n_tiles = (a.L[l].n_active + 15) / 16, n_rt = (n_tiles + tpb - 1) / tpb;
            a.L[l].tpb = n_rt > 0 ? (n_tiles + n_rt - 1) / n_rt : 1;
            a.L[l].block0 = total;
            total += 8 * a.ut_groups * n_rt;
        }
        if (total == 0) return hipSuccess;
        if (best_pair) hipLaunchKernelGGL(lstm_wreg_kernel<true>, dim3(total), dim3(512), LSTM_WREG_PAIR_LDS, s, a);
        else hipLaunchKernelGGL(lstm_wreg_kernel<false>, dim3(total), dim3(512), LSTM_WREG_LDS, s, a);
        return hipGetLastError();
    }
    int tpb_full = 4, tpb_part = 4;   // layers with K == Kmax / with a shorter K
    {
        float best = 1e30f;
        for (int cf = 4; cf >= 1; --cf)
            for (int cp = 4; cp >= 1; --cp) {
                int blocks = 0; float work = 0.0f; bool any_part = false;
                for (int l = 0; l < a.n; ++l) {
                    const bool part = Kl[l] < Kmax;
                    any_part |= part;
                    const int cand = part ? cp : cf;
                    blocks += lstm_blocks_for(a.L[l].n_active, a.U_pad, cand);
                    work = std::max(work, cand * (Kmax > 0 ? (float)Kl[l] / Kmax : 1.0f));
                }
                if (!any_part && cp != 4) continue;
                const int rounds = (blocks + 255) / 256;
                const float cost = rounds * work + 0.75f * (1.0f + 0.5f * (rounds - 1));
                if (cost < best - 1e-6f) { best = cost; tpb_full = cf; tpb_part = cp; }
            }
    }
    int total = 0;
    for (int l = 0; l < a.n; ++l) {
        const int tpb = Kl[l] < Kmax ? tpb_part : tpb_full;
        const int n_tiles = (a.L[l].n_active + 15) / 16, n_rt = (n_tiles + tpb - 1) / tpb;
        a.L[l].tpb = n_rt > 0 ? (n_tiles + n_rt - 1) / n_rt : 1;   // balanced: ceil(tiles / row blocks)
        a.L[l].block0 = total;
        total += 8 * a.ut_groups * n_rt;
        if (a.L[l].Hin && a.L[l].Kin_pad % LSTM_KB) return hipErrorInvalidValue;
    }
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(lstm_wavefront_kernel, dim3(total), dim3(512), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent cells other than L.LSTM: GRU (chainer L.GRU), the reference's MGRU with and without reset
// gate (MGRU.py:10-85) and StatefulPeepholeLSTM (chainer_networks.py:103-161).  One kernel shape for every cell and phase:
//   pre[row, col] = sum over the K segments of A_s[row, :] . W[col, segment s]  + bias[col]
// followed by the cell's element-wise epilogue.  Cells with a reset gate / an output peephole need a second, dependent
// product inside the step (U(r*h), peep_o(c')): they run as two phases, i.e. two launches per wavefront tick.
// Geometry as lstm_thin_kernel: a block owns one 16-column MFMA tile = 16/G units x G gates and up to R 16-row tiles;
// K is cut into 64-k chains dealt to the 8 waves, operands go straight from L2 to registers, the chain partials are
// folded through LDS in k order (so a frame's result does not depend on which utterances share its launch).
// grid = (G*U_pad/16 column tiles, layers in this tick, row groups of R tiles).
// ---------------------------------------------------------------------------------------------------
template <int G, int EPI>
__device__ __forceinline__ void rec_epilogue(const float (&pre)[G], float e0, float e1, int first, int act, const RecLayerStep& Ls, size_t eidx)
{
    if (EPI == REC_MGRU || EPI == REC_GRU_B) {
        // linear_interpolate(z, h~, h) = z*h~ + (1-z)*h ; without a previous state h = z*h~ (MGRU.py:77-84)
        const float z = EPI == REC_MGRU ? am_sigmoid_rec(pre[0]) : e1;
        const float hb = am_activate_rec(pre[G - 1], act);
        float h = z * hb;
        if (!first) { const float keep = (1.0f - z) * e0; h = h + keep; }
        Ls.O0[eidx] = h;
    } else if (EPI == REC_GRU_A) {
        Ls.O0[eidx] = am_sigmoid_rec(pre[G - 1]);                                   // z
        Ls.O1[eidx] = first ? 0.0f : am_sigmoid_rec(pre[0]) * e0;                   // r * h_prev
    } else if (EPI == REC_PEEP_A) {
        const float a = am_tanh_rec(pre[0]);
        const float i = am_sigmoid_rec(pre[1 % G]);
        const float f = am_sigmoid_rec(pre[2 % G]);
        const float ai = a * i;
        const float fc = f * e0;
        Ls.O0[eidx] = ai + fc;                                                  // c'
        Ls.O1[eidx] = pre[3 % G];                                               // o before its peephole term
    } else {                                                                    // REC_PEEP_B
        const float o = am_sigmoid_rec(e1 + pre[0]);
        Ls.O0[eidx] = o * am_tanh_rec(e0);
    }
}

template <int G, int EPI, int R>
__global__ __launch_bounds__(512) void rec_step_kernel(const RecStepArgs args)
{
    extern __shared__ float sp[];                           // [chain][r][16 rows][17]
    constexpr int NU = 16 / G;                              // units per column tile
    const RecLayerStep& Ls = args.L[blockIdx.y];
    const int U_pad = args.U_pad;
    const int tile = blockIdx.x;
    const int n_active = Ls.n_active;
    const int row0 = blockIdx.z * R * 16;
    const int tiles_left = (n_active - row0 + 15) >> 4;
    const int ntile = tiles_left < R ? tiles_left : R;
    if (ntile <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int first = Ls.first;

    // element-wise operands of the epilogue first: item = (row tile, row, unit in tile)
    constexpr int ITEMS = 16 * NU;                          // per row tile
    const int er = tid / ITEMS, erow = (tid % ITEMS) / NU, euu = tid % NU;
    const int eslot = row0 + er * 16 + erow;
    const bool elive = tid < ITEMS * R && er < ntile && eslot < n_active;
    const size_t eidx = (size_t)eslot * U_pad + tile * NU + euu;
    float bias[G], e0 = 0.0f, e1 = 0.0f;
#pragma unroll
    for (int g = 0; g < G; ++g) bias[g] = 0.0f;
    if (elive) {
#pragma unroll
        for (int g = 0; g < G; ++g) bias[g] = Ls.bias[tile * 16 + g * NU + euu];
        if (EPI == REC_PEEP_B) { e0 = Ls.E0[eidx]; e1 = Ls.E1[eidx]; }
        else if (EPI == REC_GRU_B) { e1 = Ls.E1[eidx]; if (!first) e0 = Ls.E0[eidx]; }
        else if (!first) e0 = Ls.E0[eidx];
    }

    const int K0 = Ls.K[0], K01 = K0 + Ls.K[1], Ktot = K01 + Ls.K[2];
    const int nc = Ktot >> 6;                               // 64-k chains
    const int m = (nc + 7) >> 3;                            // chains per wave: wave w owns [w*m, (w+1)*m)
    const float* wrow = Ls.W + (size_t)(tile * 16 + l15) * Ls.ldw + 4 * lq;
    for (int c0 = 0; c0 < m; c0 += 2) {                     // two chains = 8 sub-steps of 16 k per round trip
        const int ch0 = w * m + c0;
        const int nch = (ch0 >= nc) ? 0 : ((c0 + 1 < m && ch0 + 1 < nc) ? 2 : 1);
        f32x4 wv[8], av[R][8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if ((s >> 2) < nch) {
                const int k0 = ch0 * 64 + s * 16;
                const float* ap; int ld;
                if (k0 < K0) { ld = K0; ap = Ls.A[0] + k0; }
                else if (k0 < K01) { ld = Ls.K[1]; ap = Ls.A[1] + (k0 - K0); }
                else { ld = Ls.K[2]; ap = Ls.A[2] + (k0 - K01); }
                wv[s] = *(const f32x4*)(wrow + k0);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < ntile) av[r][s] = *(const f32x4*)(ap + (size_t)(row0 + r * 16 + l15) * ld + 4 * lq);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h < nch) {
                f32x4 acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 4 * h; s < 4 * h + 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (r < ntile) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][s][j], wv[s][j], acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        sp[(((ch0 + h) * R + r) * 16 + lq * 4 + reg) * 17 + l15] = acc[r][reg];
            }
        }
    }
    __syncthreads();
    if (!elive) return;
    float pre[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // the order rec_wide_kernel produces: even chains folded in k order, odd chains likewise, (even + odd) + bias
        float ev = 0.0f, od = 0.0f;
        for (int c = 0; c < nc; c += 2) {
            ev += sp[((c * R + er) * 16 + erow) * 17 + g * NU + euu];
            if (c + 1 < nc) od += sp[(((c + 1) * R + er) * 16 + erow) * 17 + g * NU + euu];
        }
        pre[g] = (ev + od) + bias[g];
    }
    rec_epilogue<G, EPI>(pre, e0, e1, first, args.act, Ls, eidx);
}

// ---------------------------------------------------------------------------------------------------
// The same phases for MANY active utterances (> 32 rows): geometry of lstm_wreg_kernel.  A block owns four consecutive
// 16-column tiles of the packed weights (64 columns = 64/G units x G gates) and up to 4 row tiles; 8 waves = 4 column tiles
// x 2 K halves; the activation rows are shared through a double-buffered LDS tile (two 128-k batches ahead), the weight
// fragments of four batches are kept in flight in registers, loads in consumption order.  rec_step_kernel re-reads the
// activation rows once per 16 columns; here once per 64.  The number of batches need not be a multiple of the 4-slot
// rotation: batches past the end multiply zero weights (adds +0: the summation order, shared with rec_step_kernel, holds).
// ---------------------------------------------------------------------------------------------------
constexpr int REC_WIDE_ROWS = 64;
constexpr size_t REC_WIDE_LDS = 2 * REC_WIDE_ROWS * LSTM_LDA * sizeof(float);   // 67,584 B; the exchange buffer aliases it

template <int G, int EPI, int NT, bool ZW>   // ZW: the batch count is not a multiple of the 4-slot rotation (zero weights past the end)
__device__ __forceinline__ void rec_wide_body(const RecLayerStep& Ls, int U_pad, int act, int bt, int row0, float* smem)
{
    constexpr int NU = 16 / G, UB = 4 * NU;               // units per column tile / per block
    constexpr int ROWS = NT * 16;
    constexpr int EP = (ROWS * UB + 511) / 512;           // epilogue items per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int first = Ls.first, n_active = Ls.n_active;
    float* sA0 = smem;
    float* sA1 = smem + REC_WIDE_ROWS * LSTM_LDA;

    float bias[EP][G], e0[EP], e1[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e, erow = idx / UB, ub = idx % UB, slot = row0 + erow;
        const bool live = erow < ROWS && slot < n_active;
        const size_t eidx = (size_t)slot * U_pad + bt * UB + ub;
        e0[e] = 0.0f; e1[e] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) bias[e][g] = live ? Ls.bias[bt * 64 + (ub / NU) * 16 + g * NU + ub % NU] : 0.0f;
        if (live) {
            if (EPI == REC_PEEP_B) { e0[e] = Ls.E0[eidx]; e1[e] = Ls.E1[eidx]; }
            else if (EPI == REC_GRU_B) { e1[e] = Ls.E1[eidx]; if (!first) e0[e] = Ls.E0[eidx]; }
            else if (!first) e0[e] = Ls.E0[eidx];
        }
    }

    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int K0 = Ls.K[0], K01 = K0 + Ls.K[1], Ktot = K01 + Ls.K[2];
    const int nb = Ktot / LSTM_KB;                        // every segment is a multiple of 128 (launcher)
    if (nb > 0) {
        f32x4 av[2][NT], wv[4][LSTM_KS];
        auto load_a = [&](int b, f32x4 (&dst)[NT]) {
            b = b < nb ? b : nb - 1;
            const int k0 = b * LSTM_KB;
            const float* ap; int K;
            if (k0 < K0) { K = K0; ap = Ls.A[0] + (size_t)row0 * K + k0; }
            else if (k0 < K01) { K = Ls.K[1]; ap = Ls.A[1] + (size_t)row0 * K + (k0 - K0); }
            else { K = Ls.K[2]; ap = Ls.A[2] + (size_t)row0 * K + (k0 - K01); }
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);
                dst[e] = *(const f32x4*)(ap + (size_t)r * K + 4 * c4);
            }
        };
        const float* wbase = Ls.W + (size_t)(bt * 64 + cg * 16 + l15) * Ls.ldw + kh * (LSTM_KB / 2) + 4 * lq;
        auto load_w = [&](int b, f32x4 (&dst)[LSTM_KS]) {
            const bool real = b < nb;
            const float* wp = wbase + (size_t)(real ? b : nb - 1) * LSTM_KB;
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) {
                const f32x4 v = *(const f32x4*)(wp + 16 * s);
                // without ZW the only batches past the end are the refills behind the last rotation: loaded, never multiplied
                dst[s] = (!ZW || real) ? v : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        load_a(0, av[0]); load_w(0, wv[0]);
        load_a(1, av[1]); load_w(1, wv[1]);
        load_w(2, wv[2]); load_w(3, wv[3]);
        for (int b0 = 0; b0 < nb; b0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + i;
                float* sA = (i & 1) ? sA1 : sA0;
#pragma unroll
                for (int e = 0; e < NT; ++e) {
                    const int idx = tid + 512 * e, r = idx / (LSTM_KB / 4), c4 = idx % (LSTM_KB / 4);
                    *(f32x4*)(sA + r * LSTM_LDA + 4 * c4) = av[i & 1][e];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                load_a(b + 2, av[i & 1]);
                const float* sa = sA + l15 * LSTM_LDA + kh * (LSTM_KB / 2) + 4 * lq;
                f32x4 pb[NT];
#pragma unroll
                for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < LSTM_KS; ++s) {
                    f32x4 a[NT];
#pragma unroll
                    for (int r = 0; r < NT; ++r) a[r] = *(const f32x4*)(sa + r * 16 * LSTM_LDA + 16 * s);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < NT; ++r) pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][j], wv[i][s][j], pb[r], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < NT; ++r) acc[r] += pb[r];
                load_w(b + 4, wv[i]);
            }
        }
        __syncthreads();
    }
    float* sg = smem;                       // [kh][column tile][ROWS][17]
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + cg) * ROWS + r * 16 + lq * 4 + reg) * 17 + l15] = acc[r][reg];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int idx = tid + 512 * e, erow = idx / UB, ub = idx % UB, slot = row0 + erow;
        if (erow < ROWS && slot < n_active) {
            const int ct = ub / NU, uu = ub % NU;
            float pre[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                pre[g] = (sg[((0 * 4 + ct) * ROWS + erow) * 17 + g * NU + uu] + sg[((1 * 4 + ct) * ROWS + erow) * 17 + g * NU + uu]) + bias[e][g];
            rec_epilogue<G, EPI>(pre, e0[e], e1[e], first, act, Ls, (size_t)slot * U_pad + bt * UB + ub);
        }
    }
}

template <int G, int EPI>
__global__ __launch_bounds__(512, 2) void rec_wide_kernel(const RecStepArgs args)
{
    extern __shared__ __attribute__((aligned(16))) float rec_wide_smem[];
    int li = 0;
#pragma unroll
    for (int l = 1; l < LSTM_MAX_LAYERS; ++l) if (l < args.n && (int)blockIdx.x >= args.L[l].block0) li = l;
    const RecLayerStep& Ls = args.L[li];
    const int U_pad = args.U_pad, groups = args.bt_groups;
    const int bid = blockIdx.x - Ls.block0;
    const int bt = (bid & 7) + 8 * ((bid >> 3) % groups);   // block tiles of one row block share an XCD
    const int rt = (bid >> 3) / groups;
    if (bt * 64 >= G * U_pad) return;
    const int row0 = rt * Ls.tpb * 16;
    const int tiles_left = (Ls.n_active - row0 + 15) / 16;
    const int ntile = tiles_left < Ls.tpb ? tiles_left : Ls.tpb;
    const bool zw = (((Ls.K[0] + Ls.K[1] + Ls.K[2]) / LSTM_KB) & 3) != 0;   // uniform per block
    if (zw) {
        switch (ntile) {
        case 1: rec_wide_body<G, EPI, 1, true>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 2: rec_wide_body<G, EPI, 2, true>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 3: rec_wide_body<G, EPI, 3, true>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 4: rec_wide_body<G, EPI, 4, true>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        default: break;
        }
    } else {
        switch (ntile) {
        case 1: rec_wide_body<G, EPI, 1, false>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 2: rec_wide_body<G, EPI, 2, false>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 3: rec_wide_body<G, EPI, 3, false>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        case 4: rec_wide_body<G, EPI, 4, false>(Ls, U_pad, args.act, bt, row0, rec_wide_smem); break;
        default: break;
        }
    }
}

template <int G, int EPI>
static hipError_t rec_wide_launch(hipStream_t s, RecStepArgs a)
{
    // one block per CU: rounds of 256 blocks x (tile-times of the costliest block + ~1.5 of fixed cost); layers with a
    // shorter K (the first layer: its input is narrower) may take different row tiling
    const int n_bt = G * a.U_pad / 64;
    a.bt_groups = (n_bt + 7) / 8;
    int Kl[LSTM_MAX_LAYERS], Kmax = 0;
    for (int l = 0; l < a.n; ++l) { Kl[l] = a.L[l].K[0] + a.L[l].K[1] + a.L[l].K[2]; Kmax = std::max(Kmax, Kl[l]); }
    int best_f = 4, best_p = 4; float best = 1e30f;
    for (int cf = 4; cf >= 1; --cf)
        for (int cp = 4; cp >= 1; --cp) {
            int blocks = 0; float work = 0.0f; bool any_part = false, any_full = false;
            for (int l = 0; l < a.n; ++l) {
                const bool part = Kl[l] < Kmax;
                any_part |= part; any_full |= !part;
                const int cand = part ? cp : cf;
                const int n_tiles = (a.L[l].n_active + 15) / 16;
                blocks += 8 * a.bt_groups * ((n_tiles + cand - 1) / cand);
                work = std::max(work, cand * (Kmax > 0 ? (float)Kl[l] / Kmax : 1.0f));
            }
            if ((!any_part && cp != 4) || (!any_full && cf != 4)) continue;
            const int rounds = (blocks + 255) / 256;
            const float cost = rounds * (work + 1.5f);
            if (cost < best - 1e-6f) { best = cost; best_f = cf; best_p = cp; }
        }
    int total = 0;
    for (int l = 0; l < a.n; ++l) {
        const int tpb = Kl[l] < Kmax ? best_p : best_f;
        const int n_tiles = (a.L[l].n_active + 15) / 16, n_rt = (n_tiles + tpb - 1) / tpb;
        a.L[l].tpb = n_rt > 0 ? (n_tiles + n_rt - 1) / n_rt : 1;
        a.L[l].block0 = total;
        total += 8 * a.bt_groups * n_rt;
    }
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL((rec_wide_kernel<G, EPI>), dim3(total), dim3(512), REC_WIDE_LDS, s, a);
    return hipGetLastError();
}

template <int G, int EPI>
static hipError_t rec_wide_init()
{
    return hipFuncSetAttribute((const void*)rec_wide_kernel<G, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)REC_WIDE_LDS);
}

static int rec_wide_mode()
{
    static int mode = -1;   // AM_REC_WIDE=0 keeps rec_step_kernel for every batch size (A/B measurements)
    if (mode < 0) { const char* v = getenv("AM_REC_WIDE"); mode = v ? atoi(v) : 1; }
    return mode;
}

constexpr size_t REC_LDS_MAX = 72 * 1024;

template <int G, int EPI>
static hipError_t rec_launch(hipStream_t s, const RecStepArgs& a, int max_tiles, int nc_max)
{
    // R = 2 row tiles per block while the chain partials of two tiles fit the LDS budget of two co-resident blocks
    const int R = (max_tiles >= 2 && (size_t)nc_max * 2 * 16 * 17 * sizeof(float) <= REC_LDS_MAX) ? 2 : 1;
    const size_t lds = (size_t)std::max(nc_max, 1) * R * 16 * 17 * sizeof(float);
    if (lds > REC_LDS_MAX) return hipErrorInvalidValue;
    const dim3 grid(G * a.U_pad / 16, a.n, (max_tiles + R - 1) / R);
    if (R == 2) hipLaunchKernelGGL((rec_step_kernel<G, EPI, 2>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((rec_step_kernel<G, EPI, 1>), grid, dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int G, int EPI>
static hipError_t rec_init()
{
    hipError_t e = hipFuncSetAttribute((const void*)rec_step_kernel<G, EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)REC_LDS_MAX);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)rec_step_kernel<G, EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)REC_LDS_MAX);
}

hipError_t init_rec_kernels()
{
    hipError_t e;
    if ((e = rec_init<2, REC_MGRU>()) != hipSuccess) return e;
    if ((e = rec_init<2, REC_GRU_A>()) != hipSuccess) return e;
    if ((e = rec_init<1, REC_GRU_B>()) != hipSuccess) return e;
    if ((e = rec_init<4, REC_PEEP_A>()) != hipSuccess) return e;
    if ((e = rec_init<1, REC_PEEP_B>()) != hipSuccess) return e;
    if ((e = rec_wide_init<2, REC_MGRU>()) != hipSuccess) return e;
    if ((e = rec_wide_init<2, REC_GRU_A>()) != hipSuccess) return e;
    if ((e = rec_wide_init<1, REC_GRU_B>()) != hipSuccess) return e;
    if ((e = rec_wide_init<4, REC_PEEP_A>()) != hipSuccess) return e;
    return rec_wide_init<1, REC_PEEP_B>();
}

hipError_t launch_rec_step(hipStream_t s, int epilogue, const RecStepArgs& a)
{
    if (a.n <= 0 || a.n > LSTM_MAX_LAYERS || a.U_pad % 64) return hipErrorInvalidValue;
    int max_tiles = 0, nc_max = 0;
    for (int l = 0; l < a.n; ++l) {
        const RecLayerStep& L = a.L[l];
        if (L.K[0] % 64 || L.K[1] % 64 || L.K[2] % 64) return hipErrorInvalidValue;
        max_tiles = std::max(max_tiles, (L.n_active + 15) / 16);
        nc_max = std::max(nc_max, (L.K[0] + L.K[1] + L.K[2]) / 64);
    }
    if (max_tiles == 0) return hipSuccess;
    bool wide = rec_wide_mode() != 0 && max_tiles > 2;      // <= 32 rows: the K-split form above is the faster one
    for (int l = 0; l < a.n; ++l)
        if (a.L[l].K[0] % LSTM_KB || a.L[l].K[1] % LSTM_KB || a.L[l].K[2] % LSTM_KB || a.L[l].ldw % 4) wide = false;
    if (wide) {
        switch (epilogue) {
        case REC_MGRU: return rec_wide_launch<2, REC_MGRU>(s, a);
        case REC_GRU_A: return rec_wide_launch<2, REC_GRU_A>(s, a);
        case REC_GRU_B: return rec_wide_launch<1, REC_GRU_B>(s, a);
        case REC_PEEP_A: return rec_wide_launch<4, REC_PEEP_A>(s, a);
        case REC_PEEP_B: return rec_wide_launch<1, REC_PEEP_B>(s, a);
        }
        return hipErrorInvalidValue;
    }
    switch (epilogue) {
    case REC_MGRU: return rec_launch<2, REC_MGRU>(s, a, max_tiles, nc_max);
    case REC_GRU_A: return rec_launch<2, REC_GRU_A>(s, a, max_tiles, nc_max);
    case REC_GRU_B: return rec_launch<1, REC_GRU_B>(s, a, max_tiles, nc_max);
    case REC_PEEP_A: return rec_launch<4, REC_PEEP_A>(s, a, max_tiles, nc_max);
    case REC_PEEP_B: return rec_launch<1, REC_PEEP_B>(s, a, max_tiles, nc_max);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// Triphone Viterbi decoder (consumer of surface A; reference recog_src/source/phone_recog.cpp:138-254).
// One workgroup per utterance (the reference parallelises over utterances with OpenMP, main.cpp:51); inside a frame the
// triphone updates are independent (each reads last frame's scores), then the per-node maxima over the arriving
// triphones are independent; two barriers per frame.  Every score is one float add of two floats, compared with the
// reference's strict inequalities (ties keep the stay branch / the first maximum), so segmentations are bit-identical.
// The back-trace is bounded by the frame count: the reference's loop does not terminate for utterances that cannot reach
// a triphone exit (fewer than 3 frames); such utterances yield zero segments here.
// ---------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void viterbi_kernel(const ViterbiArgs a)
{
    const int u = blockIdx.x, tid = threadIdx.x;
    const long long row0 = a.offsets[u];
    const int N = (int)(a.offsets[u + 1] - row0);
    const int T = a.T, B = a.B;
    // per-utterance state, one 16-B record per triphone: scores of its 3 states / their entry times
    f32x4* pp = (f32x4*)a.pp + (size_t)u * T;
    i32x4* tt = (i32x4*)a.tt + (size_t)u * T;
    float* pout = a.pout + (size_t)u * T;      // exit score of every triphone after this frame
    float* p = a.p + (size_t)u * B;
    int* bI = a.bestI + (size_t)row0 * B;
    int* bT = a.bestT + (size_t)row0 * B;
    const float NINF = -1e20f;
    for (int i = tid; i < B; i += 1024) p[i] = a.p0[i];
    for (int i = tid; i < T; i += 1024) { pp[i] = (f32x4){NINF, NINF, NINF, 0.0f}; tt[i] = (i32x4){0, 0, 0, 0}; }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const float* lp1 = a.lp + (size_t)(row0 + k) * a.ldl;
        for (int n = tid; n < T; n += 1024) {
            const i32x4 g = a.tri_idx[n];        // {node in, state 0, state 1, state 2}
            const f32x4 sl = a.tri_self[n], tr = a.tri_trans[n];
            const f32x4 q = pp[n];
            const i32x4 t = tt[n];
            const float pIn = p[g[0]] + a.PIP;
            const float st0 = q[0] + sl[0], st1 = q[1] + sl[1], st2 = q[2] + sl[2];
            const float go0 = pIn, go1 = q[0] + tr[0], go2 = q[1] + tr[1];
            const float e0 = lp1[g[1]], e1 = lp1[g[2]], e2 = lp1[g[3]];
            f32x4 nq; i32x4 nt;
            if (go2 > st2) { nq[2] = go2 + e2; nt[2] = t[1]; } else { nq[2] = st2 + e2; nt[2] = t[2]; }
            if (go1 > st1) { nq[1] = go1 + e1; nt[1] = t[0]; } else { nq[1] = st1 + e1; nt[1] = t[1]; }
            if (go0 > st0) { nq[0] = go0 + e0; nt[0] = k - 1; } else { nq[0] = st0 + e0; nt[0] = t[0]; }
            nq[3] = 0.0f; nt[3] = 0;
            pp[n] = nq;
            tt[n] = nt;
            pout[n] = nq[2] + tr[2];
        }
        __syncthreads();
        for (int i = tid; i < B; i += 1024) {
            int mxi = 0;
            float mx = NINF;
            for (int s = a.outOff[i]; s < a.outOff[i + 1]; ++s) {
                const int ids = a.outVal[s];
                const float pOut = pout[ids];
                if (mx < pOut) { mx = pOut; mxi = ids; }
            }
            p[i] = mx + a.lm[i];
            bI[(size_t)k * B + i] = mxi;
            bT[(size_t)k * B + i] = tt[mxi][2];
        }
        __syncthreads();
    }
    if (tid == 0) {
        int mxi = 0;
        float mx = NINF;
        for (int i = 0; i < a.P; ++i) {
            const int ids = a.endIdx[i];
            if (mx < p[ids]) { mx = p[ids]; mxi = ids; }
        }
        int count = 0, t = N - 1;
        bool ok = true;
        while (t >= 0) {
            if (count >= N) { ok = false; break; }       // the reference would loop forever here
            const int id = bI[(size_t)t * B + mxi];
            const int end = t + 1;
            t = bT[(size_t)t * B + mxi];
            a.seg_end[row0 + count] = end;
            a.seg_begin[row0 + count] = t + 1;
            a.seg_tri[row0 + count] = id;
            mxi = a.tri_idx[id][0];
            ++count;
        }
        a.seg_count[u] = ok ? count : 0;   // segments are stored last-to-first; the host reverses them
    }
}

// Compact form (graphs with < 65536 nodes and posterior columns, utterances < 65535 frames -- every TIMIT-sized set-up):
// the decoder is bound by the bytes it streams per triphone and frame, so the records shrink from 116 B to 76 B:
//   graph  c0 = {node | col0 << 16, col1 | col2 << 16, self0, self1}, c1 = {self2, trans0, trans1, trans2}   (32 B, was 48)
//   state  {score0, score1, score2, (t0+1) | (t1+1) << 16} (16 B) + (t2+1) (4 B) read, the same + exit score written
// and every thread handles two triphones per trip (twice the loads in flight: the tail of the run belongs to the longest
// utterances, one CU each, where the per-CU memory parallelism is the limit).  Same arithmetic, same comparisons, same
// tie-breaking as viterbi_kernel: bit-identical segmentations.
// (the arithmetic of one triphone's update on operands already in registers: `pn` = p[node], e0..e2 = the frame's log-posteriors of its states)
__device__ __forceinline__ void viterbi_tri_apply(const i32x4 c0, const f32x4 c1, const i32x4 q, const int t2s, const float pn,
                                                   const float e0, const float e1, const float e2, float PIP, int k, i32x4& nq, int& nt2, float& po)
{
    const float sl0 = __int_as_float(c0[2]), sl1 = __int_as_float(c0[3]), sl2 = c1[0];
    const float tr0 = c1[1], tr1 = c1[2], tr2 = c1[3];
    const float q0 = __int_as_float(q[0]), q1 = __int_as_float(q[1]), q2 = __int_as_float(q[2]);
    const int t0s = q[3] & 0xffff, t1s = (unsigned)q[3] >> 16;   // entry times + 1
    const float pIn = pn + PIP;
    const float st0 = q0 + sl0, st1 = q1 + sl1, st2 = q2 + sl2;
    const float go0 = pIn, go1 = q0 + tr0, go2 = q1 + tr1;
    float n0, n1, n2; int nt0, nt1;
    if (go2 > st2) { n2 = go2 + e2; nt2 = t1s; } else { n2 = st2 + e2; nt2 = t2s; }
    if (go1 > st1) { n1 = go1 + e1; nt1 = t0s; } else { n1 = st1 + e1; nt1 = t1s; }
    if (go0 > st0) { n0 = go0 + e0; nt0 = k; } else { n0 = st0 + e0; nt0 = t0s; }   // k = (k - 1) + 1
    nq[0] = __float_as_int(n0); nq[1] = __float_as_int(n1); nq[2] = __float_as_int(n2); nq[3] = nt0 | (nt1 << 16);
    po = n2 + tr2;
}

__device__ __forceinline__ void viterbi_tri_update(const i32x4 c0, const f32x4 c1, const i32x4 q, const int t2s, const float* __restrict__ p,
                                                    const float* __restrict__ lp1, float PIP, int k, i32x4& nq, int& nt2, float& po)
{
    const int node = c0[0] & 0xffff, col0 = (unsigned)c0[0] >> 16, col1 = c0[1] & 0xffff, col2 = (unsigned)c0[1] >> 16;
    viterbi_tri_apply(c0, c1, q, t2s, p[node], lp1[col0], lp1[col1], lp1[col2], PIP, k, nq, nt2, po);
}

__global__ __launch_bounds__(1024) void viterbi_compact_kernel(const ViterbiArgs a)
{
    const int u = blockIdx.x, tid = threadIdx.x;
    const long long row0 = a.offsets[u];
    const int N = (int)(a.offsets[u + 1] - row0);
    const int T = a.T, B = a.B;
    i32x4* pp = (i32x4*)a.pp + (size_t)u * T;      // {3 scores, t0+1 | t1+1 << 16}
    int* t2 = a.tt + (size_t)u * T;                // t2 + 1
    float* pout = a.pout + (size_t)u * T;
    float* p = a.p + (size_t)u * B;
    int* bI = a.bestI + (size_t)row0 * B;
    int* bT = a.bestT + (size_t)row0 * B;
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    const f32x4* c1 = (const f32x4*)a.tri_self;
    const float NINF = -1e20f;
    const int ninf = __float_as_int(NINF);
    for (int i = tid; i < B; i += 1024) p[i] = a.p0[i];
    for (int i = tid; i < T; i += 1024) { pp[i] = (i32x4){ninf, ninf, ninf, 1 | (1 << 16)}; t2[i] = 1; }   // entry times 0
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const float* lp1 = a.lp + (size_t)(row0 + k) * a.ldl;
        for (int n = tid; n < T; n += 2048) {
            const bool two = n + 1024 < T;
            const int m = two ? n + 1024 : n;
            const i32x4 ga = c0[n], gb = c0[m];
            const f32x4 ha = c1[n], hb = c1[m];
            const i32x4 qa = pp[n], qb = pp[m];
            const int ta = t2[n], tb = t2[m];
            i32x4 nqa, nqb; int nta, ntb; float poa, pob;
            viterbi_tri_update(ga, ha, qa, ta, p, lp1, a.PIP, k, nqa, nta, poa);
            viterbi_tri_update(gb, hb, qb, tb, p, lp1, a.PIP, k, nqb, ntb, pob);
            pp[n] = nqa; t2[n] = nta; pout[n] = poa;
            if (two) { pp[m] = nqb; t2[m] = ntb; pout[m] = pob; }
        }
        __syncthreads();
        for (int i = tid; i < B; i += 1024) {
            int mxi = a.perm ? a.inv0 : 0;                  // internal index of the reference's triphone 0
            float mx = NINF;
            for (int s = a.outOff[i]; s < a.outOff[i + 1]; ++s) {
                const int ids = a.perm ? s : a.outVal[s];   // perm: triphones are stored in arrival-list order
                const float pOut = pout[ids];
                if (mx < pOut) { mx = pOut; mxi = ids; }
            }
            p[i] = mx + a.lm[i];
            bI[(size_t)k * B + i] = mxi;
            bT[(size_t)k * B + i] = t2[mxi] - 1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int mxi = 0;
        float mx = NINF;
        for (int i = 0; i < a.P; ++i) {
            const int ids = a.endIdx[i];
            if (mx < p[ids]) { mx = p[ids]; mxi = ids; }
        }
        int count = 0, t = N - 1;
        bool ok = true;
        while (t >= 0) {
            if (count >= N) { ok = false; break; }       // the reference would loop forever here
            const int id = bI[(size_t)t * B + mxi];
            const int end = t + 1;
            t = bT[(size_t)t * B + mxi];
            a.seg_end[row0 + count] = end;
            a.seg_begin[row0 + count] = t + 1;
            a.seg_tri[row0 + count] = a.perm ? a.outVal[id] : id;   // back to the reference's numbering
            mxi = c0[id][0] & 0xffff;
            ++count;
        }
        a.seg_count[u] = ok ? count : 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Frame-synchronous form of the same decoder: the whole chip advances ALL active utterances by one frame per launch pair
// (triphone update, then node maxima), like the recurrent path's wavefront.  One workgroup per utterance leaves the tail
// of a run to a handful of CUs (the longest utterances) and HBM idle; here every launch streams the state of every active
// utterance, so the pass is HBM-bound throughout.  Utterances are sorted by length (slots), the active ones are the first
// n_active of a step.  Same records, arithmetic, comparisons and tie-breaking as viterbi_compact_kernel.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void viterbi_fs_init_kernel(const ViterbiArgs a, int n_utts)
{
    const int slot = blockIdx.y;
    const int ninf = __float_as_int(-1e20f);
    i32x4* pp = (i32x4*)a.pp + (size_t)slot * a.T;
    unsigned short* t2 = (unsigned short*)a.tt + (size_t)slot * a.T;   // (frame-synchronous forms: entry times are 16-bit, like the two packed in the record)
    float* p = a.p + (size_t)slot * a.B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.T; i += gridDim.x * 256) { pp[i] = (i32x4){ninf, ninf, ninf, 1 | (1 << 16)}; t2[i] = (unsigned short)1; }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.B; i += gridDim.x * 256) p[i] = a.p0[i];
    if (a.inv0_state && blockIdx.x == 0 && threadIdx.x == 0) {
        int* st = a.inv0_state + (size_t)slot * 16;   // parity 0: the state before frame 0
        st[0] = ninf; st[1] = ninf; st[2] = ninf; st[3] = 1 | (1 << 16); st[4] = 1;
    }
}

// Frame-synchronous triphone update, operands gathered from global memory (the tail of a run, and graphs whose two gather arrays do not fit LDS).
// Tried (round 3): one thread = two triphones x S utterance slots, the 32 B of graph constants fetched once per S slots instead of once per
// (slot, triphone) -- 292 (S = 4) against 269 us per frame at 192 active utterances: the constants are cheap L2 hits, the registers cost occupancy.
__global__ __launch_bounds__(256) void viterbi_fs_tri_kernel(const ViterbiArgs a, int k)
{
    const int slot = blockIdx.y;
    const int T = a.T;
    const int n = blockIdx.x * 512 + threadIdx.x;          // this thread: triphones n and n + 256
    if (n >= T) return;
    const long long row0 = a.offsets[a.slot_utt[slot]];
    i32x4* pp = (i32x4*)a.pp + (size_t)slot * T;
    unsigned short* t2 = (unsigned short*)a.tt + (size_t)slot * T;
    float* pout = a.pout + (size_t)slot * T;
    const float* p = ((k & 1) ? a.p2 : a.p) + (size_t)slot * a.B;
    const float* lp1 = a.lp + (size_t)(row0 + k) * a.ldl;
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    const f32x4* c1 = (const f32x4*)a.tri_self;
    const bool two = n + 256 < T;
    const int m = two ? n + 256 : n;
    const i32x4 ga = c0[n], gb = c0[m];
    const f32x4 ha = c1[n], hb = c1[m];
    const i32x4 qa = pp[n], qb = pp[m];
    const int ta = t2[n], tb = t2[m];
    i32x4 nqa, nqb; int nta, ntb; float poa, pob;
    viterbi_tri_update(ga, ha, qa, ta, p, lp1, a.PIP, k, nqa, nta, poa);
    viterbi_tri_update(gb, hb, qb, tb, p, lp1, a.PIP, k, nqb, ntb, pob);
    pp[n] = nqa; t2[n] = (unsigned short)nta; pout[n] = poa;
    if (two) { pp[m] = nqb; t2[m] = (unsigned short)ntb; pout[m] = pob; }
}

// The same update with the frame's log-posterior row and the slot's node scores staged in LDS.  Every triphone gathers four 4-B values (p[node],
// lp[col0..2]) from two small arrays (9 KB + 7.6 KB at the real size); as global loads these are 8 wave-instructions per thread pair whose 64
// lanes touch ~60 different cache lines each -- the L1's address path, not HBM, bounded the launch (269 us per frame at 192 active utterances
// for 746 MB of state: 3 TB/s).  A block of 256 threads stages both arrays once (coalesced) and updates 512 * IT triphones from them.
// (two slots per block, the constants shared between them: 66.7 against 59.1 ms per pass -- as above)
template <int IT>
__global__ __launch_bounds__(256) void viterbi_fs_tri_lds_kernel(const ViterbiArgs a, int k)
{
    extern __shared__ float vsm[];                          // [ldl] log-posteriors of frame k, [B] node scores
    const int slot = blockIdx.y, T = a.T, B = a.B, L = (int)a.ldl;
    const long long row0 = a.offsets[a.slot_utt[slot]];
    const float* lp1 = a.lp + (size_t)(row0 + k) * a.ldl;
    const float* p = ((k & 1) ? a.p2 : a.p) + (size_t)slot * B;
    i32x4* pp = (i32x4*)a.pp + (size_t)slot * T;
    unsigned short* t2 = (unsigned short*)a.tt + (size_t)slot * T;
    float* pout = a.pout + (size_t)slot * T;
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    const f32x4* c1 = (const f32x4*)a.tri_self;
    const int base = blockIdx.x * IT * 512 + threadIdx.x;
    // this thread's state and constants first (independent of the staging), then the staging loads: everything is in flight together
    i32x4 ga[IT], gb[IT], qa[IT], qb[IT]; f32x4 ha[IT], hb[IT]; int ta[IT], tb[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int n0 = base + it * 512;
        const int n = n0 < T ? n0 : T - 1, m = n0 + 256 < T ? n0 + 256 : T - 1;   // (past the end: repeats the last triphone's loads, stores nothing)
        ga[it] = c0[n]; gb[it] = c0[m]; ha[it] = c1[n]; hb[it] = c1[m];
        qa[it] = pp[n]; qb[it] = pp[m]; ta[it] = t2[n]; tb[it] = t2[m];
    }
    for (int i = threadIdx.x; i < L; i += 256) vsm[i] = lp1[i];
    for (int i = threadIdx.x; i < B; i += 256) vsm[L + i] = p[i];
    __syncthreads();
    const float* s_lp = vsm;
    const float* s_p = vsm + L;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int n = base + it * 512;
        if (n >= T) break;
        const bool two = n + 256 < T;
        i32x4 nqa, nqb; int nta, ntb; float poa, pob;
        {
            const i32x4 g = ga[it];
            viterbi_tri_apply(g, ha[it], qa[it], ta[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqa, nta, poa);
        }
        {
            const i32x4 g = gb[it];
            viterbi_tri_apply(g, hb[it], qb[it], tb[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqb, ntb, pob);
        }
        pp[n] = nqa; t2[n] = (unsigned short)nta; pout[n] = poa;
        if (two) { pp[n + 256] = nqb; t2[n + 256] = (unsigned short)ntb; pout[n + 256] = pob; }
    }
}

// 8 lanes per node (8 nodes per wave, 32 per block): a lane takes arrivals s0 + l, s0 + l + 8, ...; the first 8 trips are
// unrolled so their loads fly together (a list of <= 64 arrivals costs one round trip; one WAVE per node was latency-bound:
// 442 k waves per launch at 192 utterances, as long as the triphone kernel).  The reference takes the FIRST strict maximum
// in list order (phone_recog.cpp:190-200), i.e. the largest exit score and, among equals, the smallest list position.
__global__ __launch_bounds__(256) void viterbi_fs_node_kernel(const ViterbiArgs a, int k)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int l8 = threadIdx.x & 7;
    const bool valid = i < a.B;
    const float* pout = a.pout + (size_t)slot * a.T;
    const int s0 = valid ? a.outOff[i] : 0, s1 = valid ? a.outOff[i + 1] : 0;
    float mx = -1e20f; int ms = 0x7fffffff, mxi = 0;
    float v[8]; int id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = s0 + l8 + 8 * j;
        id[j] = s < s1 ? (a.perm ? s : a.outVal[s]) : -1;   // perm: triphones stored in arrival-list order -> contiguous reads
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = id[j] >= 0 ? pout[id[j]] : -1e20f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (id[j] >= 0 && mx < v[j]) { mx = v[j]; ms = s0 + l8 + 8 * j; mxi = id[j]; }
    for (int s = s0 + l8 + 64; s < s1; s += 8) {
        const int ids = a.perm ? s : a.outVal[s];
        const float pOut = pout[ids];
        if (mx < pOut) { mx = pOut; ms = s; mxi = ids; }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
        const float omx = __shfl_xor(mx, off);
        const int oms = __shfl_xor(ms, off), omi = __shfl_xor(mxi, off);
        if (omx > mx || (omx == mx && oms < ms)) { mx = omx; ms = oms; mxi = omi; }
    }
    if (valid && l8 == 0) {
        if (ms == 0x7fffffff) mxi = a.perm ? a.inv0 : 0;   // nothing exceeded -1e20: the reference keeps its triphone 0
        const long long row = a.offsets[a.slot_utt[slot]] + k;
        ((k & 1) ? a.p : a.p2)[(size_t)slot * a.B + i] = mx + a.lm[i];
        a.bestI[(size_t)row * a.B + i] = mxi;
        a.bestT[(size_t)row * a.B + i] = (int)((const unsigned short*)a.tt)[(size_t)slot * a.T + mxi] - 1;
    }
}

// Triphone update AND node maxima of a frame in ONE launch (round 3).  With the triphones stored in arrival-list order a node's arrivals are
// consecutive triphones; block b takes the whole lists of nodes [blk_node0[b], blk_node0[b+1]) -- at most 2,048 triphones, cut on node
// boundaries by the host -- so every maximum is taken inside one block, from LDS, right behind the update: no exit-score array written and read
// back (8 of the 48 B a triphone moved per frame), no second launch per frame.  The node scores ping-pong between two arrays by frame parity (a
// block of the same launch may still have to stage the old ones).  A node none of whose arrivals exceeds -1e20 keeps the reference's triphone 0
// and that triphone's NEW entry time (phone_recog.cpp:190-200), which another block may not have produced yet: it is recomputed here from the
// triphone's state before this frame, which its owner leaves in `inv0_state` every frame (parity again).
template <int NTHR>
__global__ __launch_bounds__(NTHR) void viterbi_fs_fused_kernel(const ViterbiArgs a, int k)
{
    constexpr int CAP = 2048, IT = CAP / (2 * NTHR), HALF = NTHR;   // a thread's triphones: n and n + HALF, IT times
    extern __shared__ float vsm[];                          // [ldl] log-posteriors of frame k, [B] node scores, [CAP] exit scores, [CAP] entry times
    const int slot = blockIdx.y, T = a.T, B = a.B, L = (int)a.ldl;
    const int tri0 = a.blk_tri0[blockIdx.x], tri1 = a.blk_tri0[blockIdx.x + 1];
    const int node0 = a.blk_node0[blockIdx.x], node1 = a.blk_node0[blockIdx.x + 1];
    const long long row = a.offsets[a.slot_utt[slot]] + k;
    const float* lp1 = a.lp + (size_t)row * a.ldl;
    const float* p = ((k & 1) ? a.p2 : a.p) + (size_t)slot * B;
    float* pnext = ((k & 1) ? a.p : a.p2) + (size_t)slot * B;
    i32x4* pp = (i32x4*)a.pp + (size_t)slot * T;
    unsigned short* t2 = (unsigned short*)a.tt + (size_t)slot * T;
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    const f32x4* c1 = (const f32x4*)a.tri_self;
    float* s_lp = vsm;
    float* s_p = vsm + L;
    float* s_po = s_p + B;
    int* s_t2 = (int*)(s_po + CAP);
    const int base = tri0 + threadIdx.x;
    i32x4 ga[IT], gb[IT], qa[IT], qb[IT]; f32x4 ha[IT], hb[IT]; int ta[IT], tb[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int n0 = base + it * 2 * HALF;
        const int n = n0 < tri1 ? n0 : tri1 - 1, m = n0 + HALF < tri1 ? n0 + HALF : tri1 - 1;   // (past the block's end: repeats its last triphone's loads, stores nothing)
        ga[it] = c0[n]; gb[it] = c0[m]; ha[it] = c1[n]; hb[it] = c1[m];
        qa[it] = __builtin_nontemporal_load(pp + n); qb[it] = __builtin_nontemporal_load(pp + m); ta[it] = (int)__builtin_nontemporal_load(t2 + n); tb[it] = (int)__builtin_nontemporal_load(t2 + m);
    }
    // the list bounds and the language-model term of this thread's nodes (8 lanes per node, NTHR / 8 nodes per round): requested here, with everything
    // else, so that the node phase below does not start with a memory round trip of its own (the first two rounds; a block has ~54 nodes at the real size)
    const int l8 = threadIdx.x & 7;
    int ns0[2], ns1[2]; float nlm[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = node0 + r * (NTHR / 8) + (threadIdx.x >> 3);
        const bool valid = i < node1;
        ns0[r] = valid ? a.outOff[i] : 0; ns1[r] = valid ? a.outOff[i + 1] : 0; nlm[r] = valid ? a.lm[i] : 0.0f;
    }
    for (int i = threadIdx.x; i < L; i += NTHR) s_lp[i] = lp1[i];
    for (int i = threadIdx.x; i < B; i += NTHR) s_p[i] = p[i];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int n = base + it * 2 * HALF;
        if (n >= tri1) break;
        const bool two = n + HALF < tri1;
        i32x4 nqa, nqb; int nta, ntb; float poa, pob;
        {
            const i32x4 g = ga[it];
            viterbi_tri_apply(g, ha[it], qa[it], ta[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqa, nta, poa);
        }
        {
            const i32x4 g = gb[it];
            viterbi_tri_apply(g, hb[it], qb[it], tb[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqb, ntb, pob);
        }
        __builtin_nontemporal_store(nqa, pp + n); __builtin_nontemporal_store((unsigned short)nta, t2 + n); s_po[n - tri0] = poa; s_t2[n - tri0] = nta;
        if (two) { __builtin_nontemporal_store(nqb, pp + n + HALF); __builtin_nontemporal_store((unsigned short)ntb, t2 + n + HALF); s_po[n + HALF - tri0] = pob; s_t2[n + HALF - tri0] = ntb; }
        // the owner of the reference's triphone 0 leaves its state for the next frame's unreachable nodes
        if (n == a.inv0 || (two && n + HALF == a.inv0)) {
            const bool first = n == a.inv0;
            int* st = a.inv0_state + (size_t)slot * 16 + ((k + 1) & 1) * 8;
            const i32x4 q = first ? nqa : nqb;
            st[0] = q[0]; st[1] = q[1]; st[2] = q[2]; st[3] = q[3]; st[4] = first ? nta : ntb;
        }
    }
    __syncthreads();
    // node maxima: 8 lanes per node, the first strict maximum in list order (the node kernel's code on the LDS copy)
    for (int i0 = node0, r = 0; i0 < node1; i0 += NTHR / 8, ++r) {
        const int i = i0 + (threadIdx.x >> 3);
        const bool valid = i < node1;
        const int s0 = r == 0 ? ns0[0] : r == 1 ? ns0[1] : (valid ? a.outOff[i] : 0);
        const int s1 = r == 0 ? ns1[0] : r == 1 ? ns1[1] : (valid ? a.outOff[i + 1] : 0);
        const float lmi = r == 0 ? nlm[0] : r == 1 ? nlm[1] : (valid ? a.lm[i] : 0.0f);
        float mx = -1e20f; int ms = 0x7fffffff, mxi = 0;
        for (int s = s0 + l8; s < s1; s += 8) {
            const float pOut = s_po[s - tri0];
            if (mx < pOut) { mx = pOut; ms = s; mxi = s; }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            const float omx = __shfl_xor(mx, off);
            const int oms = __shfl_xor(ms, off), omi = __shfl_xor(mxi, off);
            if (omx > mx || (omx == mx && oms < ms)) { mx = omx; ms = oms; mxi = omi; }
        }
        if (valid && l8 == 0) {
            int bt;
            if (ms == 0x7fffffff) {   // nothing exceeded -1e20: triphone inv0 and its entry time after THIS frame's update
                mxi = a.inv0;
                const int* st = a.inv0_state + (size_t)slot * 16 + (k & 1) * 8;
                const f32x4 h0 = c1[a.inv0];
                const float q1 = __int_as_float(st[1]), q2 = __int_as_float(st[2]);
                const int t1s = (unsigned)st[3] >> 16, t2s = st[4];
                bt = ((q1 + h0[2] > q2 + h0[0]) ? t1s : t2s) - 1;   // viterbi_tri_apply's nt2: go2 = q1 + tr1, st2 = q2 + sl2
            } else {
                bt = s_t2[mxi - tri0] - 1;
            }
            pnext[i] = mx + lmi;
            a.bestI[(size_t)row * B + i] = mxi;
            a.bestT[(size_t)row * B + i] = bt;
        }
    }
}

__global__ __launch_bounds__(64) void viterbi_fs_trace_kernel(const ViterbiArgs a, int n_utts)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_utts) return;
    const int u = a.slot_utt[slot];
    const long long row0 = a.offsets[u];
    const int N = (int)(a.offsets[u + 1] - row0), B = a.B;
    const float* p = ((N & 1) ? a.p2 : a.p) + (size_t)slot * B;   // frame N - 1 wrote the array of the other parity
    const int* bI = a.bestI + (size_t)row0 * B;
    const int* bT = a.bestT + (size_t)row0 * B;
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    int mxi = 0;
    float mx = -1e20f;
    for (int i = 0; i < a.P; ++i) {
        const int ids = a.endIdx[i];
        if (mx < p[ids]) { mx = p[ids]; mxi = ids; }
    }
    int count = 0, t = N - 1;
    bool ok = true;
    while (t >= 0) {
        if (count >= N) { ok = false; break; }           // the reference would loop forever here
        const int id = bI[(size_t)t * B + mxi];
        const int end = t + 1;
        t = bT[(size_t)t * B + mxi];
        a.seg_end[row0 + count] = end;
        a.seg_begin[row0 + count] = t + 1;
        a.seg_tri[row0 + count] = a.perm ? a.outVal[id] : id;
        mxi = c0[id][0] & 0xffff;
        ++count;
    }
    a.seg_count[u] = ok ? count : 0;
}

hipError_t launch_viterbi_fs_init(hipStream_t s, const ViterbiArgs& a, int n_utts)
{
    if (n_utts <= 0) return hipSuccess;
    hipLaunchKernelGGL(viterbi_fs_init_kernel, dim3(32, n_utts), dim3(256), 0, s, a, n_utts);
    return hipGetLastError();
}

hipError_t launch_viterbi_fs_step(hipStream_t s, const ViterbiArgs& a, int k, int n_active)
{
    if (n_active <= 0) return hipSuccess;
    const char* sv = getenv("AM_VITERBI_LDS");       // (measurement: 0 = every launch on the form that gathers from global memory; 1 = staged, node maxima in their own launch)
    const size_t lds = ((size_t)a.ldl + (size_t)a.B) * sizeof(float);
    const char* fm = getenv("AM_VITERBI_FUSED_MIN");
    if ((!sv || atoi(sv) > 1) && a.n_blk > 0 && a.perm && lds + 2 * 2048 * sizeof(float) <= 48 * 1024 && n_active >= (fm ? atoi(fm) : 1)) {   // (to the last utterance: 48.1 ms per real-size pass against 48.6 with the gathering form below 12 active ones)
        // 512 threads x 4 triphones: 50.8 ms per real-size pass; 256 x 8: 52.7; 1,024 x 2: 59.2 (AM_VITERBI_THREADS=256 for A/B timing)
        const char* tv = getenv("AM_VITERBI_THREADS");
        if (tv && atoi(tv) == 256) hipLaunchKernelGGL(viterbi_fs_fused_kernel<256>, dim3(a.n_blk, n_active), dim3(256), lds + 2 * 2048 * sizeof(float), s, a, k);
        else hipLaunchKernelGGL(viterbi_fs_fused_kernel<512>, dim3(a.n_blk, n_active), dim3(512), lds + 2 * 2048 * sizeof(float), s, a, k);
        return hipGetLastError();
    }
    if ((!sv || atoi(sv) != 0) && lds <= 32 * 1024 && n_active >= 12) {
        // trips per thread: 4 (2,048 triphones per staging) while that still gives several blocks per CU, 2 towards the tail of the run
        const int trips = (a.T + 511) / 512;
        if (n_active >= 24) hipLaunchKernelGGL(viterbi_fs_tri_lds_kernel<4>, dim3((trips + 3) / 4, n_active), dim3(256), lds, s, a, k);
        else hipLaunchKernelGGL(viterbi_fs_tri_lds_kernel<2>, dim3((trips + 1) / 2, n_active), dim3(256), lds, s, a, k);
    } else {
        hipLaunchKernelGGL(viterbi_fs_tri_kernel, dim3((a.T + 511) / 512, n_active), dim3(256), 0, s, a, k);
    }
    hipLaunchKernelGGL(viterbi_fs_node_kernel, dim3((a.B + 31) / 32, n_active), dim3(256), 0, s, a, k);
    return hipGetLastError();
}

hipError_t launch_viterbi_fs_trace(hipStream_t s, const ViterbiArgs& a, int n_utts)
{
    if (n_utts <= 0) return hipSuccess;
    hipLaunchKernelGGL(viterbi_fs_trace_kernel, dim3((n_utts + 63) / 64), dim3(64), 0, s, a, n_utts);
    return hipGetLastError();
}

hipError_t launch_viterbi(hipStream_t s, const ViterbiArgs& a, int n_utts)
{
    if (n_utts <= 0) return hipSuccess;
    if (a.compact) hipLaunchKernelGGL(viterbi_compact_kernel, dim3(n_utts), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(viterbi_kernel, dim3(n_utts), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace am
