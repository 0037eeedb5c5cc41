// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: Linear layers (one launch per layer, and the whole stack as one work-queue launch)
#include "am_device.h"

namespace am {

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

// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// MODE: 0 relu, 1 sigmoid, 2 tanh, 3 linear, 4 scale by alpha, 5 accumulate into C.
// Pad columns (col >= n_real; gate-tiled layouts: unit 16 (col / 64) + col % 16 >= tiled_units) of an activation / linear output are forced to 0.
// Their weights and bias are zero, so they ARE zero for finite inputs -- but an infinite input value makes them inf x 0 = NaN, and the next layer would
// multiply that NaN by its own zero pad weights into every real column, where the reference (which has no padding) stays finite (sigmoid / tanh
// layers, LSTM gates saturate on an infinite pre-activation).  Found by the round-4 fuzz (tests/fuzz_gpu.py, tools/nonfinite_probe.py).
template <int MODE, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[MI][NI], const float* __restrict__ bias, float* __restrict__ C,
                                              int ldc, int erow, int ecol, float alpha, int n_real, int tiled_units)
{
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = ecol + ni * 32;
            const float bv = bias[col];
            const bool padc = MODE <= 3 && (tiled_units > 0 ? 16 * (col >> 6) + (col & 15) >= tiled_units : col >= n_real);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = erow + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                float v = acc[mi][ni][reg] + bv;
                float* dst = C + (size_t)row * ldc + col;
                if (MODE == 0) v = am_relu(v);
                else if (MODE == 1) v = am_sigmoid(v);
                else if (MODE == 2) v = tanhf(v);
                else if (MODE == 4) v = v * alpha;
                else if (MODE == 5) v = *dst + v;
                if (padc) v = 0.0f;
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
                                                   float* __restrict__ C, int ldc, int m0, int n0, int act, int epi, float alpha, int n_real, int tiled_units)
{
    constexpr int MI = GEMM_BM / GEMM_WM / 32, NI = BN / GEMM_WN / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / GEMM_WN, wn = wave % GEMM_WN, li = lane & 31, lh = lane >> 5;
    // one uniform dispatch on the mode, then straight-line code
    const int erow = m0 + wm * (GEMM_BM / GEMM_WM) + 4 * lh, ecol = n0 + wn * (BN / GEMM_WN) + li;
    if (epi == EPI_ACT) {
        if (act == 0) gemm_epilogue<0, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
        else if (act == 1) gemm_epilogue<1, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
        else gemm_epilogue<2, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
    } else if (epi == EPI_LINEAR) gemm_epilogue<3, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
    else if (epi == EPI_SCALE) gemm_epilogue<4, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
    else gemm_epilogue<5, MI, NI>(acc, bias, C, ldc, erow, ecol, alpha, n_real, tiled_units);
}

template <int BN, int BK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_f32_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    float* __restrict__ C, int ldc, int tiles_n, int nk, int act, int epi, float alpha, int row_p, int row_skip, int* __restrict__ queue, int n_tiles,
    int n_real, int tiled_units)
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
    gemm_tile_epilogue<BN>(acc, bias, C, ldc, m0, n0, act, epi, alpha, n_real, tiled_units);
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
hipError_t init_viterbi_kernels();

hipError_t init_kernels()
{
    hipError_t se = init_split_kernels();
    if (se != hipSuccess) return se;
    se = init_lstm_kernels();
    if (se != hipSuccess) return se;
    se = init_rec_kernels();
    if (se != hipSuccess) return se;
    se = init_viterbi_kernels();
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
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 32, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles, g.n_real, g.tiled_units);
    } else if (g.N_pad % 192 == 0 && !getenv("AM_GEMM_384")) {
        // 1909 classes -> 1920 = 10 x 192: half the tile of the 384-wide form (the round-1 choice, kept behind AM_GEMM_384 for A/B runs) --
        // 96 accumulator registers instead of 192, no spills, and twice as many tiles to fill the last round of a launch with
        const int tiles_n = g.N_pad / 192, n_tiles = tiles_m * tiles_n;
        const int grid = g.queue ? std::min(n_tiles, g.n_cus) : n_tiles;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<192, 32>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_192, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 32, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles, g.n_real, g.tiled_units);
    } else if (g.N_pad % 384 == 0) {
        const int tiles_n = g.N_pad / 384, n_tiles = tiles_m * tiles_n;
        const int grid = g.queue ? std::min(n_tiles, 2 * g.n_cus) : n_tiles;
        hipLaunchKernelGGL((gemm_nt_f32_kernel<384, 16>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_384, s,
                           g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, tiles_n, g.K_pad / 16, g.act, g.epi, g.alpha, g.row_p, g.row_skip, g.queue, n_tiles, g.n_real, g.tiled_units);
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
// (No pad-column mask here -- it costs this kernel 9 registers it does not have (256 + scratch): the host only takes this form for hidden widths that
//  need no padding, ff_stack_shape_ok; padded widths run one launch per layer, whose epilogue masks: gemm_epilogue.)
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
                    if (MODE == 0) x = am_relu(x);
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
                gemm_tile_epilogue<BN>(acc, Ly.bias, Ly.C, Ly.ldc, m0, n0, a.act, a.out_epi, a.out_alpha, 1 << 30, 0);   // plain stores: read by later launches only
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

}  // namespace am
