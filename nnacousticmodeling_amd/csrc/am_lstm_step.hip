// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: the LSTM stack as one launch per (layer, time) diagonal
#include "am_device.h"

namespace am {

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
// (LSTM_RB = 64 rows per block, LSTM_KB = 128 k per batch, LSTM_LDA: am_device.h)

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


constexpr size_t LSTM_THIN_LDS_MAX = 72 * 1024;   // two blocks per CU stay co-resident
static int lstm_wreg_mode()
{
    static int mode = -1;   // AM_LSTM_WREG=0 keeps the streaming kernel (A/B measurements)
    if (mode < 0) { const char* v = getenv("AM_LSTM_WREG"); mode = v ? atoi(v) : 1; }
    return mode;
}

hipError_t init_lstm_persist_kernels();   // am_lstm_persist.hip
hipError_t init_lstm_stream_kernels();    // am_lstm_stream.hip

hipError_t init_lstm_kernels()
{
    hipError_t we = hipFuncSetAttribute((const void*)lstm_wreg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_WREG_LDS);
    if (we != hipSuccess) return we;
    we = init_lstm_persist_kernels();
    if (we != hipSuccess) return we;
    we = init_lstm_stream_kernels();
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

}  // namespace am
