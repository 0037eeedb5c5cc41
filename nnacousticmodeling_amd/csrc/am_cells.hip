// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: GRU / MGRU / peephole-LSTM phase kernels
#include "am_device.h"

namespace am {

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
    float o0, o1 = 0.0f;
    rec_cell_math<G, EPI>(pre, e0, e1, first, act, o0, o1);   // (am_device.h: shared with the persistent form)
    Ls.O0[eidx] = o0;
    if (EPI == REC_GRU_A || EPI == REC_PEEP_A) Ls.O1[eidx] = o1;
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
    // (pad units -- their weights and bias are zero -- are never stored: their planes keep the zeros of the upload.  Computed, an infinite input value would
    //  make them inf x 0 = NaN, which the next product spreads into every real unit; see gemm_epilogue)
    const bool elive = tid < ITEMS * R && er < ntile && eslot < n_active && tile * NU + euu < args.U_real;
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
    // the same fragments out of the stream-ordered copy (rec_stream_pack_weights: a 64-k chain of a 16-column tile is the fragment of (64-column block
    // tile / 4, column group tile % 4, 128-k batch chain / 2, K half chain % 2)): 1 KB of consecutive memory per load instruction, not 16 rows x 64 B
    const float* wso = Ls.Ws ? Ls.Ws + ((size_t)(tile >> 2) * (Ls.ldw / LSTM_KB) * 8 + (tile & 3)) * 1024 + 4 * lane : nullptr;
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
                const int ch = ch0 + (s >> 2);
                wv[s] = wso ? *(const f32x4*)(wso + ((size_t)(ch >> 1) * 8 + (ch & 1) * 4) * 1024 + (s & 3) * 256) : *(const f32x4*)(wrow + k0);
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
__device__ __forceinline__ void rec_wide_body(const RecLayerStep& Ls, int U_pad, int U_real, int act, int bt, int row0, float* smem)
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
        // the weights in stream order ([column block][batch][wave][sub-step][lane][4], rec_stream_pack_weights): a load instruction reads 1 KB of
        // consecutive memory -- from the row-major matrix it touched 16 rows x 64 B, 64 line look-ups in the vector cache (am_cells_stream.hip, rs_load_w)
        const float* wbase = Ls.Ws + ((size_t)bt * (Ls.ldw / LSTM_KB) * 8 + wave) * 1024 + 4 * lane;
        auto load_w = [&](int b, f32x4 (&dst)[LSTM_KS]) {
            const bool real = b < nb;
            const float* wp = wbase + (size_t)(real ? b : nb - 1) * 8192;
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) {
                const f32x4 v = *(const f32x4*)(wp + 256 * s);
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
        if (erow < ROWS && slot < n_active && bt * UB + ub < U_real) {   // (pad units are never stored: rec_step_kernel)
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
        case 1: rec_wide_body<G, EPI, 1, true>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 2: rec_wide_body<G, EPI, 2, true>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 3: rec_wide_body<G, EPI, 3, true>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 4: rec_wide_body<G, EPI, 4, true>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        default: break;
        }
    } else {
        switch (ntile) {
        case 1: rec_wide_body<G, EPI, 1, false>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 2: rec_wide_body<G, EPI, 2, false>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 3: rec_wide_body<G, EPI, 3, false>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
        case 4: rec_wide_body<G, EPI, 4, false>(Ls, U_pad, args.U_real, args.act, bt, row0, rec_wide_smem); break;
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

hipError_t init_rec_stream_kernel();   // am_cells_stream.hip

hipError_t init_rec_kernels()
{
    hipError_t e;
    if ((e = init_rec_stream_kernel()) != hipSuccess) return e;
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
    static const int wide_min = getenv("AM_REC_WIDE_MIN_TILES") ? atoi(getenv("AM_REC_WIDE_MIN_TILES")) : 3;
    bool wide = rec_wide_mode() != 0 && max_tiles >= wide_min;   // <= 32 rows: the K-split form above is the faster one (re-measured with the weights
                                                                 // in stream order, 3 x 512 GRU: from 3 tiles 17.1 ms, from 2 tiles 17.3, always 18.4)
    for (int l = 0; l < a.n; ++l)
        if (a.L[l].K[0] % LSTM_KB || a.L[l].K[1] % LSTM_KB || a.L[l].K[2] % LSTM_KB || a.L[l].ldw % LSTM_KB || !a.L[l].Ws) wide = false;
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

}  // namespace am
