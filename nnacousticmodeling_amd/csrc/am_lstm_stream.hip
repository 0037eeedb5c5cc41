// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: the LSTM stack and its whole time axis as one persistent launch
// with the weights STREAMED (stacks whose weights do not fit the register file: 1024 units, the reference's published recipe)
#include "am_device.h"

namespace am {

// ---------------------------------------------------------------------------------------------------
// lstm_stream_kernel (round 4): the persistent form for wide stacks.
//
// lstm_persist_kernel keeps a workgroup's 64 gate columns x K of weights in registers; at 1024 units that is 512 KB per workgroup (4 x 1024:
// 134 MB for the stack -- the chip's whole register file), so such stacks ran one launch per (layer, time) diagonal: 651 launches x ~77 us for
// ~38 us of matrix work at the average 89 active rows (BENCH_r03 lstm_4x1024_td5: 50 ms, 0.46 of the fp32-MFMA peak).  Every launch paid a
// boundary, a cold start (the first weight batch and the first activation tile arrive before the first MFMA) and a drain.
// Here the stack runs as ONE launch: workgroup = (layer, 16-unit tile), all of them resident (4 x 64 = 256: one per CU), each walking the time
// axis.  The weights are step-invariant but too large to keep, so they stream: per 128-k batch a wave fetches its 16 gate columns x 64 k
// (16 VGPRs) from L2 / Infinity Cache into a small register ring, two or three batches ahead of the multiply, while the activation rows of
// the batch travel global -> LDS by LDS-DMA (sc1: they were written by other workgroups of this launch) -- lstm_persist_kernel's tile ring.
// Nothing is synchronised but the data flow: layer l's step t needs h_{l-1}[t] (its upward product) and h_l[t-1] (its lateral product), both
// published through one arrival counter per layer (every workgroup of a layer adds 1 per step; sc1 stores, drain, barrier, add: the protocol of
// lstm_persist_kernel, MI355X_MICROARCH.md inter-workgroup visibility, first table row).  A workgroup takes ALL active rows of its units, in
// passes of at most 6 row tiles (the LDS ring holds 96 rows x 128 k per buffer); with > 64 active rows a step is bound by its matrix work
// and the hand-offs vanish behind it (after publishing step t a workgroup multiplies step t+1's upward half for tens of microseconds before it
// needs anybody's step t); with few rows a step is bound by the 512 KB of weights each CU pulls per step.
// Same summation order as every other recurrent kernel (per K half, 64-k chains from zero folded in batch order, upward batches then lateral,
// (half 0 + half 1) + G / bias) and the same gate functions: a frame's posteriors do not depend on the form that computed them.
// Every wait is bounded (LSTM_P_TIMEOUT); a timeout raises abort_flag, every workgroup leaves, the host repeats the pass on the per-step kernels.
// ---------------------------------------------------------------------------------------------------
constexpr int STREAM_TILE_LD = LSTM_KB;                                  // floats per tile row (unpadded, swizzled on the source address)
constexpr size_t LSTM_STREAM_LDS = 3 * 96 * STREAM_TILE_LD * sizeof(float);   // 147,456 B: 3 x 96 rows or 4 x 64 rows; the gate exchange aliases it
constexpr int STREAM_SGLD = 20;

#define AM_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void stream_wait_vmcnt(int k)   // (k is a constant after unrolling: folds to one instruction)
{
    switch (k) {
    AM_VMCNT_CASE(0) AM_VMCNT_CASE(1) AM_VMCNT_CASE(2) AM_VMCNT_CASE(3) AM_VMCNT_CASE(4) AM_VMCNT_CASE(5) AM_VMCNT_CASE(6) AM_VMCNT_CASE(7)
    AM_VMCNT_CASE(8) AM_VMCNT_CASE(9) AM_VMCNT_CASE(10) AM_VMCNT_CASE(11) AM_VMCNT_CASE(12) AM_VMCNT_CASE(13) AM_VMCNT_CASE(14) AM_VMCNT_CASE(15)
    AM_VMCNT_CASE(16) AM_VMCNT_CASE(17) AM_VMCNT_CASE(18) AM_VMCNT_CASE(19) AM_VMCNT_CASE(20) AM_VMCNT_CASE(21) AM_VMCNT_CASE(22) AM_VMCNT_CASE(23)
    AM_VMCNT_CASE(24) AM_VMCNT_CASE(25) AM_VMCNT_CASE(26) AM_VMCNT_CASE(27) AM_VMCNT_CASE(28) AM_VMCNT_CASE(29) AM_VMCNT_CASE(30) AM_VMCNT_CASE(31)
    AM_VMCNT_CASE(32) AM_VMCNT_CASE(33) AM_VMCNT_CASE(34) AM_VMCNT_CASE(35) AM_VMCNT_CASE(36) AM_VMCNT_CASE(37) AM_VMCNT_CASE(38) AM_VMCNT_CASE(39)
    AM_VMCNT_CASE(40) AM_VMCNT_CASE(41) AM_VMCNT_CASE(42)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#undef AM_VMCNT_CASE

// diagnostic timeline (AM_LSTM_STAMPS=1): the ut == 0 workgroup of every layer stamps the 100 MHz counter into a buffer nothing else reads
// (slots: 0 step start, 1 upward wait done, 2 first pass's products done, 4 every pass done, 6 published)
__device__ __forceinline__ void stream_stamp(const LstmPersistArgs& a, int li, int ut, int t, int k)
{
    if (a.stamps && ut == 0 && threadIdx.x == 0) a.stamps[((size_t)li * a.T + t) * 8 + k] = __builtin_amdgcn_s_memrealtime();
}

// One lane polls `*c >= target`; then the workgroup's barrier.  false (all threads) when the pass has been aborted.
__device__ __forceinline__ bool stream_wait(const LstmPersistArgs& a, const unsigned* c, unsigned target, int* s_flag)
{
    if (threadIdx.x == 0) {
        bool ok = true;
        if (ld_agent_u32(c) < target) {
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
    // a barrier that does not drain vector memory: the stages already requested stay in flight (__syncthreads() would wait for them)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (read as an LDS access: through the generic pointer it is a FLAT load, whose wait drains every vector-memory operation in flight)
    return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)AM_LDS_PTR(s_flag)) == 0;
}

// One pass of a step: NT row tiles of this workgroup's 16 units -- the products over NBU upward and NBL lateral 128-k batches, the gate
// arithmetic, the stores of h and c.  Stage b = (activation tile b by LDS-DMA: NT instructions per wave, weight fragments of batch b: 4 loads per
// wave); stages are requested LEAD = NBUF - 1 ahead, always tile first, so the counted wait for stage b is (stages requested behind it) x (NT + 4).
template <int NBU, int NBL, int NT>
__device__ __forceinline__ bool stream_pass(const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int ut, int t, int rows_valid, unsigned r_t, unsigned r_p,
                                            unsigned lat_target, float* ring, int* s_flag, bool first_pass)
{
    constexpr int NBT = NBU + NBL;
    constexpr int ROWS = NT * 16;
    // tile ring: 3 x 96 rows (5-6 tiles) or 4 x 64 rows.  (Measured for 1-2 tiles: 8 x 32 rows, seven stages ahead instead of three, does not help --
    // with row-major weights 23.5 against 22.7 us per thin step, with the weights in stream order 20.2 against 18.2: such a step moves 512 KB of
    // weights per CU, 134 MB chip-wide, now at 8.7 TB/s out of the Infinity Cache; what is left is the per-batch chain, not latency.)
    constexpr int BUF_ROWS = NT <= 4 ? 64 : 96;
    constexpr int NBUF = NT <= 4 ? 4 : 3;
    constexpr int LEAD = NBUF - 1;
    constexpr int BUF_FLOATS = BUF_ROWS * STREAM_TILE_LD;
    constexpr int PER_STAGE = NT + LSTM_KS;            // vector-memory instructions per wave and stage
    static_assert(PER_STAGE * LEAD <= 42, "the counted waits cover at most 42 operations in flight (vmcnt is a 6-bit counter)");
    // (an opaque copy of the thread index per pass: the weight and tile addresses are invariant over the time loop, and hoisted out of it they would
    //  hold ~100 registers for the whole launch)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gate = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int U_pad = a.U_pad, ldg = 4 * U_pad;
    bool ok = true;
    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    if constexpr (NBT > 0) {
        // DMA instruction e of this wave covers tile rows 2 (wave + 8 e) and + 1: lane -> (row, stored chunk); source chunk = stored ^ (row & 15).
        // Rows past the valid ones re-read the last valid row: in the packed layout the rows behind them belong to step t + 1, which another XCD may
        // not have written yet (a line fetched early would stay stale in this XCD's L2).  No lane is masked: every wave issues exactly NT per tile.
        const int drow = lane >> 5, dchunk = lane & 31;
        unsigned doff[NT];
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            const int row = 2 * (wave + 8 * e) + drow;
            const int srow = row < rows_valid ? row : rows_valid - 1;
            doff[e] = (unsigned)srow * (unsigned)U_pad + 4u * (unsigned)(dchunk ^ (row & 15));
        }
        const float* up0 = NBU ? Ly.Hin + (size_t)r_t * U_pad : nullptr;
        const float* lat0 = Ly.H + (size_t)r_p * U_pad;
        // this wave's weight fragments of a batch: its gate's 16 columns x its K half (64 k) = 4 x 16 B per lane.  The weights come in STREAM ORDER
        // (rec_stream_pack_weights: [unit tile][batch][wave][sub-step][lane][4]): a load instruction reads 1 KB of consecutive memory.  From the
        // row-major matrix it touched 16 rows x 64 B -- 64 line look-ups in the vector cache per instruction, 2048 per stage and workgroup, which is
        // what bounded a thin step's weight stream (am_cells_stream.hip, rs_load_w).
        const size_t wlane = (size_t)wave * 1024 + 4 * lane;
        f32x4 wv[NBUF][LSTM_KS];
        auto issue_piece = [&](int b, int e) {   // one 1-KB piece (two rows) of activation tile b
            const float* src = (b < NBU ? up0 + b * LSTM_KB : lat0 + (b - NBU) * LSTM_KB);
            float* buf = ring + (b % NBUF) * BUF_FLOATS;
            __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(src + doff[e]), AM_LDS_PTR(buf + (wave + 8 * e) * 256), 16, 0, 16 /* sc1 */);
        };
        auto issue_weights = [&](int b) {
            const float* wp = (b < NBU ? Ly.Wup + (size_t)(ut * NBU + b) * 8192 : Ly.Wlat + (size_t)(ut * NBL + (b - NBU)) * 8192) + wlane;
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) wv[b % NBUF][s] = *(const f32x4*)(wp + 256 * s);
        };
        auto issue = [&](int b) {
#pragma unroll
            for (int e = 0; e < NT; ++e) issue_piece(b, e);
            issue_weights(b);
        };
        auto lat_wait = [&] { return stream_wait(a, a.counters + li, lat_target, s_flag); };   // step t - 1 of every workgroup of this layer
        // fragment addressing: row r*16 + l15, chunk (kh*16 + 4 s + lq) ^ l15
        const float* fbase = ring + l15 * STREAM_TILE_LD + kh * (LSTM_KB / 2);
        int xs[LSTM_KS];
#pragma unroll
        for (int s = 0; s < LSTM_KS; ++s) xs[s] = 4 * (((4 * s + lq) ^ l15) & 15);
        if (NBL > 0 && NBU < LEAD) ok = lat_wait();                       // a lateral tile is among the first requests
#pragma unroll
        for (int b = 0; b < LEAD; ++b) if (b < NBT) issue(b);
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            const int issued = (b + LEAD < NBT) ? b + LEAD : NBT;         // stages requested so far
            stream_wait_vmcnt(PER_STAGE * (issued - (b + 1)));             // stage b has landed (this wave's part)
            __builtin_amdgcn_s_barrier();                                  // ... everybody's; and everyone has left tile b - 1
            const bool due = b + LEAD < NBT;
            if (due && NBL > 0 && NBU >= LEAD && b + LEAD == NBU) ok = lat_wait() && ok;
            // Few rows: the batch is a chain of latencies, the next stage is requested at once.  Many rows: every wave of the workgroup leaves the
            // barrier at the same moment, and requested right there the NT LDS-DMA instructions of BOTH waves of a SIMD (~100-180 cycles each) find
            // the matrix pipe idle; so the pieces go out between the MFMAs of the batch, K half 0's under sub-steps 0 and 1, K half 1's under 2 and 3.
            constexpr bool SPREAD = NT >= 3;
            if (due && !SPREAD) issue(b + LEAD);
            const float* sa = fbase + (b % NBUF) * BUF_FLOATS;
            f32x4 pb[NT];
#pragma unroll
            for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < LSTM_KS; ++s) {
                f32x4 af[NT];
#pragma unroll
                for (int r = 0; r < NT; ++r) af[r] = *(const f32x4*)(sa + r * 16 * STREAM_TILE_LD + xs[s]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        if constexpr (NT == 1) {
                            // One row tile = ONE chain of 16 dependent MFMAs per batch, and that chain must accumulate IN PLACE (vDst = SrcC): written as
                            // the builtin, hipcc (ROCm 7.2) gave the fourth MFMA of a sub-step a destination other than its SrcC and no wait states in
                            // front of it -- v_mfma_f32_16x16x4_f32 v[10:13], v13, v17, v[38:41] right behind the MFMA that writes v[38:41] -- and rows
                            // 8..15 of the product came out wrong (tools/lstm_forms_diff.py; every multi-tile instantiation interleaves independent chains and
                            // is not affected).  In place the hardware forwards the running sum from one MFMA to the next.
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(pb[0]) : "v"(af[0][j]), "v"(wv[b % NBUF][s][j]));
                        } else
                        pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j], wv[b % NBUF][s][j], pb[r], 0, 0, 0);
                    }
                if (due && SPREAD && (s >> 1) == kh) {   // (wave-uniform)
#pragma unroll
                    for (int e = 0; e < NT; ++e) if ((2 * e) / NT == (s & 1)) issue_piece(b + LEAD, e);
                    if ((s & 1) == 1) issue_weights(b + LEAD);
                }
            }
            if constexpr (NT == 1) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the chain's last result must have left the matrix pipe before a vector instruction reads it
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                acc[r] += pb[r];
                // (pinned here: left to itself the scheduler defers these adds to the end of the pass and spills the chains' partial sums meanwhile)
                asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]));
            }
        }
        __syncthreads();   // every fragment read is done: the ring may be rewritten (gate exchange, next pass)
        if (!ok) return false;
    }
    if (first_pass) stream_stamp(a, li, ut, t, 2);

    // gates.  thread = (row, 4 units): the four gates' upward terms (G rows of the stack's first layer, the bias otherwise), previous cell state
    float* sg = ring;   // [kh][gate][ROWS][STREAM_SGLD]
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + gate) * ROWS + r * 16 + lq * 4 + reg) * STREAM_SGLD + l15] = acc[r][reg];
    const int erow = tid >> 2, quad = tid & 3;
    const bool live = erow < ROWS && erow < rows_valid;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.Cst, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)Ly.H, 0, 0x7fffffff, 0x00020000);
    f32x4 gq[4], cq = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gq[k] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        if (live) gq[k] = NBU ? *(const f32x4*)(Ly.bias + ut * 64 + k * 16 + 4 * quad)
                              : *(const f32x4*)(Ly.G + (size_t)(r_t + erow) * ldg + ut * 64 + k * 16 + 4 * quad);
    }
    if (live && NBL > 0) cq = ld_sc1_b128(crs, ((r_p + erow) * (unsigned)U_pad + (unsigned)(ut * 16 + 4 * quad)) * 4u);
    __syncthreads();
    if (live) {
        f32x4 pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            pre[k] = (*(const f32x4*)(sg + ((0 * 4 + k) * ROWS + erow) * STREAM_SGLD + 4 * quad) + *(const f32x4*)(sg + ((1 * 4 + k) * ROWS + erow) * STREAM_SGLD + 4 * quad)) + gq[k];
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
    __syncthreads();   // the gate exchange has been read: the next pass may refill the ring
    return true;
}

template <int NBU, int NBL>
__device__ __forceinline__ bool stream_pass_nt(int nt, const LstmPersistArgs& a, const LstmPersistLayer& Ly, int li, int ut, int t, int rows_valid, unsigned r_t, unsigned r_p,
                                               unsigned lat_target, float* ring, int* s_flag, bool first_pass)
{
    switch (nt) {   // uniform per workgroup
    case 1: return stream_pass<NBU, NBL, 1>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    case 2: return stream_pass<NBU, NBL, 2>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    case 3: return stream_pass<NBU, NBL, 3>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    case 4: return stream_pass<NBU, NBL, 4>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    case 5: return stream_pass<NBU, NBL, 5>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    case 6: return stream_pass<NBU, NBL, 6>(a, Ly, li, ut, t, rows_valid, r_t, r_p, lat_target, ring, s_flag, first_pass);
    default: return true;
    }
}

template <int NB>
__global__ __launch_bounds__(512, 2) void lstm_stream_kernel(const LstmPersistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float stream_smem[];
    __shared__ int s_flag[2];
    if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    // role: workgroups b, b + n_layers, ... belong to one layer -- with 4 layers those sit on two XCDs under round-robin placement and share the
    // layer's activation rows in their L2s (speed only)
    const int li = blockIdx.x % a.n_layers, ut = blockIdx.x / a.n_layers;
    const LstmPersistLayer& Ly = a.L[li];
    const unsigned per_layer = (unsigned)(a.U_pad / 16);        // arrivals a layer's counter receives per step
    const bool upward = li > 0 || a.ext_upward;                 // multiplies an input plane (else: adds the hoisted rows G)
    const int t_end = a.t_end > 0 ? a.t_end : a.T;
    for (int t = a.t_begin; t < t_end; ++t) {
        int n = a.n_active[t] - a.slot_base;
        n = n < 0 ? 0 : (n > a.slot_cap ? a.slot_cap : n);
        if (n <= 0) break;
        const unsigned off_t = (unsigned)a.step_off[t], off_p = t > 0 ? (unsigned)a.step_off[t - 1] : 0u;
        stream_stamp(a, li, ut, t, 0);
        // the layer below has finished step t (it runs ahead: normally no wait at all)
        if (li > 0 && !stream_wait(a, a.counters + (li - 1), per_layer * (unsigned)(t + 1), s_flag)) return;
        stream_stamp(a, li, ut, t, 1);
        const int tiles = (n + 15) >> 4, npass = (tiles + 5) / 6, q = (tiles + npass - 1) / npass;
        for (int p = 0; p < npass; ++p) {
            const int row0 = p * q * 16;
            int nt = tiles - p * q;
            nt = nt > q ? q : nt;
            if (nt <= 0) break;
            const unsigned r_t = off_t + (unsigned)(a.slot_base + row0), r_p = off_p + (unsigned)(a.slot_base + row0);
            const unsigned lat_target = per_layer * (unsigned)t;
            bool ok;
            if (upward) ok = t > 0 ? stream_pass_nt<NB, NB>(nt, a, Ly, li, ut, t, n - row0, r_t, r_p, lat_target, stream_smem, s_flag, p == 0)
                                   : stream_pass_nt<NB, 0>(nt, a, Ly, li, ut, t, n - row0, r_t, r_p, lat_target, stream_smem, s_flag, p == 0);
            else ok = t > 0 ? stream_pass_nt<0, NB>(nt, a, Ly, li, ut, t, n - row0, r_t, r_p, lat_target, stream_smem, s_flag, p == 0)
                            : stream_pass_nt<0, 0>(nt, a, Ly, li, ut, t, n - row0, r_t, r_p, lat_target, stream_smem, s_flag, p == 0);
            if (!ok) return;
        }
        stream_stamp(a, li, ut, t, 4);
        // publish: every storing wave drains its stores, the workgroup meets, ONE lane signals for all of them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counters + li, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stream_stamp(a, li, ut, t, 6);
    }
}

// shapes with a streamed form: U_pad / 128 batches per product
static const void* lstm_stream_fn(int U_pad)
{
    switch (U_pad) {
    case 768: return (const void*)lstm_stream_kernel<6>;
    case 1024: return (const void*)lstm_stream_kernel<8>;
    default: return nullptr;
    }
}

hipError_t init_lstm_stream_kernels()
{
    for (int U_pad : {768, 1024}) {
        const hipError_t e = hipFuncSetAttribute(lstm_stream_fn(U_pad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_STREAM_LDS);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Workgroups of the streamed form the device holds at once (one per CU: 147 KB of LDS each), 0 when the shape has none or this build of the kernel
// uses scratch memory (see lstm_persist_scratch_free: the first launch of a scratch-using kernel does not get all its workgroups resident).
int lstm_stream_resident_wgs(int U_pad, int n_cus)
{
    const void* fn = lstm_stream_fn(U_pad);
    if (!fn) return 0;
    static int scratch_free[2] = {-1, -1};
    int& st = scratch_free[U_pad == 1024 ? 0 : 1];
    if (st < 0) {
        hipFuncAttributes at{};
        const hipError_t e = hipFuncGetAttributes(&at, fn);
        st = (e == hipSuccess && at.localSizeBytes == 0) ? 1 : 0;
        if (!st) fprintf(stderr, "libamfwd: the streamed recurrent kernel of this build uses scratch memory (%zu B per lane); staying on the per-step kernels\n", (size_t)at.localSizeBytes);
    }
    if (!st) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, LSTM_STREAM_LDS) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu > 0 ? n_cus : 0;
}

// Layers one launch can take (every workgroup must be resident): 0 when the shape has no streamed form.
int lstm_stream_layers(int U_pad, int max_wgs)
{
    if (!lstm_stream_fn(U_pad) || max_wgs <= 0) return 0;
    const int l = max_wgs / (U_pad / 16);
    return l > LSTM_MAX_LAYERS ? LSTM_MAX_LAYERS : l;
}

hipError_t launch_lstm_stream(hipStream_t s, const LstmPersistArgs& a)
{
    const void* fn = lstm_stream_fn(a.U_pad);
    if (!fn || a.n_layers < 1 || a.n_layers > LSTM_MAX_LAYERS) return hipErrorInvalidValue;
    const int grid = a.n_layers * (a.U_pad / 16);
    if (grid > a.max_wgs) return hipErrorInvalidValue;   // every workgroup must be resident: never more than the device holds
    if (a.t_begin < 0 || a.t_begin >= a.T || a.t_end < 0 || a.t_end > a.T || (a.t_end > 0 && a.t_end <= a.t_begin)) return hipErrorInvalidValue;
    if (a.U_pad == 1024) hipLaunchKernelGGL(lstm_stream_kernel<8>, dim3(grid), dim3(512), LSTM_STREAM_LDS, s, a);
    else hipLaunchKernelGGL(lstm_stream_kernel<6>, dim3(grid), dim3(512), LSTM_STREAM_LDS, s, a);
    return hipGetLastError();
}

}  // namespace am
