// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: the LSTM stack and its whole time axis as one persistent launch
#include "am_device.h"

namespace am {

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

hipError_t init_lstm_persist_kernels()
{
    const hipError_t e = hipFuncSetAttribute((const void*)lstm_persist_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_PERSIST_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)lstm_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSTM_PERSIST_LDS);
}

int lstm_persist_mode()
{
    const char* v = getenv("AM_LSTM_PERSIST");   // 0 keeps the per-step kernels (A/B measurements, cross-form parity tests); read at every upload
    return v ? atoi(v) : 1;
}

}  // namespace am
