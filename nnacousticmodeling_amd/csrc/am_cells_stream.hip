// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: GRU / MGRU / peephole-LSTM stacks and their whole time axis as ONE persistent
// launch (round 4)
#include "am_device.h"

namespace am {

// ---------------------------------------------------------------------------------------------------
// rec_stream_kernel: the persistent form of the cells other than L.LSTM.
//
// The per-tick form (rec_wide_kernel / rec_step_kernel) runs one launch per phase and (layer, time) diagonal: 1,300 launches for a 3 x 512 GRU pass,
// each re-fetching the whole weight set and paying a boundary and a cold start -- 20.3 + 15.8 us per tick against 13.7 us of matrix work.
// Here the stack runs as one launch of resident workgroups that walk the time axis, with lstm_stream_kernel's machinery: weights streamed per 128-k
// batch into a register ring, activation rows by LDS-DMA into a swizzled tile ring, counted vmcnt waits, one raw barrier per batch, sc1 stores + one
// arrival counter add per workgroup and step, bounded waits.
//
// A ROLE is one phase of one layer (rec_epilogue's five phases: MGRU; GRU A / B; peephole A / B): a packed weight matrix [G U_pad, sum K], up to three
// K segments (x | h[t-1] | c[t-1], or x | r.h, or c'), an element-wise epilogue.  A workgroup = (role, 64 packed columns, row group).  ROW GROUPS fill the
// chip: the 16-row tiles of a step are dealt round-robin to NG groups (tile j -> group j % NG), so every group keeps a share while utterances end, and --
// the map being static -- every dependency stays inside a group: role R of group g at step t needs rows that the producing role's workgroups OF THE SAME
// GROUP wrote, and since the active tiles only ever decrease, "group g has finished step t" is simply counter[R][g] >= columns(R) x (t + 1).  No host table.
// All handed-off planes (h, and per layer two more: GRU Z and r.h, peephole c and the o pre-activation) are TIME-MAJOR and written once per launch: a
// line rewritten in place could be served stale from another XCD's L2.
// Data flow of a step, e.g. GRU layer l: role A (r, z) multiplies x = h_{l-1}[t] as soon as the layer below has published it -- normally long before --,
// waits for h_l[t-1], multiplies it, stores Z and r.h, publishes; role B (candidate) has multiplied x meanwhile, waits for A, multiplies r.h, stores
// h_l[t], publishes.  The chain of a step is two half-products and two hand-offs instead of two launches.
// Same summation order (64-k chains from zero per K half, folded in k order, (even + odd) + bias) and the same cell arithmetic (rec_cell_math) as the
// per-tick kernels: bit-equal to them.  Pad units are never stored (their planes keep the upload's zeros: see gemm_epilogue).
// ---------------------------------------------------------------------------------------------------
constexpr int RS_TILE_LD = LSTM_KB;
// Stage ring: 4 x 64 rows with three or four row tiles; 8 x 32 rows with one or two.  A thin step is a chain of latencies (the timeline: ~1 us per
// 128-k batch whatever the row count, i.e. the ~3 us round trip of the write-through rows divided by the three stages in flight): with seven
// stages in flight every load of a 512-unit step's product (4 + 4 batches) is requested before the first one is consumed.
constexpr size_t REC_STREAM_LDS = (size_t)4 * 64 * RS_TILE_LD * sizeof(float);   // 131,072 B; the K-half exchange aliases it
constexpr int RS_SGLD = 17;

// The role table, n_active and step_off are written before the launch and never during it: read through the constant address space they are scalar
// loads into SGPRs.  Read through the generic pointers of the argument block they became VECTOR loads (the compiler cannot prove a global location
// unchanged across the asm barriers) and every pointer taken from them a FLAT access -- whose `s_waitcnt vmcnt(0)` in front of each stage's issue
// drained the whole prefetch ring: no stage was ever in flight behind the one being multiplied (measured: 24 % slower than the per-tick kernels).
#define AM_CONST_AS __attribute__((address_space(4)))
#define AM_GLOBAL_AS __attribute__((address_space(1)))
typedef const AM_CONST_AS RecStreamRole* RsRolePtr;
typedef const AM_CONST_AS RecStreamSeg* RsSegPtr;
// (rs_wait's seen values travel as four separate scalars: as an array or a struct, the select chain over them is folded back into a runtime index -> scratch)

// `later` stages of PS vector-memory instructions each were requested behind the one that must have landed
template <int PS, int LEAD>
__device__ __forceinline__ void rs_wait_stages(int later)
{
    static_assert(PS * (LEAD - 1) <= 63, "vmcnt is a 6-bit counter");
#define RS_WAIT_CASE(K) case K: if constexpr (K < LEAD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PS * K) : "memory"); break;
    switch (later) {   // (uniform)
    RS_WAIT_CASE(1) RS_WAIT_CASE(2) RS_WAIT_CASE(3) RS_WAIT_CASE(4) RS_WAIT_CASE(5) RS_WAIT_CASE(6)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef RS_WAIT_CASE
}

// The four weight loads of a stage.  The weights come in STREAM ORDER (rec_stream_pack_weights): the 16 B lane `lane` of wave `wave` needs for
// sub-step q of batch b of column workgroup bt sit at float index ((((bt NB + b) 8 + wave) 4 + q) 64 + lane) 4 -- one load instruction reads 1 KB
// of consecutive memory (8 cache lines).  Read from the row-major matrix the same instruction touched 16 rows x 64 B: 64 line look-ups in the vector
// cache per instruction, 2048 per stage and workgroup -- the timeline's fixed ~1 us per 128-k batch, whatever the row count and however many stages
// were in flight (L2 hit rate 85 %, 3 TB/s delivered: neither HBM nor the L2 was the limit).
// Issued where the compiler does not see them as memory operations: the batch loop has a runtime trip count, and for loads it tracks across the
// back edge the compiler puts `s_waitcnt vmcnt(0)` in front of the first MFMA that reads them -- with the tile pieces and the weights of the stages
// behind still in flight, i.e. no prefetch at all.  The counted waits are rs_wait_stages'; rs_pin_w, placed after them, is the point from which the
// values may be read.
__device__ __forceinline__ void rs_load_w(f32x4 (&wv)[LSTM_KS], const float* batch /* uniform */, unsigned tid16)
{
    static_assert(LSTM_KS == 4, "four 16-B loads per lane and stage");
    asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                 "global_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]) : "v"(tid16), "s"(batch) : "memory");
}
// (the accumulator tiles of a split phase travel the same way: 16 B per lane, L2-served)
__device__ __forceinline__ void rs_load_acc(f32x4& v, const float* tile /* uniform */, unsigned tid16)
{
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(v) : "v"(tid16), "s"(tile) : "memory");
}
template <int NT>
__device__ __forceinline__ void rs_pin_acc(f32x4 (&acc)[NT])
{
#pragma unroll
    for (int r = 0; r < NT; ++r) asm volatile("" : "+v"(acc[r]) :: "memory");
}
__device__ __forceinline__ void rs_pin_w(f32x4 (&wv)[LSTM_KS])
{
    asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]) :: "memory");
}

// One lane polls `*c >= target`; then a barrier that does not drain vector memory.  false (all threads) when the pass has been aborted.
// `seen` (a wave-uniform register per dependency of this workgroup's role) keeps the last value the poll returned: counters only grow, and a producer
// that runs ahead (the layer below, normally by many steps) leaves it far beyond the next targets -- the callers skip those waits altogether.
// (Everything published before a counter value was reached is visible once that value has been seen: later targets included.)  The value travels from
// the polling lane to the other waves through s_seen[slot], read AFTER the barrier: every wave of the workgroup then holds the same number and takes the
// same decision at the next wait.  (Deciding from the LDS word BEFORE the barrier did not: a wave that arrived after lane 0 had stored the new value
// skipped the barrier the earlier waves were waiting at -- measured as wrong outputs, 0.11 off.)  Two waits on the same slot are always separated by
// another workgroup barrier (a slot is waited on once per pass), so the next store cannot overtake a late reader.
__device__ __forceinline__ bool rs_wait(const RecStreamArgs& a, const unsigned* c, unsigned target, int* s_flag, int slot, unsigned& seen, unsigned long long* srow)
{
    if (threadIdx.x == 0) {
        bool ok = true;
        const unsigned long long ts = srow ? __builtin_amdgcn_s_memrealtime() : 0ull;
        unsigned v = ld_agent_u32(c);
        if (v < target) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int it = 0;
            while ((v = ld_agent_u32(c)) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++it & 63) == 0 && (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                        __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)LSTM_P_TIMEOUT)) { ok = false; break; }
            }
        }
        if (srow && !(a.dbg & 128)) { const unsigned long long te = __builtin_amdgcn_s_memrealtime(); srow[1 + (slot > 2 ? 2 : slot)] += te - ts; if (slot == 1) srow[7] = te; }
        if (ok) s_flag[4 + slot] = (int)v;
        else { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_flag = 1; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    seen = (unsigned)__builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)AM_LDS_PTR(s_flag + 4 + slot));
    return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)AM_LDS_PTR(s_flag)) == 0;
}

#define RS_DUMP(v) do { if (srow && threadIdx.x == 0 && (a.dbg & 128)) { \
        srow[2] = ((unsigned long long)__builtin_bit_cast(unsigned, (v)[1]) << 32) | __builtin_bit_cast(unsigned, (v)[0]); \
        srow[3] = ((unsigned long long)__builtin_bit_cast(unsigned, (v)[3]) << 32) | __builtin_bit_cast(unsigned, (v)[2]); } } while (0)

// One pass of a step of one role: NT row tiles (this group's tiles i0 .. i0 + NT - 1, i.e. slots 16 (g + NG i)..) x this workgroup's 64 packed columns.
template <int G, int EPI, int NT>
__device__ __forceinline__ bool rs_pass(const RecStreamArgs& a, RsRolePtr R, int g, int bt, int t, int i0, int n, int toff, unsigned off_t, unsigned off_p,
                                        float* ring, int* s_flag, unsigned& sn0, unsigned& sn1, unsigned& sn2, unsigned& sn3, unsigned& sn4, unsigned long long* srow)
{
    constexpr int ROWS = NT * 16;
    constexpr int RS_NBUF = NT <= 2 ? 8 : 4, RS_LEAD = RS_NBUF - 1, RS_BUF_FLOATS = (NT <= 2 ? 32 : 64) * RS_TILE_LD;
    constexpr int PS = NT + LSTM_KS;                 // vector-memory instructions per wave and stage: NT tile pieces + 4 weight loads
    constexpr int NU = 16 / G, UB = 4 * NU;          // units per 16-column tile / per workgroup
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                    // (addresses derived from it are invariant over the time loop: not to be hoisted out of it and held)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    const int NG = a.NG, U_pad = a.U_pad;
    const bool first = t == 0;
    bool ok = true;

    // the K segments of this step: those of step t - 1 do not exist at t == 0.  (Scalars, selected by comparison: a runtime-indexed array would live in scratch.)
    const int s0nb = R->seg[0].nb, s1nb = R->seg[1].nb, s2nb = R->seg[2].nb;
    const int nb0 = (s0nb > 0 && !(first && R->seg[0].skip_first)) ? s0nb : 0, nb1 = (s1nb > 0 && !(first && R->seg[1].skip_first)) ? s1nb : 0,
              nb2 = (s2nb > 0 && !(first && R->seg[2].skip_first)) ? s2nb : 0;
    const int nbt = nb0 + nb1 + nb2;
    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned tid16 = 16u * (unsigned)(tid & 63) + 4096u * (unsigned)(tid >> 6);   // this lane's 16 B within a 32-KB weight batch: 4 KB per wave (4 sub-steps x 1 KB)
    const unsigned acc16 = 16u * (unsigned)tid;                                         // ... and within an 8-KB accumulator tile of a split phase: 1 KB per wave
    // The state part of a split phase: the accumulators start from what the input part's role left.  The loads go out first -- older than every stage
    // load, so the first counted wait below covers them -- and by inline asm, like the weights; a register with a load in flight must not be READ by
    // anything before that wait, and a copy counts: with the loads under `if (init)` the compiler merged their destination with the zeros of the
    // other path through a copy it was free to place ahead of the wait (seen in the ISA; 3 x 512 GRU wrong).  So every role issues them, unconditionally
    // and straight into `ai` (a role without such a plane reads its own weights' first bytes and drops them), and the choice is made behind the wait.
    const float* init = R->init;
    if (init) {
        const unsigned tg = R->idep_mult * (unsigned)(t + 1);
        if (sn4 < tg || (a.dbg & 1)) { if (!rs_wait(a, a.counters + R->idep * NG + g, tg, s_flag, 4, sn4, srow)) return false; }
    }
    // (a wait the compiler can see: nothing is in flight here but a multi-pass step's last stores, and without it the compiler's own bookkeeping of the
    //  epilogue's loads and stores puts an `s_waitcnt vmcnt(0)` behind these loads -- their whole round trip, every step)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    f32x4 ai[NT];
    {
        const size_t tile0 = (size_t)(toff + g + NG * i0), tstep = init ? (size_t)NG : 0;
        const float* src = init ? init : R->W;
        const size_t per_tile = init ? (size_t)R->n_col * (8 * 256) : 0, col = init ? (size_t)bt * (8 * 256) : 0;
#pragma unroll
        for (int r = 0; r < NT; ++r) rs_load_acc(ai[r], src + (tile0 + tstep * r) * per_tile + col, acc16);
    }
    if (nbt == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rs_pin_acc<NT>(ai);
#pragma unroll
        for (int r = 0; r < NT; ++r) acc[r] = init ? ai[r] : acc[r];
        RS_DUMP(ai[0]);
    }

    if (nbt > 0) {
        // tile piece e of this wave = rows 2 wave, 2 wave + 1 of row tile e (slots 16 (g + NG (i0 + e)) + row); rows past the active ones repeat the
        // last active row: the rows behind them belong to step t + 1 and may not have been written yet.  No lane is masked: NT pieces per wave and tile.
        const int drow = lane >> 5, dchunk = lane & 31;
        unsigned drw[NT];                                  // packed row (without the step offset) of this lane's piece row, per tile
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            const int slot = 16 * (g + NG * (i0 + e)) + 2 * wave + drow;
            drw[e] = (unsigned)(slot < n ? slot : n - 1);
        }
        const unsigned dsw = 4u * (unsigned)(dchunk ^ ((2 * wave + drow) & 15));
        const float* wbase = R->W + (size_t)bt * (R->ldw / LSTM_KB) * (8 * 1024);   // this column workgroup's batches, 32 KB each (stream order: rs_load_w)
        f32x4 wv[RS_NBUF][LSTM_KS];
        // stage b -> (segment, batch in the segment): at most three segments, walked by comparison (plain value-returning helpers: a lambda that
        // writes captured variables from inside the runtime batch loop kept its closure -- and everything it points to -- in scratch)
        struct SegAt { int sgi, bi; };
        auto seg_of = [nb0, nb1](int b) { SegAt r{0, b}; if (r.bi >= nb0) { r.bi -= nb0; r.sgi = 1; if (r.bi >= nb1) { r.bi -= nb1; r.sgi = 2; } } return r; };
#define RS_DEP_WAIT(b) do { const SegAt at_ = seg_of(b); const RsSegPtr Sd_ = &R->seg[at_.sgi]; \
            if (at_.bi == 0) { const int dep_ = Sd_->dep; \
                if (dep_ >= 0) { const unsigned tg_ = Sd_->dep_mult * (unsigned)(Sd_->prev ? t : t + 1); \
                    unsigned sv_ = at_.sgi == 0 ? sn0 : at_.sgi == 1 ? sn1 : sn2; \
                    if (sv_ < tg_ || (a.dbg & 1)) { \
                        ok = rs_wait(a, a.counters + dep_ * NG + g, tg_, s_flag, at_.sgi, sv_, srow) && ok; \
                        sn0 = at_.sgi == 0 ? sv_ : sn0; sn1 = at_.sgi == 1 ? sv_ : sn1; sn2 = at_.sgi == 2 ? sv_ : sn2;   /* (plain selects: stores under if / else are sunk into one store through a selected address -> scratch) */ } } } } while (0)
#define RS_ISSUE(b, SLOT) do { const SegAt at_ = seg_of(b); const RsSegPtr Si_ = &R->seg[at_.sgi]; \
            const unsigned ld_ = (unsigned)Si_->ld; \
            const float* src_ = Si_->A + (size_t)(Si_->prev ? off_p : off_t) * ld_ + at_.bi * LSTM_KB; \
            float* buf_ = ring + (SLOT) * RS_BUF_FLOATS; \
            _Pragma("unroll") for (int e = 0; e < NT; ++e) \
                __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(src_ + (size_t)drw[e] * ld_ + dsw), AM_LDS_PTR(buf_ + (wave + 8 * e) * 256), 16, 0, 16 /* sc1 */); \
            rs_load_w(wv[SLOT], wbase + (size_t)(Si_->wb0 + at_.bi) * (8 * 1024), tid16); } while (0)
        const float* fbase = ring + l15 * RS_TILE_LD + kh * (LSTM_KB / 2);
        int xs[LSTM_KS];
#pragma unroll
        for (int q = 0; q < LSTM_KS; ++q) xs[q] = 4 * (((4 * q + lq) ^ l15) & 15);
#pragma unroll
        for (int b = 0; b < RS_LEAD; ++b)
            if (b < nbt) { RS_DEP_WAIT(b); RS_ISSUE(b, b); }
        for (int b0 = 0; b0 < nbt; b0 += RS_NBUF) {
#pragma unroll
            for (int i = 0; i < RS_NBUF; ++i) {
                const int b = b0 + i;
                if (b < nbt) {   // (uniform)
                    const int issued = b + RS_LEAD < nbt ? b + RS_LEAD : nbt;
                    unsigned long long* trow = (srow && threadIdx.x == 0 && t == (a.dbg >> 8) && b < 16 && !(a.dbg & 128)) ? a.stamps + (size_t)a.n_roles * a.T * 8 + (srow - a.stamps) / ((size_t)a.T * 8) * 64 + b * 4 : nullptr;
                    if (trow) trow[0] = __builtin_amdgcn_s_memrealtime();
                    rs_wait_stages<PS, RS_LEAD>(issued - (b + 1));                // stage b has landed (this wave's part)
                    rs_pin_w(wv[i]);
                    if (b == 0) {
                        rs_pin_acc<NT>(ai);
#pragma unroll
                        for (int r = 0; r < NT; ++r) acc[r] = init ? ai[r] : acc[r];
                        RS_DUMP(ai[0]);
                    }
                    if (trow) trow[1] = __builtin_amdgcn_s_memrealtime();
                    __builtin_amdgcn_s_barrier();                        // ... everybody's; and everyone has left tile b - 1
                    if (trow) trow[2] = __builtin_amdgcn_s_memrealtime();
                    // (measured and dropped: the request behind the batch's fragment reads / between its MFMAs, as persist_products does it, and all of a
                    //  thin batch's fragments fetched before its first MFMA -- 3 x 512 MGRU 10.96 -> 11.46 ms, GRU 21.4 -> 22.2)
                    if (b + RS_LEAD < nbt) { RS_DEP_WAIT(b + RS_LEAD); RS_ISSUE(b + RS_LEAD, (i + RS_LEAD) % RS_NBUF); }
                    if (trow) trow[3] = __builtin_amdgcn_s_memrealtime();
                    const float* sa = fbase + i * RS_BUF_FLOATS;
                    f32x4 pb[NT];
#pragma unroll
                    for (int r = 0; r < NT; ++r) pb[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q < LSTM_KS; ++q) {
                        f32x4 af[NT];
#pragma unroll
                        for (int r = 0; r < NT; ++r) af[r] = *(const f32x4*)(sa + r * 16 * RS_TILE_LD + xs[q]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < NT; ++r) {
                                if constexpr (NT == 1) {   // a single chain of dependent MFMAs accumulates in place (see stream_pass, am_lstm_stream.hip)
                                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(pb[0]) : "v"(af[0][j]), "v"(wv[i][q][j]));
                                } else
                                pb[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j], wv[i][q][j], pb[r], 0, 0, 0);
                            }
                    }
                    if constexpr (NT == 1) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        acc[r] += pb[r];
                        asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][1]), "+v"(acc[r][2]), "+v"(acc[r][3]));   // (the adds stay here: am_lstm_stream.hip)
                    }
                }
            }
        }
#undef RS_ISSUE
#undef RS_DEP_WAIT
        __syncthreads();   // every fragment read is done: the ring may be rewritten (K-half exchange, next pass)
        if (!ok) return false;
    }
    if (srow && threadIdx.x == 0) srow[4] = __builtin_amdgcn_s_memrealtime();
    if constexpr (EPI == REC_XPART) {   // the input part of a split phase: the accumulators as they are, 1 KB per wave and tile
        RS_DUMP(acc[0]);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(R->xplane + ((size_t)(toff + g + NG * i0) * R->n_col + bt) * (8 * 256)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < NT; ++r) st_sc1_b128(xrs, (unsigned)(NG * r) * (unsigned)R->n_col * 8192u + acc16, acc[r]);
        return true;
    }
    // the epilogue's element-wise inputs may come from a role no tile wait of this step covered (GRU B at t == 0: Z of phase A)
    if (first) { const int edep = R->edep; if (edep >= 0 && !rs_wait(a, a.counters + edep * NG + g, R->edep_mult, s_flag, 3, sn3, srow)) return false; }

    // K halves meet through LDS: [kh][column tile][ROWS][17]
    float* sg = ring;
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sg[((kh * 4 + cg) * ROWS + r * 16 + lq * 4 + reg) * RS_SGLD + l15] = acc[r][reg];
    __syncthreads();
    // element-wise epilogue: item = (row, column tile, quad of 4 consecutive units): 16-B loads of the inputs, 16-B write-through stores of the outputs
    constexpr int QPT = NU / 4;                             // quads per column tile
    constexpr int ITEMS = ROWS * 4 * QPT;
    const AM_GLOBAL_AS float* bias = (const AM_GLOBAL_AS float*)AM_GLOBAL_PTR(first ? R->bias_first : R->bias_rest);
    const __amdgpu_buffer_rsrc_t e0rs = __builtin_amdgcn_make_buffer_rsrc((void*)R->E0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t e1rs = __builtin_amdgcn_make_buffer_rsrc((void*)R->E1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t o0rs = __builtin_amdgcn_make_buffer_rsrc((void*)R->O0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t o1rs = __builtin_amdgcn_make_buffer_rsrc((void*)R->O1, 0, 0x7fffffff, 0x00020000);
    for (int it = tid; it < ITEMS; it += 512) {
        const int erow = it / (4 * QPT), rest = it - erow * (4 * QPT), ct = rest / QPT, q = rest - ct * QPT;
        const int tile = erow >> 4, slot = 16 * (g + NG * (i0 + tile)) + (erow & 15);
        const int unit0 = bt * UB + ct * NU + 4 * q;
        if (slot >= n || unit0 >= a.U_real) continue;
        const unsigned eo_t = ((off_t + (unsigned)slot) * (unsigned)U_pad + (unsigned)unit0) * 4u;
        const unsigned eo_p = ((off_p + (unsigned)slot) * (unsigned)U_pad + (unsigned)unit0) * 4u;
        f32x4 e0 = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, e1 = e0;
        if (EPI == REC_PEEP_B) { e0 = ld_sc1_b128(e0rs, eo_t); e1 = ld_sc1_b128(e1rs, eo_t); }
        else if (EPI == REC_GRU_B) { e1 = ld_sc1_b128(e1rs, eo_t); if (!first) e0 = ld_sc1_b128(e0rs, eo_p); }
        else if (!first) e0 = ld_sc1_b128(e0rs, eo_p);
        f32x4 o0, o1 = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre[G];
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                const int c16 = gg * NU + 4 * q + j;
                pre[gg] = (sg[((0 * 4 + ct) * ROWS + erow) * RS_SGLD + c16] + sg[((1 * 4 + ct) * ROWS + erow) * RS_SGLD + c16]) + bias[bt * 64 + ct * 16 + c16];
            }
            float r0, r1 = 0.0f;
            rec_cell_math<G, EPI>(pre, e0[j], e1[j], first, a.act, r0, r1);
            o0[j] = r0; o1[j] = r1;
        }
        constexpr bool TWO = EPI == REC_GRU_A || EPI == REC_PEEP_A;
        if (unit0 + 4 <= a.U_real) {
            st_sc1_b128(o0rs, eo_t, o0);
            if (TWO) st_sc1_b128(o1rs, eo_t, o1);
        } else {   // the last real units of a width that is not a multiple of 4
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (unit0 + j < a.U_real) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o0[j]), o0rs, (int)(eo_t + 4u * j), 0, 16 /* sc1 */);
                    if (TWO) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o1[j]), o1rs, (int)(eo_t + 4u * j), 0, 16 /* sc1 */);
                }
        }
    }
    __syncthreads();   // the exchange has been read: the next pass may refill the ring
    return true;
}

template <int G, int EPI>
__device__ __forceinline__ bool rs_pass_nt(int nt, const RecStreamArgs& a, RsRolePtr R, int g, int bt, int t, int i0, int n, int toff, unsigned off_t, unsigned off_p,
                                           float* ring, int* s_flag, unsigned& sn0, unsigned& sn1, unsigned& sn2, unsigned& sn3, unsigned& sn4, unsigned long long* srow)
{
    switch (nt) {   // uniform per workgroup
    case 1: return rs_pass<G, EPI, 1>(a, R, g, bt, t, i0, n, toff, off_t, off_p, ring, s_flag, sn0, sn1, sn2, sn3, sn4, srow);
    case 2: return rs_pass<G, EPI, 2>(a, R, g, bt, t, i0, n, toff, off_t, off_p, ring, s_flag, sn0, sn1, sn2, sn3, sn4, srow);
    case 3: return rs_pass<G, EPI, 3>(a, R, g, bt, t, i0, n, toff, off_t, off_p, ring, s_flag, sn0, sn1, sn2, sn3, sn4, srow);
    case 4: return rs_pass<G, EPI, 4>(a, R, g, bt, t, i0, n, toff, off_t, off_p, ring, s_flag, sn0, sn1, sn2, sn3, sn4, srow);
    default: return true;
    }
}

__global__ __launch_bounds__(512, 2) void rec_stream_kernel(const RecStreamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float rs_smem[];
    __shared__ int s_flag[12];   // [0] the pass has been aborted; [4..8] the value rs_wait's polling lane read, per dependency slot (0..4)
    if (threadIdx.x < 12) s_flag[threadIdx.x] = 0;
    __syncthreads();
    // Workgroup -> (column workgroup w of a group, group g).  The NG workgroups with the same w stream the SAME 256 KB of weights every step: they get
    // the same workgroup id % 8, i.e. one XCD under round-robin placement (speed only), so that its L2 serves all but the first of them -- dealt as
    // w = id % NG ... they sat on NG different XCDs and the launch pulled NG times the weight set through the fabric per step (3 x 512 MGRU: 61 MB,
    // ~12 us at the rate the chip streams; measured 14.6 ms per pass against the per-tick kernels' 11.0).
    const int id = blockIdx.x;
    const int w = (a.dbg & 2) ? id / a.NG : (id & 7) + 8 * (id / (8 * a.NG)), g = (a.dbg & 2) ? id % a.NG : (id >> 3) % a.NG;
    if (w >= a.wgs_per_group) return;                        // (the grid is padded to 8 NG ceil(W / 8))
    int ri = 0;
    const RsRolePtr roles = (RsRolePtr)a.roles;
    for (int r = 1; r < a.n_roles; ++r) if (w >= roles[r].wg0) ri = r;
    // (the role table lives in device memory, not in the kernel arguments: indexed by a runtime value there, the compiler copies the whole 3-KB argument
    //  block to scratch)
    const RsRolePtr R = roles + ri;
    const int bt = w - R->wg0;
    const int epi = R->epi, counter = R->counter;
    const AM_CONST_AS int* n_active = (const AM_CONST_AS int*)a.n_active;
    const AM_CONST_AS int* step_off = (const AM_CONST_AS int*)a.step_off;
    unsigned sn0 = 0u, sn1 = 0u, sn2 = 0u, sn3 = 0u, sn4 = 0u;   // rs_wait: the last value read from the counter behind each dependency
    for (int t = 0; t < a.T; ++t) {
        const int n = n_active[t];
        const int tiles = (n + 15) >> 4;
        if (g >= tiles) break;                               // this group's share has ended (the active tiles only ever decrease)
        const int ntg = (tiles - g + a.NG - 1) / a.NG;       // tiles g, g + NG, ...
        const unsigned off_t = (unsigned)step_off[t], off_p = t > 0 ? (unsigned)step_off[t - 1] : 0u;
        const int toff = a.tile_off ? ((const AM_CONST_AS int*)a.tile_off)[t] : 0;
        const int npass = (ntg + 3) / 4, q = (ntg + npass - 1) / npass;
        // diagnostic timeline (AM_LSTM_STAMPS=1; workgroup 0 of every role, group 0): [role][t][8] = step start, time polling for the x rows / the
        // previous step's rows / anything else, products done, stores issued, published, end of the last poll for the previous step's rows
        unsigned long long* srow = (a.stamps && g == 0 && bt == 0) ? a.stamps + ((size_t)ri * a.T + t) * 8 : nullptr;
        if (srow && threadIdx.x == 0) srow[0] = __builtin_amdgcn_s_memrealtime();
        for (int p = 0; p < npass; ++p) {
            const int i0 = p * q;
            int nt = ntg - i0;
            nt = nt > q ? q : nt;
            if (nt <= 0) break;
            bool ok;
            switch (epi) {
            case REC_MGRU: ok = rs_pass_nt<2, REC_MGRU>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            case REC_GRU_A: ok = rs_pass_nt<2, REC_GRU_A>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            case REC_GRU_B: ok = rs_pass_nt<1, REC_GRU_B>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            case REC_XPART: ok = rs_pass_nt<1, REC_XPART>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            case REC_PEEP_A: ok = rs_pass_nt<4, REC_PEEP_A>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            default: ok = rs_pass_nt<1, REC_PEEP_B>(nt, a, R, g, bt, t, i0, n, toff, off_t, off_p, rs_smem, s_flag, sn0, sn1, sn2, sn3, sn4, srow); break;
            }
            if (!ok) return;
        }
        // publish: every storing wave drains its stores, the workgroup meets, ONE lane signals for all of them
        if (srow && threadIdx.x == 0) srow[5] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counters + counter * a.NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (srow && threadIdx.x == 0) srow[6] = __builtin_amdgcn_s_memrealtime();
    }
}

// Row-major packed weights [rows, ld] (rows a multiple of 64, ld of 128) -> the order rec_stream_kernel streams them in (rs_load_w)
void rec_stream_pack_weights(const float* w, int rows, int ld, float* out)
{
    const int NB = ld / LSTM_KB;
    for (int bt = 0; bt < rows / 64; ++bt)
        for (int b = 0; b < NB; ++b)
            for (int wave = 0; wave < 8; ++wave) {
                const int cg = wave & 3, kh = wave >> 2;
                for (int q = 0; q < LSTM_KS; ++q)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int l15 = lane & 15, lq = lane >> 4;
                        const float* src = w + (size_t)(bt * 64 + cg * 16 + l15) * ld + b * LSTM_KB + kh * (LSTM_KB / 2) + 16 * q + 4 * lq;
                        float* dst = out + ((((size_t)(bt * NB + b) * 8 + wave) * LSTM_KS + q) * 64 + lane) * 4;
                        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                    }
            }
}

hipError_t init_rec_stream_kernel()
{
    return hipFuncSetAttribute((const void*)rec_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)REC_STREAM_LDS);
}

// Workgroups of the kernel the device holds at once (one per CU), 0 when this build of it uses scratch memory (see lstm_persist_scratch_free).
int rec_stream_resident_wgs(int n_cus)
{
    static int st = -1;
    if (st < 0) {
        hipFuncAttributes at{};
        const hipError_t e = hipFuncGetAttributes(&at, (const void*)rec_stream_kernel);
        st = (e == hipSuccess && at.localSizeBytes == 0) ? 1 : 0;
        if (!st) fprintf(stderr, "libamfwd: the persistent cell kernel of this build uses scratch memory (%zu B per lane); staying on the per-tick kernels\n", (size_t)at.localSizeBytes);
    }
    if (!st) return 0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)rec_stream_kernel, 512, REC_STREAM_LDS) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu > 0 ? n_cus : 0;
}

hipError_t launch_rec_stream(hipStream_t s, const RecStreamArgs& a)
{
    if (a.n_roles < 1 || a.n_roles > RS_MAX_ROLES || a.NG < 1 || a.T < 1) return hipErrorInvalidValue;
    if (a.NG * a.wgs_per_group > a.max_wgs) return hipErrorInvalidValue;   // every workgroup must be resident
    const int grid = 8 * a.NG * ((a.wgs_per_group + 7) / 8);               // (padded: see the workgroup map; the pad workgroups exit at once)
    RecStreamArgs b = a;
    const char* dv = getenv("AM_REC_DBG");
    b.dbg = dv ? atoi(dv) : 0;
    hipLaunchKernelGGL(rec_stream_kernel, dim3(grid), dim3(512), REC_STREAM_LDS, s, b);
    return hipGetLastError();
}

}  // namespace am
