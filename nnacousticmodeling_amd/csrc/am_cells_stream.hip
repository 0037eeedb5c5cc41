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
constexpr int RS_NBUF = 4, RS_LEAD = RS_NBUF - 1, RS_BUF_ROWS = 64;
constexpr int RS_BUF_FLOATS = RS_BUF_ROWS * RS_TILE_LD;
constexpr size_t REC_STREAM_LDS = (size_t)RS_NBUF * RS_BUF_FLOATS * sizeof(float);   // 131,072 B; the K-half exchange aliases it
constexpr int RS_SGLD = 17;

// (k is one of three compile-time values per call site)
template <int PS>
__device__ __forceinline__ void rs_wait_stages(int later)   // stages requested behind the one that must have landed
{
    if (later >= 2) {
        if constexpr (PS == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (PS == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (PS == 7) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (later == 1) {
        if constexpr (PS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (PS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (PS == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One lane polls `*c >= target`; then a barrier that does not drain vector memory.  false (all threads) when the pass has been aborted.
__device__ __forceinline__ bool rs_wait(const RecStreamArgs& a, const unsigned* c, unsigned target, int* s_flag)
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)AM_LDS_PTR(s_flag)) == 0;
}

// One pass of a step of one role: NT row tiles (this group's tiles i0 .. i0 + NT - 1, i.e. slots 16 (g + NG i)..) x this workgroup's 64 packed columns.
template <int G, int EPI, int NT>
__device__ __forceinline__ bool rs_pass(const RecStreamArgs& a, const RecStreamRole& R, int g, int bt, int t, int i0, int n, unsigned off_t, unsigned off_p,
                                        float* ring, int* s_flag)
{
    constexpr int ROWS = NT * 16;
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
    const RecStreamSeg& S0 = R.seg[0]; const RecStreamSeg& S1 = R.seg[1]; const RecStreamSeg& S2 = R.seg[2];
    const int nb0 = (S0.nb > 0 && !(first && S0.skip_first)) ? S0.nb : 0, nb1 = (S1.nb > 0 && !(first && S1.skip_first)) ? S1.nb : 0, nb2 = (S2.nb > 0 && !(first && S2.skip_first)) ? S2.nb : 0;
    const int nbt = nb0 + nb1 + nb2;
    f32x4 acc[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

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
        const size_t wrow = (size_t)(bt * 64 + cg * 16 + l15) * R.ldw + kh * (LSTM_KB / 2) + 4 * lq;
        f32x4 wv[RS_NBUF][LSTM_KS];
        // stage b -> (segment, batch in the segment): at most three segments, walked by comparison (plain value-returning helpers: a lambda that
        // writes captured variables from inside the runtime batch loop kept its closure -- and everything it points to -- in scratch)
        struct SegAt { int sgi, bi; };
        auto seg_of = [nb0, nb1](int b) { SegAt r{0, b}; if (r.bi >= nb0) { r.bi -= nb0; r.sgi = 1; if (r.bi >= nb1) { r.bi -= nb1; r.sgi = 2; } } return r; };
#define RS_SEG(at) ((at).sgi == 0 ? S0 : (at).sgi == 1 ? S1 : S2)
#define RS_DEP_WAIT(b) do { const SegAt at_ = seg_of(b); const RecStreamSeg& Sd_ = RS_SEG(at_); \
            if (at_.bi == 0 && Sd_.dep >= 0) ok = rs_wait(a, a.counters + Sd_.dep * NG + g, Sd_.dep_mult * (unsigned)(Sd_.prev ? t : t + 1), s_flag) && ok; } while (0)
#define RS_ISSUE(b, SLOT) do { const SegAt at_ = seg_of(b); const RecStreamSeg& Si_ = RS_SEG(at_); \
            const int k0_ = (at_.sgi >= 1 ? S0.nb * LSTM_KB : 0) + (at_.sgi >= 2 ? S1.nb * LSTM_KB : 0);   /* weight columns of the segments before it (whether or not they exist at this step) */ \
            const unsigned ld_ = (unsigned)Si_.ld; \
            const float* src_ = Si_.A + (size_t)(Si_.prev ? off_p : off_t) * ld_ + at_.bi * LSTM_KB; \
            float* buf_ = ring + (SLOT) * RS_BUF_FLOATS; \
            _Pragma("unroll") for (int e = 0; e < NT; ++e) \
                __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(src_ + (size_t)drw[e] * ld_ + dsw), AM_LDS_PTR(buf_ + (wave + 8 * e) * 256), 16, 0, 16 /* sc1 */); \
            const float* wp_ = R.W + wrow + k0_ + at_.bi * LSTM_KB; \
            _Pragma("unroll") for (int q = 0; q < LSTM_KS; ++q) wv[SLOT][q] = *(const f32x4*)(wp_ + 16 * q); } while (0)
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
                    rs_wait_stages<PS>(issued - (b + 1));                // stage b has landed (this wave's part)
                    __builtin_amdgcn_s_barrier();                        // ... everybody's; and everyone has left tile b - 1
                    if (b + RS_LEAD < nbt) { RS_DEP_WAIT(b + RS_LEAD); RS_ISSUE(b + RS_LEAD, (i + RS_LEAD) % RS_NBUF); }
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
#undef RS_SEG
        __syncthreads();   // every fragment read is done: the ring may be rewritten (K-half exchange, next pass)
        if (!ok) return false;
    }
    // the epilogue's element-wise inputs may come from a role no tile wait of this step covered (GRU B at t == 0: Z of phase A)
    if (R.edep >= 0 && first && !rs_wait(a, a.counters + R.edep * NG + g, R.edep_mult, s_flag)) return false;

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
    const float* bias = first ? R.bias_first : R.bias_rest;
    const __amdgpu_buffer_rsrc_t e0rs = __builtin_amdgcn_make_buffer_rsrc((void*)R.E0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t e1rs = __builtin_amdgcn_make_buffer_rsrc((void*)R.E1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t o0rs = __builtin_amdgcn_make_buffer_rsrc((void*)R.O0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t o1rs = __builtin_amdgcn_make_buffer_rsrc((void*)R.O1, 0, 0x7fffffff, 0x00020000);
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
__device__ __forceinline__ bool rs_pass_nt(int nt, const RecStreamArgs& a, const RecStreamRole& R, int g, int bt, int t, int i0, int n, unsigned off_t, unsigned off_p,
                                           float* ring, int* s_flag)
{
    switch (nt) {   // uniform per workgroup
    case 1: return rs_pass<G, EPI, 1>(a, R, g, bt, t, i0, n, off_t, off_p, ring, s_flag);
    case 2: return rs_pass<G, EPI, 2>(a, R, g, bt, t, i0, n, off_t, off_p, ring, s_flag);
    case 3: return rs_pass<G, EPI, 3>(a, R, g, bt, t, i0, n, off_t, off_p, ring, s_flag);
    case 4: return rs_pass<G, EPI, 4>(a, R, g, bt, t, i0, n, off_t, off_p, ring, s_flag);
    default: return true;
    }
}

__global__ __launch_bounds__(512, 2) void rec_stream_kernel(const RecStreamArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float rs_smem[];
    __shared__ int s_flag[2];
    if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    // workgroups b, b + NG, ... form a group: a group's roles share rows, not an XCD (speed only either way)
    const int g = blockIdx.x % a.NG, w = blockIdx.x / a.NG;
    int ri = 0;
    for (int r = 1; r < a.n_roles; ++r) if (w >= a.roles[r].wg0) ri = r;
    // (the role table lives in device memory, not in the kernel arguments: indexed by a runtime value there, the compiler copies the whole 3-KB argument
    //  block to scratch)
    const RecStreamRole& R = a.roles[ri];
    const int bt = w - R.wg0;
    for (int t = 0; t < a.T; ++t) {
        const int n = a.n_active[t];
        const int tiles = (n + 15) >> 4;
        if (g >= tiles) break;                               // this group's share has ended (the active tiles only ever decrease)
        const int ntg = (tiles - g + a.NG - 1) / a.NG;       // tiles g, g + NG, ...
        const unsigned off_t = (unsigned)a.step_off[t], off_p = t > 0 ? (unsigned)a.step_off[t - 1] : 0u;
        const int npass = (ntg + 3) / 4, q = (ntg + npass - 1) / npass;
        for (int p = 0; p < npass; ++p) {
            const int i0 = p * q;
            int nt = ntg - i0;
            nt = nt > q ? q : nt;
            if (nt <= 0) break;
            bool ok;
            switch (R.epi) {
            case REC_MGRU: ok = rs_pass_nt<2, REC_MGRU>(nt, a, R, g, bt, t, i0, n, off_t, off_p, rs_smem, s_flag); break;
            case REC_GRU_A: ok = rs_pass_nt<2, REC_GRU_A>(nt, a, R, g, bt, t, i0, n, off_t, off_p, rs_smem, s_flag); break;
            case REC_GRU_B: ok = rs_pass_nt<1, REC_GRU_B>(nt, a, R, g, bt, t, i0, n, off_t, off_p, rs_smem, s_flag); break;
            case REC_PEEP_A: ok = rs_pass_nt<4, REC_PEEP_A>(nt, a, R, g, bt, t, i0, n, off_t, off_p, rs_smem, s_flag); break;
            default: ok = rs_pass_nt<1, REC_PEEP_B>(nt, a, R, g, bt, t, i0, n, off_t, off_p, rs_smem, s_flag); break;
            }
            if (!ok) return;
        }
        // publish: every storing wave drains its stores, the workgroup meets, ONE lane signals for all of them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counters + R.counter * a.NG + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const int grid = a.NG * a.wgs_per_group;
    if (grid > a.max_wgs) return hipErrorInvalidValue;   // every workgroup must be resident
    hipLaunchKernelGGL(rec_stream_kernel, dim3(grid), dim3(512), REC_STREAM_LDS, s, a);
    return hipGetLastError();
}

}  // namespace am
