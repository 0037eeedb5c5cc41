// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: the frame-synchronous Viterbi decoder
#include "am_device.h"

namespace am {

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

// The fused launch as a STREAM (round 4).  viterbi_fs_fused_kernel's block lives for one (slot, triphone block) item: scalar preamble, every load
// in flight, wait, update, stores, node maxima, exit -- with two blocks per CU the memory pipe drains in every block's compute and node phases
// and in the gaps between blocks, and each item re-fetches its 32 B of graph constants per triphone from L2 (64 KB of the 119 KB an item loads):
// 4.3 TB/s at the fabric where a copy reaches 6.3.  Here a block keeps ONE triphone block for the whole launch and walks the slots g, g + G,
// g + 2G, ...: the constants, the list bounds and the language-model terms of its nodes are loaded once per launch and stay in registers; slot
// i + 1's state (18 B per triphone, into a second register set) and its two gather arrays (log-posterior row + node scores, by LDS-DMA into the
// other half of a double buffer: no registers) are requested before slot i is updated, so a CU's loads never stop while it computes.
// G = resident blocks / triphone blocks (2 x 256 / 44 = 11 at the real size); a launch with fewer active utterances than that runs one slot per
// block.  Same arithmetic, comparisons, tie-breaking and store order per address as the fused kernel: bit-identical segmentations.
// Threads past the end of a short block repeat its last triphone (same loads, same results, same stores): every wave issues the same number of
// vector-memory operations per slot, nothing is predicated in the update.
struct VitSlotState { i32x4 qa[2], qb[2]; int ta[2], tb[2]; };

template <int NTHR>
__global__ __launch_bounds__(NTHR, (2 * NTHR) / 256) void viterbi_fs_stream_kernel(const ViterbiArgs a, int k, int n_active, int G, int xcd_map)
{
    constexpr int CAP = 2048, IT = CAP / (2 * NTHR), HALF = NTHR;
    static_assert(IT == 2, "VitSlotState holds two trips");
    extern __shared__ float vsm[];
    const int T = a.T, B = a.B, L = (int)a.ldl;
    const int SB = (L + B + NTHR - 1) / NTHR * NTHR;           // floats per staging buffer: [L] log-posteriors of frame k, [B] node scores, padding
    float* s_po = vsm + 2 * SB;
    int* s_t2 = (int*)(s_po + CAP);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (slot group, triphone block) of this workgroup.  The blocks of a slot group stage the same log-posterior rows and node scores; under
    // round-robin placement (workgroup id % 8 = XCD: speed only) the first 8 groups get an XCD each, so that those 17 KB per slot are fetched
    // into one L2 instead of eight; the remaining groups are dealt over all XCDs.
    int grp, blk;
    {
        const int id = blockIdx.x, nb = a.n_blk;
        if (xcd_map && G >= 8 && id < 8 * nb) { grp = id & 7; blk = id >> 3; }   // the first 8 groups: one XCD each
        else { grp = id / nb; blk = id - grp * nb; }                            // (ids >= 8 * nb: groups 8 .. G - 1, over all XCDs)
    }
    const int tri0 = a.blk_tri0[blk], tri1 = a.blk_tri0[blk + 1];
    const int node0 = a.blk_node0[blk], node1 = a.blk_node0[blk + 1];
    const i32x4* c0 = (const i32x4*)a.tri_idx;
    const f32x4* c1 = (const f32x4*)a.tri_self;
    // per launch: this thread's triphones, their graph constants, its nodes' list bounds
    int na[IT], nb[IT];
    i32x4 ga[IT], gb[IT]; f32x4 ha[IT], hb[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int n0 = tri0 + tid + it * 2 * HALF;
        na[it] = n0 < tri1 ? n0 : tri1 - 1; nb[it] = n0 + HALF < tri1 ? n0 + HALF : tri1 - 1;
        ga[it] = c0[na[it]]; gb[it] = c0[nb[it]]; ha[it] = c1[na[it]]; hb[it] = c1[nb[it]];
    }
    // the list bounds and language-model terms of this block's nodes: per launch, in LDS (registers are what bounds this kernel's occupancy)
    int* s_ns = (int*)(s_t2 + CAP);                           // [nodes + 1] list bounds, then [nodes] lm terms
    const int n_nodes = node1 - node0;
    for (int i = tid; i <= n_nodes; i += NTHR) s_ns[i] = a.outOff[node0 + i];
    for (int i = tid; i < n_nodes; i += NTHR) ((float*)s_ns)[n_nodes + 1 + i] = a.lm[node0 + i];
    // state accesses: buffer instructions (resource in SGPRs, a slot's base as the scalar offset, ONE 32-bit per-thread offset per triphone:
    // no 64-bit address registers per access); nt = streamed once per launch
    const __amdgpu_buffer_rsrc_t rs_pp = __builtin_amdgcn_make_buffer_rsrc((void*)a.pp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.tt, 0, 0x7fffffff, 0x00020000);
    constexpr int NT = 2;   // aux: nt
    // The first rows of the slots' utterances travel two slots ahead of their use, in SGPRs: the compiler cannot prove the table constant (the
    // kernel stores through other pointers), so a look-up at the point of use is a VECTOR load whose wait -- vmcnt counts in order -- would also
    // wait for the state loads just requested for the next slot.
    auto uniform64 = [](long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    long long row_cur = 0, row_nxt = 0;
    auto request = [&](int slot, long long row0, VitSlotState& S, int buf) {   // everything slot `slot` needs, in flight
        const long long row = row0 + k;
        const float* lp1 = a.lp + (size_t)row * a.ldl;
        const float* p = ((k & 1) ? a.p2 : a.p) + (size_t)slot * B;
        const int so_pp = slot * T * 16, so_t2 = slot * T * 2;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            S.qa[it] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_pp, na[it] * 16, so_pp, NT));
            S.qb[it] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_pp, nb[it] * 16, so_pp, NT));
            S.ta[it] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_t2, na[it] * 2, so_t2, NT);
            S.tb[it] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_t2, nb[it] * 2, so_t2, NT);
        }
        float* stage = vsm + buf * SB;
        for (int i0 = wave * 64; i0 < L + B; i0 += NTHR) {      // one 256-B LDS-DMA piece per wave and trip; lanes past the end repeat the last word
            int i = i0 + lane;
            i = i < L + B ? i : L + B - 1;
            const float* src = i < L ? lp1 + i : p + (i - L);
            __builtin_amdgcn_global_load_lds(AM_GLOBAL_PTR(src), AM_LDS_PTR(stage + i0), 4, 0, 0);
        }
    };
    auto advance = [&](int slot, const VitSlotState& S, VitSlotState& N, int buf) {
        long long row_n2 = 0;
        if (slot + 2 * G < n_active) row_n2 = a.slot_row0[slot + 2 * G];   // (lands under the wait below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this slot's state and this wave's staging pieces have landed ...
        row_n2 = uniform64(row_n2);
        __syncthreads();                                       // ... everyone's have; everyone is done with the previous slot's node phase
        if (slot + G < n_active) request(slot + G, row_nxt, N, buf ^ 1);
        const float* s_lp = vsm + buf * SB;
        const float* s_p = s_lp + L;
        const long long row = row_cur + k;
        row_cur = row_nxt; row_nxt = row_n2;
        float* pnext = ((k & 1) ? a.p : a.p2) + (size_t)slot * B;
        const int so_pp = slot * T * 16, so_t2 = slot * T * 2;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            i32x4 nqa, nqb; int nta, ntb; float poa, pob;
            // (the packed gather indices are made opaque per slot: loop-invariant code motion would otherwise keep the 32 LDS addresses derived
            //  from them -- 4 triphones x 4 gathers x 2 buffers -- in registers across the slot loop, which is what this kernel has none to spare of)
            asm volatile("" : "+v"(ga[it][0]), "+v"(ga[it][1]), "+v"(gb[it][0]), "+v"(gb[it][1]));
            {
                const i32x4 g = ga[it];
                viterbi_tri_apply(g, ha[it], S.qa[it], S.ta[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqa, nta, poa);
            }
            {
                const i32x4 g = gb[it];
                viterbi_tri_apply(g, hb[it], S.qb[it], S.tb[it], s_p[g[0] & 0xffff], s_lp[(unsigned)g[0] >> 16], s_lp[g[1] & 0xffff], s_lp[(unsigned)g[1] >> 16], a.PIP, k, nqb, ntb, pob);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, nqa), rs_pp, na[it] * 16, so_pp, NT);
            __builtin_amdgcn_raw_buffer_store_b16((short)nta, rs_t2, na[it] * 2, so_t2, NT);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, nqb), rs_pp, nb[it] * 16, so_pp, NT);
            __builtin_amdgcn_raw_buffer_store_b16((short)ntb, rs_t2, nb[it] * 2, so_t2, NT);
            s_po[na[it] - tri0] = poa; s_t2[na[it] - tri0] = nta; s_po[nb[it] - tri0] = pob; s_t2[nb[it] - tri0] = ntb;
            const int mine = (na[it] == a.inv0 ? 1 : 0) | (nb[it] == a.inv0 ? 2 : 0);
            if (mine) {   // the owner of the reference's triphone 0 leaves its state for the next frame's unreachable nodes
                int* st = a.inv0_state + (size_t)slot * 16 + ((k + 1) & 1) * 8;
                const i32x4 q = (mine & 1) ? nqa : nqb;
                st[0] = q[0]; st[1] = q[1]; st[2] = q[2]; st[3] = q[3]; st[4] = (mine & 1) ? nta : ntb;
            }
        }
        // the exit scores and entry times are in LDS for everyone -- a barrier that does NOT drain vector memory: __syncthreads() would wait
        // for the next slot's loads requested above (and for this slot's stores), which is the overlap this kernel exists for
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // node maxima: 8 lanes per node, the first strict maximum in list order (viterbi_fs_fused_kernel's node phase)
        const int l8 = tid & 7;
        for (int i0 = node0; i0 < node1; i0 += NTHR / 8) {
            const int i = i0 + (tid >> 3);
            const bool valid = i < node1;
            const int s0 = valid ? s_ns[i - node0] : 0, s1 = valid ? s_ns[i - node0 + 1] : 0;
            const float lmi = valid ? ((const float*)s_ns)[n_nodes + 1 + i - node0] : 0.0f;
            float mx = -1e20f; int ms = 0x7fffffff, mxi = 0;
            for (int s2 = s0 + l8; s2 < s1; s2 += 8) {
                const float pOut = s_po[s2 - tri0];
                if (mx < pOut) { mx = pOut; ms = s2; mxi = s2; }
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
                    bt = ((q1 + h0[2] > q2 + h0[0]) ? t1s : t2s) - 1;
                } else {
                    bt = s_t2[mxi - tri0] - 1;
                }
                pnext[i] = mx + lmi;
                a.bestI[(size_t)row * B + i] = mxi;
                a.bestT[(size_t)row * B + i] = bt;
            }
        }
    };
    VitSlotState SA, SBs;
    int slot = grp;
    if (slot >= n_active) return;
    row_cur = uniform64(a.slot_row0[slot]);
    row_nxt = slot + G < n_active ? uniform64(a.slot_row0[slot + G]) : 0;
    request(slot, row_cur, SA, 0);
    for (;;) {
        advance(slot, SA, SBs, 0);
        slot += G;
        if (slot >= n_active) break;
        advance(slot, SBs, SA, 1);
        slot += G;
        if (slot >= n_active) break;
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

constexpr size_t VITERBI_STREAM_LDS_MAX = 80 * 1024;   // two blocks per CU
hipError_t init_viterbi_kernels()
{
    return hipFuncSetAttribute((const void*)viterbi_fs_stream_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VITERBI_STREAM_LDS_MAX);
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
    const char* stv = getenv("AM_VITERBI_STREAM");   // 0: viterbi_fs_fused_kernel (A/B timing, cross-form tests); > 1: that many slot groups; read per launch like the others
    const int stream_mode = stv ? atoi(stv) : 1;
    // two staging buffers, exit scores + entry times of a block's <= 2,048 triphones, list bounds + lm terms of its nodes
    const size_t stream_lds = (2 * ((size_t)(a.ldl + a.B + 511) / 512 * 512) + 2 * 2048 + 2 * (size_t)a.blk_nodes_max + 2) * sizeof(float);
    if (stream_mode && (!sv || atoi(sv) > 2) && a.n_blk > 0 && a.perm && a.slot_row0 && a.blk_nodes_max > 0 && stream_lds <= VITERBI_STREAM_LDS_MAX &&
        (size_t)n_active * a.T * 16 < 0x7fffffffu) {   // (a slot's state is addressed by a 32-bit buffer offset)
        // slots per block: as many blocks as the device holds at once (two per CU), each walking n_active / G slots
        int G = std::max(1, 2 * std::max(a.n_cus, 1) / a.n_blk);
        if (stream_mode > 1) G = stream_mode;   // (measurement: a given number of slot groups)
        G = std::min(G, n_active);
        const char* xm = getenv("AM_VITERBI_XCD_MAP");   // (measurement: 0 = groups in workgroup-id order)
        hipLaunchKernelGGL(viterbi_fs_stream_kernel<512>, dim3(a.n_blk * G), dim3(512), stream_lds, s, a, k, n_active, G, xm ? atoi(xm) : 1);
        return hipGetLastError();
    }
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
