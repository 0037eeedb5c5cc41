// GEMM laboratory: times variants of the f32-MFMA NT GEMM on the bench shape (M=196608, N=1024/1920, K=1024/448)
// in ONE process, interleaved rounds (guide rule 24).  Not part of the product; the winner is ported to am_kernels.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM x BN block tile, BK k-step, STAGES LDS buffers (prefetch distance STAGES-1), WM x WN waves, MINW = min waves/SIMD.
// ORDER 0: [issue next][compute][vmcnt(0)][__syncthreads]  (the round-1a kernel)
// ORDER 1: [counted vmcnt][raw barrier][issue kt+P][compute]  (loads stay in flight across barriers)
// FRAGPF 1: explicit register double-buffer of the LDS fragments
template <int BM, int BN, int BK, int STAGES, int WM, int WN, int MINW, int ORDER, int FRAGPF, int PRIO, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_var(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, float* __restrict__ C, int ldc, int tiles_n, int nk)
{
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int CH = BK / 4;                   // 16-B chunks per row
    constexpr int RPI = 64 / CH;                 // rows per wave-instruction (1 KiB)
    constexpr int STAGE_BYTES = (BM + BN) * BK * 4;
    constexpr int A_PIECES = BM / RPI, W_PIECES = BN / RPI;  // wave-instructions per stage
    constexpr int A_PER_WAVE = A_PIECES / NW, W_PER_WAVE = W_PIECES / NW;
    constexpr int LOADS = A_PER_WAVE + W_PER_WAVE;           // glds per wave per stage
    constexpr int TS = BK / 8;                                // t-steps per k-step
    // swizzle: chunk' = chunk ^ f(row); 128-B rows: (row>>1)&7 ; 64-B rows: (row>>2)&3
    constexpr int SW_SHIFT = (BK == 32) ? 1 : 2;
    constexpr int SW_MASK = CH - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

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
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const int k0 = (ABL & 4) ? 0 : kt * BK;
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j)
            __builtin_amdgcn_global_load_lds(GP(ga[j] + k0), LP(base + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < W_PER_WAVE; ++j)
            __builtin_amdgcn_global_load_lds(GP(gw[j] + k0), LP(base + BM * BK * 4 + (j * NW + wave) * 1024), 16, 0, 0);
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
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * (BM / WM) + mi * 32 + li; aidx[mi] = r * CH; asw[mi] = (r >> SW_SHIFT) & SW_MASK; }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * (BN / WN) + ni * 32 + li; widx[ni] = r * CH; wsw[ni] = (r >> SW_SHIFT) & SW_MASK; }

    auto load_frag = [&](const f32x4* a4, const f32x4* w4, int t, f32x4 (&av)[MI], f32x4 (&bv)[NI]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[mi] = a4[aidx[mi] + ((2 * t + lh) ^ asw[mi])];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = w4[widx[ni] + ((2 * t + lh) ^ wsw[ni])];
    };
    auto mma = [&](const f32x4 (&av)[MI], const f32x4 (&bv)[NI]) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto compute = [&](int buf) {
        const f32x4* a4 = (const f32x4*)(smem + buf * STAGE_BYTES);
        const f32x4* w4 = (const f32x4*)(smem + buf * STAGE_BYTES + BM * BK * 4);
        if (FRAGPF) {
            f32x4 av[2][MI], bv[2][NI];
            load_frag(a4, w4, 0, av[0], bv[0]);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (t + 1 < TS) load_frag(a4, w4, t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
                mma(av[t & 1], bv[t & 1]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                f32x4 av[MI], bv[NI];
                load_frag(a4, w4, t, av, bv);
                mma(av, bv);
            }
        }
    };

    f32x4 fav[MI], fbv[NI];
    if (ABL & 2) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fav[mi] = *(const f32x4*)(A + (size_t)(m0 + mi * 32 + li) * lda + 4 * lh);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fbv[ni] = *(const f32x4*)(W + (size_t)(n0 + ni * 32 + li) * ldw + 4 * lh);
    }
    auto compute_nolds = [&]() {
#pragma unroll
        for (int t = 0; t < TS; ++t) { mma(fav, fbv); asm volatile("" : "+v"(fav[0]), "+v"(fbv[0])); }
    };
    if constexpr (ORDER == 4) {
        // PING-PONG: the two waves of a SIMD alternate.  Waves 0..3 ("A") multiply in even phases and fetch in odd ones, waves 4..7
        // ("B") the other way round; a phase = one 8-k sub-step (32 MFMAs) for the multiplying half, and for the fetching half: the
        // LDS fragment reads of its next sub-step + half of a stage's LDS-DMA requests (stage s+3 of a 4-stage ring) + its waits.
        // One s_barrier between phases.  The matrix pipe always has one wave with operands in registers; LDS latency, the
        // DMA issue and the waits of one half hide behind the MFMAs of the other.
        // MEASURED (round 2, M=196608 N=K=1024): 129.9 TF/s against 143.0 for the product's order 2 -- a barrier every 2,048 cycles
        // makes every SIMD wait for the slowest multiplying wave eight times as often; the two waves of a SIMD interleave well
        // enough on their own.  Kept for the record.
        static_assert(STAGES == 4 && TS == 2 && NW == 8 && LOADS % 2 == 0, "ping-pong: 4 stages of two sub-steps, 8 waves");
        const bool grpB = wave >= NW / 2;
        auto issue_piece = [&](int st, int l) {
            const int kt = st < nk ? st : nk - 1;            // past the end: re-fetch the last stage into a buffer nobody reads
            char* base = smem + (st & 3) * STAGE_BYTES;
            const int k0 = kt * BK;
            if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(GP(ga[l] + k0), LP(base + (l * NW + wave) * 1024), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(GP(gw[l - A_PER_WAVE] + k0), LP(base + BM * BK * 4 + ((l - A_PER_WAVE) * NW + wave) * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int l = 0; l < LOADS; ++l) issue_piece(st, l);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        f32x4 av[MI], bv[NI];
        const int nu = nk * TS;
        auto fetch = [&](int u) {        // the fetching half's phase for sub-step u
            const int st = u >> 1, t = u & 1;
            const f32x4* a4 = (const f32x4*)(smem + (st & 3) * STAGE_BYTES);
            const f32x4* w4 = (const f32x4*)(smem + (st & 3) * STAGE_BYTES + BM * BK * 4);
            load_frag(a4, w4, t, av, bv);
#pragma unroll
            for (int l = 0; l < LOADS / 2; ++l) issue_piece(st + 3, t * (LOADS / 2) + l);
            if (t == 1) wait_vmcnt<2 * LOADS>();     // this wave's pieces of stage st+1 (first read in its next fetch phase) have landed
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(av[mi]));   // fragments complete before the barrier
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(bv[ni]));
        };
        auto multiply = [&] {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (!grpB) {
            fetch(0);
            __builtin_amdgcn_s_barrier();
            for (int u = 0; u < nu; ++u) {
                multiply();
                __builtin_amdgcn_s_barrier();
                if (u + 1 < nu) fetch(u + 1);
                __builtin_amdgcn_s_barrier();
            }
        } else {
            __builtin_amdgcn_s_barrier();
            for (int u = 0; u < nu; ++u) {
                fetch(u);
                __builtin_amdgcn_s_barrier();
                multiply();
                __builtin_amdgcn_s_barrier();
            }
        }
        wait_vmcnt<0>();
    } else if constexpr (ORDER == 3) {
        // order 2 (LDS-DMA issue interleaved between MFMA groups) + software pipelining across the barrier: the last
        // 8-k sub-step of stage kt is multiplied AFTER the barrier that publishes stage kt+1, covering the latency of the
        // first fragment reads of the new stage.
        constexpr int SLOTS = (TS - 1) * 4;      // the last sub-step runs after the barrier: no LDS-DMA issue there
        static_assert(LOADS <= SLOTS, "order 3 needs (TS-1)*4 >= LOADS");
        constexpr int EVERY = SLOTS / LOADS > 0 ? SLOTS / LOADS : 1;
        stage(0, 0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        f32x4 av[2][MI], bv[2][NI];
        {
            const f32x4* a4 = (const f32x4*)(smem);
            const f32x4* w4 = (const f32x4*)(smem + BM * BK * 4);
            load_frag(a4, w4, 0, av[0], bv[0]);
        }
        int cur = 0;   // register set holding the fragments of the NEXT sub-step to multiply
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const int kn = kt + 1 < nk ? kt + 1 : nk - 1;
            char* sbase = smem + (buf ^ 1) * STAGE_BYTES;
            const int k0 = kn * BK;
            const f32x4* a4 = (const f32x4*)(smem + buf * STAGE_BYTES);
            const f32x4* w4 = (const f32x4*)(smem + buf * STAGE_BYTES + BM * BK * 4);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (t + 1 < TS) load_frag(a4, w4, t + 1, av[cur ^ 1], bv[cur ^ 1]);
                else {
                    // stage kt fully read into registers: publish point for stage kt+1
                    wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    const f32x4* na4 = (const f32x4*)(smem + (buf ^ 1) * STAGE_BYTES);
                    const f32x4* nw4 = (const f32x4*)(smem + (buf ^ 1) * STAGE_BYTES + BM * BK * 4);
                    load_frag(na4, nw4, 0, av[cur ^ 1], bv[cur ^ 1]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi][j], bv[cur][ni][j], acc[mi][ni], 0, 0, 0);
                    // LDS-DMA of stage kt+1 goes into the other buffer: legal only while that buffer is not being read,
                    // i.e. during sub-steps 0..TS-2 of this stage (its last readers passed the previous barrier)
                    const int slot = t * 4 + j;
                    if (t + 1 < TS && slot % EVERY == 0 && slot / EVERY < LOADS) {
                        const int l = slot / EVERY;
                        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(GP(ga[l] + k0), LP(sbase + (l * NW + wave) * 1024), 16, 0, 0);
                        else __builtin_amdgcn_global_load_lds(GP(gw[l - A_PER_WAVE] + k0), LP(sbase + BM * BK * 4 + ((l - A_PER_WAVE) * NW + wave) * 1024), 16, 0, 0);
                    }
                }
                cur ^= 1;
            }
        }
    } else if constexpr (ORDER == 2) {
        constexpr int P = STAGES - 1;
        constexpr int SLOTS = TS * 4;                    // MFMA groups per k-step
        constexpr int EVERY = SLOTS / LOADS > 0 ? SLOTS / LOADS : 1;
#pragma unroll
        for (int s = 0; s < P; ++s) stage(s, s < nk ? s : nk - 1);
        int buf = 0, nbuf = P % STAGES;
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;
            if (P >= 3 && rem >= 2) wait_vmcnt<2 * LOADS>();
            else if (P >= 2 && rem >= 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const int kn = kt + P < nk ? kt + P : nk - 1;   // clamped: tail prefetches land in a buffer nobody reads
            char* sbase = smem + nbuf * STAGE_BYTES;
            const int k0 = kn * BK;
            const f32x4* a4 = (const f32x4*)(smem + buf * STAGE_BYTES);
            const f32x4* w4 = (const f32x4*)(smem + buf * STAGE_BYTES + BM * BK * 4);
            f32x4 av[2][MI], bv[2][NI];
            load_frag(a4, w4, 0, av[0], bv[0]);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (t + 1 < TS) load_frag(a4, w4, t + 1, av[(t + 1) & 1], bv[(t + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][mi][j], bv[t & 1][ni][j], acc[mi][ni], 0, 0, 0);
                    const int slot = t * 4 + j;
                    if (slot % EVERY == 0 && slot / EVERY < LOADS) {
                        const int l = slot / EVERY;
                        if (l < A_PER_WAVE) __builtin_amdgcn_global_load_lds(GP(ga[l] + k0), LP(sbase + (l * NW + wave) * 1024), 16, 0, 0);
                        else __builtin_amdgcn_global_load_lds(GP(gw[l - A_PER_WAVE] + k0), LP(sbase + BM * BK * 4 + ((l - A_PER_WAVE) * NW + wave) * 1024), 16, 0, 0);
                    }
                    if (PRIO) __builtin_amdgcn_sched_barrier(0);
                }
            }
            buf = (buf + 1 == STAGES) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
        }
    } else if constexpr (ORDER == 0) {
        stage(0, 0);
        wait_vmcnt<0>();
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt % STAGES;
            if (kt + 1 < nk) stage((kt + 1) % STAGES, kt + 1);
            compute(buf);
            wait_vmcnt<0>();
            __syncthreads();
        }
    } else {
        constexpr int P = STAGES - 1;
#pragma unroll
        for (int s = 0; s < P; ++s) if (s < nk) stage(s, s);
        int buf = 0, nbuf = P % STAGES;
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;  // stages issued after kt (capped at P-1) may stay in flight
            if (P >= 3 && rem >= 2) wait_vmcnt<2 * LOADS>();
            else if (P >= 2 && rem >= 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + P < nk && !(ABL & 1)) stage(nbuf, kt + P);
            if (ABL & 2) compute_nolds(); else compute((ABL & 1) ? 0 : buf);
            buf = (buf + 1 == STAGES) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
        }
    }

    const int erow = m0 + wm * (BM / WM) + 4 * lh, ecol = n0 + wn * (BN / WN) + li;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = ecol + ni * 32;
            const float bv = bias[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = erow + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                C[(size_t)row * ldc + col] = fmaxf(acc[mi][ni][reg] + bv, 0.0f);
            }
        }
}

struct Variant {
    const char* name;
    void (*launch)(const float*, int, const float*, int, const float*, float*, int, int, int, int, hipStream_t);
};

template <int BM, int BN, int BK, int STAGES, int WM, int WN, int MINW, int ORDER, int FRAGPF, int PRIO, int ABL = 0>
void launch_var(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, hipStream_t s)
{
    auto kern = gemm_var<BM, BN, BK, STAGES, WM, WN, MINW, ORDER, FRAGPF, PRIO, ABL>;
    constexpr int lds = STAGES * (BM + BN) * BK * 4;
    static bool init = false;
    if (!init) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); init = true; }
    hipLaunchKernelGGL(kern, dim3((M / BM) * (N / BN)), dim3(64 * WM * WN), lds, s, A, lda, W, ldw, bias, C, ldc, N / BN, K / BK);
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 196608, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int rounds = argc > 4 ? atoi(argv[4]) : 5;
    printf("M=%d N=%d K=%d rounds=%d\n", M, N, K, rounds);
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *dA, *dW, *db, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, hb.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));

    std::vector<Variant> vars;
    if (N % 384 == 0 && N % 256 != 0) vars = {
        {"v30 256x384x16 s2 4x2 w2 order1 (product)", launch_var<256, 384, 16, 2, 4, 2, 2, 1, 0, 0>},
        {"v32 256x384x32 s2 4x2 w2 order2", launch_var<256, 384, 32, 2, 4, 2, 2, 2, 0, 0>},
        {"v42 256x384x32 s2 4x2 w2 order3", launch_var<256, 384, 32, 2, 4, 2, 2, 3, 0, 0>},
    };
    else vars = {
        {"v26 256x256x32 s2 4x2 w2 order2 (product)", launch_var<256, 256, 32, 2, 4, 2, 2, 2, 0, 0>},
        {"v40 256x256x32 s2 4x2 w2 order3 (pipelined across the barrier)", launch_var<256, 256, 32, 2, 4, 2, 2, 3, 0, 0>},
        {"v15 256x256x32 s2 4x2 w2 order0", launch_var<256, 256, 32, 2, 4, 2, 2, 0, 0, 0>},
        {"v50 256x256x16 s4 4x2 w2 order4 (ping-pong)", launch_var<256, 256, 16, 4, 4, 2, 2, 4, 0, 0>},
    };
    if (M % 256) { printf("M must be a multiple of 256\n"); return 1; }

    // reference on sampled entries (fmaf chain order differs per variant: tolerance)
    std::vector<int> sr = {0, 1, 127, 128, 255, 256, M / 2 + 3, M - 1}, sc = {0, 1, 31, 32, 63, 64, 127, 128, N / 2 + 5, N - 1};
    std::vector<float> hC((size_t)M * N);
    std::vector<double> times(vars.size(), 0.0), best(vars.size(), 1e30);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t v = 0; v < vars.size(); ++v) {
        CK(hipMemsetAsync(dC, 0xff, (size_t)M * N * 4, s));
        vars[v].launch(dA, K, dW, K, db, dC, N, M, N, K, s);
        CK(hipStreamSynchronize(s));
        CK(hipGetLastError());
        CK(hipMemcpy(hC.data(), dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int r : sr) for (int c : sc) {
            double ref = hb[c];
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hW[(size_t)c * K + k];
            ref = std::max(ref, 0.0);
            maxerr = std::max(maxerr, std::fabs(ref - hC[(size_t)r * N + c]));
        }
        // full-matrix checksum against variant 0 would need identical k order; report sampled error only
        printf("%-60s sampled max err %.3g %s\n", vars[v].name, maxerr, maxerr < 1e-4 ? "ok" : "WRONG");
    }
    const int iters = 4;
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < vars.size(); ++v) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) vars[v].launch(dA, K, dW, K, db, dC, N, M, N, K, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            if (r > 0) { times[v] += ms; best[v] = std::min(best[v], (double)ms); }
        }
    const double flops = 2.0 * M * N * K;
    for (size_t v = 0; v < vars.size(); ++v) {
        const double avg = times[v] / (rounds - 1);
        printf("%-60s avg %.3f ms  %.1f TF/s | best %.3f ms %.1f TF/s\n", vars[v].name, avg, flops / avg / 1e9, best[v], flops / best[v] / 1e9);
    }
    return 0;
}
