// Device-side helpers shared by the kernel translation units (am_gemm.hip, am_gemm_split.hip, am_frontend.hip, am_lstm_step.hip,
// am_lstm_persist.hip, am_cells.hip, am_viterbi.hip).
//
// All buffers are laid out by the host layer so that the kernels need no bounds checks in their main loops:
// activation matrices have M padded to 256 rows (GEMM_BM) and their leading dimension padded to the next layer's K_pad,
// weights are zero-padded to [N_pad, K_pad].  Wavefront = 64 lanes everywhere.
#pragma once
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

// ReLU / max / min as numpy's np.maximum / np.minimum (what Chainer's F.relu, F.maximum, F.minimum call: chainer_networks.py:19-22,
// RPL.py:71-72): a NaN operand gives NaN.  fmaxf / fminf return the other operand instead, which would turn a frame the reference
// reports as NaN into a plausible-looking distribution.  IEEE 754-2019 maximum / minimum: one v_maximum3_f32 / v_minimum3_f32 on gfx950.
__device__ __forceinline__ float am_max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float am_min_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float am_relu(float v) { return am_max_nan(v, 0.0f); }

__device__ __forceinline__ float am_sigmoid(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }

__device__ __forceinline__ float am_activate(float v, int act) {
    if (act == 0) return am_relu(v);
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
    if (act == 0) return am_relu(v);
    if (act == 1) return am_sigmoid_rec(v);
    return am_tanh_rec(v);
}

// 16-B accesses of rows another workgroup of the SAME launch stored / will load (hand-offs of the persistent kernels): write-through stores,
// L1-bypassing loads (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ f32x4 ld_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */));
}

__device__ __forceinline__ void st_sc1_b128(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, 16 /* sc1 */);
}

// The element-wise arithmetic of the cells other than L.LSTM, one phase (RecEpilogue) of one unit: shared by the per-tick kernels (am_cells.hip) and the
// persistent form (am_cells_stream.hip), so that a frame's posteriors do not depend on the form that computed them.
//   MGRU / GRU_B: linear_interpolate(z, h~, h) = z h~ + (1 - z) h; without a previous state h = z h~ (MGRU.py:77-84).  e0 = h_prev, e1 = Z (GRU_B)
//   GRU_A       : o0 = Z = sig(z), o1 = sig(r) h_prev                                                                   e0 = h_prev
//   PEEP_A      : o0 = c' = tanh(a) sig(i) + sig(f) c, o1 = o before its peephole term                                  e0 = c_prev
//   PEEP_B      : o0 = h = sig(O + peep_o c') tanh(c')                                                                  e0 = c', e1 = O
template <int G, int EPI>
__device__ __forceinline__ void rec_cell_math(const float (&pre)[G], float e0, float e1, int first, int act, float& o0, float& o1)
{
    if (EPI == REC_MGRU || EPI == REC_GRU_B) {
        const float z = EPI == REC_MGRU ? am_sigmoid_rec(pre[0]) : e1;
        const float hb = am_activate_rec(pre[G - 1], act);
        float h = z * hb;
        if (!first) { const float keep = (1.0f - z) * e0; h = h + keep; }
        o0 = h;
    } else if (EPI == REC_GRU_A) {
        o0 = am_sigmoid_rec(pre[G - 1]);
        o1 = first ? 0.0f : am_sigmoid_rec(pre[0]) * e0;
    } else if (EPI == REC_PEEP_A) {
        const float a = am_tanh_rec(pre[0]);
        const float i = am_sigmoid_rec(pre[1 % G]);
        const float f = am_sigmoid_rec(pre[2 % G]);
        const float ai = a * i;
        const float fc = f * e0;
        o0 = ai + fc;
        o1 = pre[3 % G];
    } else {
        const float o = am_sigmoid_rec(e1 + pre[0]);
        o0 = o * am_tanh_rec(e0);
    }
}

// geometry shared by the GEMM kernels (am_gemm.hip, am_gemm_split.hip): 512-thread blocks, 8 waves as 4 (M) x 2 (N)
constexpr int GEMM_WM = 4, GEMM_WN = 2, GEMM_THREADS = 64 * GEMM_WM * GEMM_WN;

// geometry shared by the recurrent kernels (am_lstm_step.hip, am_lstm_persist.hip, am_cells.hip)
constexpr int LSTM_RB = 64;               // rows per block
// (LSTM_KB = 128, k per batch (both K halves): am_internal.h -- the host plans in batches too)
constexpr int LSTM_KS = LSTM_KB / 32;     // 16-k sub-steps per wave per batch
constexpr int LSTM_LDA = LSTM_KB + 8;     // LDS row stride (floats): +32 B -> ds_read_b128 of (row l15, chunk lq) is conflict-free in every 16-lane service
                                          // group (+16 B left a 2-way conflict per group: SQ_LDS_BANK_CONFLICT, profiles/r2b)
constexpr int LSTM_APT = LSTM_RB * LSTM_KB / 4 / 512;  // 16-B A chunks per thread per batch

}  // namespace am
