// gfx950 (MI355X / CDNA4) kernels of the acoustic-model posterior engine: splice + feature transform, the finishing pass (log-softmax etc.), the recurrent path's gather, the TDNN splice
#include "am_device.h"

namespace am {

// ---------------------------------------------------------------------------------------------------
// Fused splice + Kaldi feature transform + i-vector passthrough.  HBM-bound: every output element is written
// once (coalesced, 16 B per lane), every input row is read n_shifts times out of L1/L2.
// out[r, w*D + d] = (x[clamp(first + r + shift_w, 0, ext_rows-1), d] + add[w*D+d]) * scale[w*D+d]
// out[r, n_shifts*D + j] = ivec[row0 + r, j];  remaining columns up to ldo and rows [n, n_pad) are zero.
// VEC = 4 needs D % 4 == 0 and d_iv % 4 == 0 (then no float4 straddles a segment); VEC = 1 is the general form.
// ---------------------------------------------------------------------------------------------------
// HALVES: write the result as hi/lo f16 halves (split-f16 precision mode: hi at out_hi, lo at out_lo, same [n_pad, ldo] shape)
// instead of f32, raising *overflow when a value does not fit f16.
template <int VEC, bool HALVES>
__global__ __launch_bounds__(256) void splice_ft_kernel(
    const float* __restrict__ x, long long ext_rows, long long first, const float* __restrict__ ivec, long long row0,
    long long n, long long n_pad, int D, int n_shifts, const int* __restrict__ shifts,
    const float* __restrict__ add, const float* __restrict__ scale, int d_iv, float* __restrict__ out, int ldo,
    int rows_per_block, _Float16* __restrict__ out_hi, _Float16* __restrict__ out_lo, int* __restrict__ overflow)
{
    const int Q = ldo / VEC;                 // vector columns per row
    const int Qt = Q < 256 ? Q : 256;        // threads form a [rows_per_pass][Qt] grid; surplus threads idle
    const int rows_per_pass = 256 / Qt;
    const int tid = threadIdx.x;
    const int rsub = tid / Qt;
    if (rsub >= rows_per_pass) return;
    const long long rblock = (long long)blockIdx.x * rows_per_block;
    const int spl = n_shifts * D;

    for (int q = tid - rsub * Qt; q < Q; q += Qt) {
        const int col = q * VEC;
        int w = 0, d = 0, kind;  // 0 = spliced feature, 1 = i-vector, 2 = zero pad
        int shift = 0;
        if (col < spl) { kind = 0; w = col / D; d = col - w * D; shift = shifts[w]; }
        else if (col < spl + d_iv) { kind = 1; d = col - spl; }
        else kind = 2;
        float av[VEC], sv[VEC];
        const bool has_ft = (add != nullptr);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { av[e] = (kind == 0 && has_ft) ? add[col + e] : 0.0f; sv[e] = (kind == 0 && has_ft) ? scale[col + e] : 1.0f; }

        // the common case first: the spliced feature columns of a block of real rows, four rows' loads in flight per lane before
        // the first store (one row at a time left the kernel at 3.6 TB/s, latency-bound: MI355X_MICROARCH.md, keep >= 4 loads per lane)
        int rr0 = rsub;
        if (!HALVES && VEC == 4 && kind == 0 && rblock + rows_per_block <= n) {
            for (; rr0 + 3 * rows_per_pass < rows_per_block; rr0 += 4 * rows_per_pass) {
                f32x4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    long long src = first + rblock + rr0 + u * rows_per_pass + shift;
                    src = src < 0 ? 0 : (src > ext_rows - 1 ? ext_rows - 1 : src);
                    t[u] = *(const f32x4*)(x + src * D + d);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (has_ft) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float s1 = t[u][e] + av[e]; t[u][e] = s1 * sv[e]; }
                    }
                    *(f32x4*)(out + (rblock + rr0 + u * rows_per_pass) * ldo + col) = t[u];
                }
            }
        }
        for (int rr = rr0; rr < rows_per_block; rr += rows_per_pass) {
            const long long r = rblock + rr;
            if (r >= n_pad) break;
            float v[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = 0.0f;
            if (r < n) {
                if (kind == 0) {
                    long long src = first + r + shift;
                    src = src < 0 ? 0 : (src > ext_rows - 1 ? ext_rows - 1 : src);
                    const float* p = x + src * D + d;
                    if (VEC == 4) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                    else v[0] = p[0];
                    if (has_ft) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) { const float s1 = v[e] + av[e]; v[e] = s1 * sv[e]; }
                    }
                } else if (kind == 1) {
                    const float* p = ivec + (row0 + r) * d_iv + d;
                    if (VEC == 4) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                    else v[0] = p[0];
                }
            }
            if (HALVES) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                _Float16 hh[VEC], ll[VEC];
                bool bad = false;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    hh[e] = (_Float16)v[e];
                    ll[e] = (_Float16)(v[e] - (float)hh[e]);
                    bad |= !(fabsf(v[e]) <= SPLIT_F16_LIMIT);
                }
                if (VEC == 4) {
                    *(h4*)(out_hi + r * ldo + col) = (h4){hh[0], hh[1], hh[2], hh[3]};
                    *(h4*)(out_lo + r * ldo + col) = (h4){ll[0], ll[1], ll[2], ll[3]};
                } else { out_hi[r * ldo + col] = hh[0]; out_lo[r * ldo + col] = ll[0]; }
                if (bad) *overflow = 1;
            } else {
                float* o = out + r * ldo + col;
                if (VEC == 4) { f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *(f32x4*)o = t; }
                else o[0] = v[0];
            }
        }
    }
}

hipError_t launch_splice_ft(hipStream_t s, const SpliceArgs& a)
{
    if (a.n_pad <= 0) return hipSuccess;
    const bool vec = (a.D % 4 == 0) && (a.d_iv % 4 == 0) && (a.ldo % 4 == 0);
    const int rows_per_block = 32;
    const long long blocks = (a.n_pad + rows_per_block - 1) / rows_per_block;
#define AM_SPLICE(V, H) hipLaunchKernelGGL((splice_ft_kernel<V, H>), dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first, \
        a.ivec, (long long)a.row0, (long long)a.n, (long long)a.n_pad, a.D, a.n_shifts, a.shifts, a.add, a.scale, a.d_iv, a.out, a.ldo, rows_per_block, \
        (_Float16*)a.out_hi, (_Float16*)a.out_lo, a.overflow)
    if (a.out_hi) { if (vec) AM_SPLICE(4, true); else AM_SPLICE(1, true); }
    else { if (vec) AM_SPLICE(4, false); else AM_SPLICE(1, false); }
#undef AM_SPLICE
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Finish: one wavefront per frame; lane l holds columns l, l+64, ... in registers (256-B coalesced dword
// accesses).  v -> v/divisor -> RPL4 (log_softmax, x + x*W + b, log-add-exp with floor lb) -> v - prior ->
// v - logsum(v), logsum = mx + log(sum exp(v - mx)) with NaN -> -1e20 (reference kw_utils.py:38-43).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// PATCH: kw_utils.logsum's `lps[isnan(lps)] = -1e20` (kw_utils.py:42).  RPL4's F.log_softmax has no such rule (RPL.py:69): a row
// holding a NaN or a +inf comes out of it all-NaN, as in the reference.
template <int NIT, bool PATCH = true>
__device__ __forceinline__ float row_logsum(const float (&v)[NIT], int lane, int C)
{
    // np.max propagates NaN; fmaxf does not, but a NaN element makes the exp-sum NaN, which is all that matters.
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (it * 64 + lane < C) m = fmaxf(m, v[it]);
    m = wave_max(m);
    float sacc = 0.0f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (it * 64 + lane < C) sacc += expf(v[it] - m);
    sacc = wave_sum(sacc);
    float lse = m + logf(sacc);
    if (PATCH && lse != lse) lse = -1e20f;
    return lse;
}

// n_in > 1: the members of an ensemble left their logits in n_in buffers (in + k * in_stride); they are combined here in member order with
// the float operations of the one-buffer path -- v = in_0 (* first_scale: a master's weight, NNWithRPL); v = v + in_k -- or, with
// member_norm (the dev-mode mean of log-posteriors, predict_folds.py:207-212), v = (in_0 - logsum(in_0)); v = v + (in_k - logsum(in_k)).
// (MULTI is an instantiation of its own: its second row of registers costs the one-buffer path occupancy -- 530 -> 622 us on the headline's
// 196,608 x 1909 rows when both lived in one kernel)
template <int NIT, bool MULTI>
__global__ __launch_bounds__(256) void finish_kernel(
    const float* __restrict__ in, int ldin, float* __restrict__ out, long long ldout, const int* __restrict__ row_map,
    long long n, int C, float divisor, const float* __restrict__ rpl_w, const float* __restrict__ rpl_b,
    const float* __restrict__ rpl_lb, const float* __restrict__ prior, int normalise, int accumulate,
    int n_in, long long in_stride, float first_scale, int member_norm)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    long long orow = row;
    if (row_map) { orow = row_map[row]; if (orow < 0) return; }

    float v[NIT];
    const float* src = in + row * ldin;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; v[it] = c < C ? src[c] : 0.0f; }
    if (MULTI && n_in > 1) {
        if (first_scale != 1.0f) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] * first_scale;
        }
        if (member_norm) {
            const float lse = row_logsum<NIT>(v, lane, C);
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] - lse;
        }
        for (int k = 1; k < n_in; ++k) {
            float w[NIT];
            const float* sk = src + (long long)k * in_stride;
#pragma unroll
            for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; w[it] = c < C ? sk[c] : 0.0f; }
            if (member_norm) {
                const float lse = row_logsum<NIT>(w, lane, C);
#pragma unroll
                for (int it = 0; it < NIT; ++it) w[it] = w[it] - lse;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = v[it] + w[it];
        }
    }

    if (divisor > 0.0f) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = v[it] / divisor;
    }
    if (rpl_w) {
        const float lse = row_logsum<NIT, false>(v, lane, C);  // F.log_softmax: no NaN patch-up
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < C) {
                const float x = v[it] - lse;
                const float xw = x * rpl_w[c];
                const float h = (x + xw) + rpl_b[c];
                const float lb = rpl_lb[c];
                const float mx = am_max_nan(h, lb), mn = am_min_nan(h, lb);   // F.maximum / F.minimum: NaN in, NaN out
                v[it] = mx + logf(1.0f + expf(mn - mx));
            }
        }
    }
    if (prior) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int c = it * 64 + lane; if (c < C) v[it] = v[it] - prior[c]; }
    }
    if (normalise) {
        const float lse = row_logsum<NIT>(v, lane, C);
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = v[it] - lse;
    }
    float* dst = out + orow * ldout;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < C) dst[c] = accumulate ? dst[c] + v[it] : v[it];
    }
}

hipError_t launch_finish(hipStream_t s, const FinishArgs& a)
{
    if (a.n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((a.n + 3) / 4);
    const int nit = (a.C + 63) / 64;
#define AM_FINISH_ARGS a.in, a.ldin, a.out, (long long)a.ldout, a.row_map, (long long)a.n, a.C, a.divisor, a.rpl_w, a.rpl_b, a.rpl_lb, a.prior, a.normalise, \
        a.accumulate, a.n_in, (long long)a.in_stride, a.first_scale, a.member_norm
#define AM_FINISH(N) do { if (a.n_in > 1) hipLaunchKernelGGL((finish_kernel<N, true>), dim3(blocks), dim3(256), 0, s, AM_FINISH_ARGS); \
                          else hipLaunchKernelGGL((finish_kernel<N, false>), dim3(blocks), dim3(256), 0, s, AM_FINISH_ARGS); } while (0)
    if (nit <= 1) AM_FINISH(1);
    else if (nit <= 2) AM_FINISH(2);
    else if (nit <= 4) AM_FINISH(4);
    else if (nit <= 8) AM_FINISH(8);
    else if (nit <= 16) AM_FINISH(16);
    else if (nit <= 30) AM_FINISH(30);
    else if (nit <= 32) AM_FINISH(32);
    else return hipErrorInvalidValue;
#undef AM_FINISH
#undef AM_FINISH_ARGS
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent path, stage 1: gather file-order frames into the time-major packed order + per-dim transform.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_ft_kernel(
    const float* __restrict__ x, const float* __restrict__ ivec, const int* __restrict__ src, long long n, long long n_pad,
    int D, const float* __restrict__ add, const float* __restrict__ scale, int d_iv, float* __restrict__ out, int ldo)
{
    const long long total = n_pad * ldo;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / ldo;
        const int c = (int)(idx - r * ldo);
        float v = 0.0f;
        if (r < n) {
            const long long f = src[r];
            if (c < D) {
                v = x[f * D + c];
                if (add) { const float s1 = v + add[c]; v = s1 * scale[c]; }
            } else if (c < D + d_iv) {
                v = ivec[f * d_iv + (c - D)];
            }
        }
        out[idx] = v;
    }
}

hipError_t launch_gather_ft(hipStream_t s, const GatherArgs& a)
{
    if (a.n_pad <= 0) return hipSuccess;
    long long total = a.n_pad * a.ldo;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_ft_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.x, a.ivec, a.src, (long long)a.n, (long long)a.n_pad,
                       a.D, a.add, a.scale, a.d_iv, a.out, a.ldo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// TDNN frontend.  The reference splices +-s frames, applies the (tiled centre-frame) transform and then VIEWS each row
// of W*D values as [C = D channels][W positions] (chainer_networks.py:39: F.reshape(x, (B, -1, 1, input_win_size))), i.e.
// channel c, position p is flat element f = c*W + p of the spliced row -- which came from frame shift f / D, dimension
// f % D.  The convolutions run here as GEMMs over channel-last activations, so this kernel writes
//   out[(n*W + p) * Cp + c] = (x[clamp(first + n + shift[f / D]), f % D] + add[f]) * scale[f],   f = c*W + p
// with zero channels c >= D (Cp = D rounded up to 32) and zero rows for n >= n_real.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splice_tdnn_kernel(
    const float* __restrict__ x, long long ext_rows, long long first, long long n_real, long long n_rows, int D, int W,
    const int* __restrict__ shifts, const float* __restrict__ add, const float* __restrict__ scale, float* __restrict__ out, int Cp)
{
    const long long total = n_rows * W * Cp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % Cp);
        const long long np = idx / Cp;
        const int p = (int)(np % W);
        const long long n = np / W;
        float v = 0.0f;
        if (c < D && n < n_real) {
            const int f = c * W + p;
            long long r = first + n + shifts[f / D];
            r = r < 0 ? 0 : (r >= ext_rows ? ext_rows - 1 : r);
            v = x[r * D + f % D];
            if (add) { const float s1 = v + add[f]; v = s1 * scale[f]; }
        }
        out[idx] = v;
    }
}

hipError_t launch_splice_tdnn(hipStream_t s, const SpliceArgs& a, int W, int Cp, long long n_rows)
{
    if (n_rows <= 0) return hipSuccess;
    long long blocks = (n_rows * W * Cp + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(splice_tdnn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.x, (long long)a.ext_rows, (long long)a.first,
                       (long long)a.n, n_rows, a.D, W, a.shifts, a.add, a.scale, a.out, Cp);
    return hipGetLastError();
}

}  // namespace am
