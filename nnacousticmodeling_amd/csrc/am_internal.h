// Internal interface between the C-ABI host layer (am_api.hip) and the gfx950 kernels (am_gemm.hip, am_gemm_split.hip, am_frontend.hip,
// am_lstm_step.hip, am_lstm_persist.hip, am_cells.hip, am_viterbi.hip; device-side helpers they share: am_device.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace am {

// kernel ids for the per-kernel event profile (am_profile_read)
enum KernelId {
    K_SPLICE_FT = 0,   // fused splice + (x+shift)*scale + i-vector passthrough
    K_GEMM_HIDDEN = 1, // Linear + bias + activation (MFMA f32)
    K_GEMM_OUT = 2,    // output Linear + bias (+ ensemble scale/accumulate) (MFMA f32)
    K_FINISH = 3,      // combine divide / RPL4 / prior / log-softmax
    K_GATHER_FT = 4,   // recurrent path: gather rows to time-major + transform
    K_LSTM_UPWARD = 5, // hoisted upward GEMM of an LSTM layer (MFMA f32)
    K_LSTM_STEP = 6,   // one wavefront step of the LSTM stack
    K_VITERBI = 7,     // triphone Viterbi decode, one workgroup per utterance
    K_SPLIT_CONVERT = 8, // f32 -> hi/lo halves (split-f16 mode)
    K_GEMM_SPLIT = 9,  // Linear layers on the f16 pipe with hi/lo operands (split-f16 mode)
    K_GEMM_STACK = 10, // the whole Linear stack of a network in one launch (work queue over layers and tiles)
    K_COUNT = 11
};

constexpr int GEMM_BM = 256;  // rows per block tile: every activation matrix is padded to a multiple of this
constexpr int GEMM_BK = 32;   // every K (leading dimension) is padded to a multiple of this
int gemm_n_pad(int n);         // padded output width: multiple of 256, or of 384 when that wastes less

enum Epilogue {
    EPI_ACT = 0,    // C = act(acc + bias)
    EPI_LINEAR = 1, // C = acc + bias
    EPI_SCALE = 2,  // C = (acc + bias) * alpha
    EPI_ACCUM = 3   // C = C + (acc + bias)
};

struct GemmArgs {
    const float* A; int lda;      // [M_pad, lda]  activations, K-contiguous
    const float* W; int ldw;      // [N_pad, ldw]  weights in Chainer [out,in] layout, K-contiguous
    const float* bias;            // [N_pad]
    float* C; int ldc;            // [M_pad, ldc]
    int M_pad, N_pad, K_pad;      // multiples of GEMM_BM / (256 or 384) / GEMM_BK
    int act;                      // AM_ACT_* (EPI_ACT only)
    int epi; float alpha;
    int row_p = 0, row_skip = 0;  // TDNN: output row R reads A row R + (R / row_p) * row_skip (0 = plain GEMM)
    int n_real = 1 << 30;         // EPI_ACT / EPI_LINEAR: output columns >= n_real are padding and are forced to 0 (an infinite input must not turn them into NaN)
    int tiled_units = 0;          // > 0: the output is gate-tiled (LSTM: column -> unit 16 (col / 64) + col % 16) and units >= tiled_units are padding
    int* queue = nullptr;         // work-queue mode: device counter (zero before the launch) the blocks pull tile numbers from
    int n_cus = 256;              // work-queue mode: CUs of the device (grid = one block per CU)
};
hipError_t init_kernels();  // once per device after hipSetDevice
hipError_t launch_gemm(hipStream_t s, const GemmArgs& g);

// The Linear stack of one network as ONE launch: work queue over (layer, row tile, column tile), see gemm_stack_kernel.
constexpr int GEMM_STACK_MAX_HIDDEN = 12;   // (the reference sweeps "ff -l 6..9", scripts/papers/tsd2018/run.sh:8-19)
struct GemmStackLayer {
    const float* W; const float* bias; float* C;   // [N_pad, ldw], [N_pad], [M_pad, ldc]
    int ldw, ldc, tiles_n, nk;                     // column tiles (256 wide; output layer 256 or 192), K stages of 32
    int cum_n;                                     // column tiles of all the layers before this one
};
constexpr int GEMM_STACK_QSTRIDE = 32;             // ints between the eight queue counters (a 128-B line each)
struct GemmStackArgs {
    GemmStackLayer L[GEMM_STACK_MAX_HIDDEN + 1];     // hidden layers 0 .. n_hidden-1 (Linear + bias + act), then the output layer
    const float* A0; int lda0;   // layer 0's input rows (written by an earlier launch)
    int n_hidden, tiles_m, act;
    int out_epi; float out_alpha;
    unsigned* done;              // [n_hidden][tiles_m] finished column tiles per (hidden layer, row tile); zero before the launch
    int* queues;                 // eight item counters GEMM_STACK_QSTRIDE ints apart, zero before the launch: queue x holds the row tiles x, x + 8, ...
    int* abort_flag;             // set when a wait ran into its bound (the pass is then invalid)
    int cum_total;               // column tiles of all layers: queue x has (row tiles of x) * cum_total items
    int test_extra;              // added to every wait's target (0; the give-up test sets 1)
    unsigned long long* stamps;  // diagnostic [workgroups][16] or nullptr (am_ff_stack_stamps): selects the STAMPS instantiation
};
hipError_t launch_gemm_stack(hipStream_t s, const GemmStackArgs& a, int out_n_pad, int n_cus);

// split-f16 precision mode (see gemm_nt_split_kernel): operands are hi/lo half matrices
struct SplitGemmArgs {
    const void* Ahi; const void* Alo; int lda;      // [M_pad, lda] halves
    const void* Whi; const void* Wlo; int ldw;      // [N_pad, ldw] halves, pre-scaled by 1/out_scale
    const float* bias;                              // [N_pad]
    float* C;                                       // output layer: f32 [M_pad, ldc]; nullptr for hidden layers
    void* Chi; void* Clo; int ldc;                  // hidden layers: act(...) as hi/lo halves [M_pad, ldc]
    int M_pad, N_pad, K_pad;                        // multiples of GEMM_BM / 256 / 32
    int act; int epi; float alpha; float out_scale;
    int* overflow;                                  // device flag, set when an activation does not fit f16
    int row_p = 0, row_skip = 0;                    // TDNN: windowed A rows (see GemmArgs)
};
hipError_t launch_gemm_split(hipStream_t s, const SplitGemmArgs& g);
hipError_t launch_split_f32(hipStream_t s, const float* x, void* hi, void* lo, size_t n, float scale, int* overflow);

struct SpliceArgs {
    const float* x; int64_t ext_rows; int64_t first;  // x: [ext_rows, D]; output row r reads x[first + r + shift]
    const float* ivec;                                // [>= row0 + n, d_iv] or nullptr, indexed by (row0 + r)
    int64_t row0; int64_t n; int64_t n_pad;           // rows [row0, row0+n) are real, [n, n_pad) are zero-filled
    int D; int n_shifts; const int* shifts;           // device pointer
    const float* add; const float* scale;             // device [n_shifts*D] or nullptr
    int d_iv; float* out; int ldo;
    void* out_hi; void* out_lo; int* overflow;        // split-f16 mode: write hi/lo halves instead of `out`
};
hipError_t launch_splice_ft(hipStream_t s, const SpliceArgs& a);
// TDNN: channel-last [n_rows * W, Cp] layout of the spliced window (see splice_tdnn_kernel); rows >= a.n are zero
hipError_t launch_splice_tdnn(hipStream_t s, const SpliceArgs& a, int W, int Cp, long long n_rows);

struct FinishArgs {
    const float* in; int ldin;      // [n, ldin]
    float* out; int64_t ldout;      // [*, ldout]
    const int* row_map;             // device [n] destination row (<0 = skip) or nullptr (identity)
    int64_t n; int C;
    float divisor;                  // > 0: v = v / divisor first
    const float* rpl_w; const float* rpl_b; const float* rpl_lb; // RPL4 or nullptr
    const float* prior;             // [C] or nullptr
    int normalise;                  // apply y - logsum(y)
    int accumulate;                 // out += result instead of out = result
    int n_in = 1;                   // > 1: combine the members' logit buffers in + k * in_stride, k < n_in, in member order (finish_kernel)
    int64_t in_stride = 0;          // floats between two members' buffers
    float first_scale = 1.0f;       // member 0 is multiplied by this first (a master's weight)
    int member_norm = 0;            // every member becomes in_k - logsum(in_k) before it is added (mean of log-posteriors)
};
hipError_t launch_finish(hipStream_t s, const FinishArgs& a);

struct GatherArgs {
    const float* x; const float* ivec;  // [N, D], [N, d_iv] or nullptr
    const int* src;                     // device [n] source frame per packed row
    int64_t n; int64_t n_pad;
    int D; const float* add; const float* scale; int d_iv;
    float* out; int ldo;
};
hipError_t launch_gather_ft(hipStream_t s, const GatherArgs& a);

constexpr int LSTM_MAX_LAYERS = 8;
constexpr int LSTM_KB = 128;   // k per batch of the recurrent kernels (both K halves): recurrent widths and inputs are padded to multiples of it
struct LstmLayerStep {   // one layer's time step inside a wavefront launch
    const float* G;      // [>= rows, 4*U_pad] hoisted upward pre-activations (+bias) of this step's rows, or nullptr
    const float* bias;   // [4*U_pad] gate-tiled, used when G == nullptr
    const float* Hin;    // [>= rows, Kin_pad] this step's input rows (previous layer's h at the same t); nullptr with G
    const float* Wup;    // [4*U_pad, Kin_pad] gate-tiled rows
    const float* Hprev;  // [>= rows, U_pad]   previous step's h of this layer (same slot order), unused if first
    const float* Wlat;   // [4*U_pad, U_pad]   gate-tiled rows
    float* Cst;          // [>= slots, U_pad]  cell state of this layer
    float* Hout;         // [>= rows, U_pad]
    int Kin_pad; int n_active; int first;
    int block0;          // set by the launcher: first block of this layer in the launch
    int tpb;             // set by the launcher: 16-row tiles per block (1..4)
};
struct LstmWavefrontArgs {
    LstmLayerStep L[LSTM_MAX_LAYERS];
    int n;               // layers active in this launch
    int U_pad;           // common to all layers (multiple of 64)
    int ut_groups;       // set by the launcher
};
hipError_t launch_lstm_wavefront(hipStream_t s, LstmWavefrontArgs& a);

// The whole LSTM stack over the whole time axis in ONE persistent launch (lstm_persist_kernel): one workgroup per CU keeps
// its slice of the weights in registers for the entire pass and exchanges h / c with the other workgroups through
// agent-scope (sc1) stores + per-group arrival counters instead of kernel boundaries.
struct LstmPersistLayer {
    const float* G;      // layer 0: hoisted upward pre-activations (+bias) [NP, 4*U_pad], time-major packed; else nullptr
    const float* bias;   // layers >= 1: [4*U_pad] gate-tiled
    const float* Wup;    // layers >= 1: [4*U_pad, U_pad] gate-tiled rows
    const float* Wlat;   // [4*U_pad, U_pad] gate-tiled rows
    const float* Hin;    // layers >= 1: packed H plane of the layer below [>= NP, U_pad]
    float* H;            // this layer's packed H plane
    float* Cst;          // this layer's cell-state plane, time-major packed like H [>= NP, U_pad] (written once per address and launch)
    int n_groups;        // row groups of this layer
    int group0;          // index of its first group among all groups (counter index)
};
struct LstmPersistArgs {
    LstmPersistLayer L[LSTM_MAX_LAYERS];
    int n_layers, U_pad, T;
    int t_begin, t_end;            // the steps [t_begin, t_end) of this launch (t_end == 0: up to T).  A launch that starts at t_begin > 0 finds h, c and
                                   // the arrival counters as the launch for the steps before it left them
    int n_groups_total;            // grid = n_groups_total * (U_pad / 16)
    int max_wgs;                   // workgroups of this kernel the device holds at once (lstm_persist_resident_wgs): the plan's budget
    int slot_base, slot_cap;       // this launch covers the utterance slots [slot_base, slot_base + slot_cap) (longest first: a chunk of a larger file)
    int ext_upward;                // 1: L[0] is not the stack's first layer -- it multiplies an upward input plane (L[0].Hin) that an EARLIER launch has
                                   // completed (a deep stack run as several launches over layer ranges), instead of adding hoisted rows (L[0].G)
    const int* n_active;           // device [T]
    const int* step_off;           // device [T + 1]
    const unsigned* cum;           // device [n_groups_total][T]: arrivals group g has made once its step t is complete
    unsigned* counters;            // device [n_groups_total], zero before the launch
    int* abort_flag;               // device, zero before the launch; set when a wait timed out (the pass is then invalid)
    unsigned long long* stamps;    // diagnostic timeline [n_groups_total][T][8] of 100 MHz ticks, or nullptr (the normal case)
};
// Host-side plan: fills n_groups / group0 / n_groups_total for `n_slots` utterance slots and returns false when the
// stack does not fit the persistent form (then the per-step kernels run).  cum_out: [n_groups_total][T] arrival targets.
bool plan_lstm_persist(LstmPersistArgs& a, int n_slots, const int* n_active_host, unsigned* cum_out /*nullable*/);
size_t lstm_persist_cum_size(const LstmPersistArgs& a);   // entries of `cum`
int lstm_persist_resident_wgs(int U_pad, int n_cus);      // co-resident workgroups of the kernel on a device of n_cus CUs (0: none)
int lstm_persist_capacity(int U_pad, int n_layers, int max_wgs, bool ext_upward = false);   // utterance slots one launch can take (0: this shape has no persistent form)
hipError_t launch_lstm_persist(hipStream_t s, const LstmPersistArgs& a);

int lstm_persist_mode();   // AM_LSTM_PERSIST: 0 per-step kernels, 1 (default) the persistent form

// The persistent form for stacks whose weights do not fit the register file (lstm_stream_kernel: 768 / 1024 units): the same arguments, workgroup =
// (layer, 16-unit tile) taking all active rows, weights streamed per step; counters[l] = arrivals of layer l (U_pad / 16 per step); `cum`, the group
// fields and `stamps` are unused.
int lstm_stream_resident_wgs(int U_pad, int n_cus);       // co-resident workgroups of the kernel (0: this shape has no streamed form)
int lstm_stream_layers(int U_pad, int max_wgs);           // layers one launch can take
hipError_t launch_lstm_stream(hipStream_t s, const LstmPersistArgs& a);

// Recurrent cells other than L.LSTM (GRU / MGRU / peephole LSTM): one phase of one wavefront tick, see rec_step_kernel.
enum RecEpilogue {
    REC_MGRU = 0,    // 2 gates (z, h~):   h = z*act(h~) + (1-z)*h_prev                               (MGRU.py:67-85, no reset gate)
    REC_GRU_A = 1,   // 2 gates (r, z):    Z = sig(z), RH = sig(r)*h_prev                             (phase A of a cell with reset gate)
    REC_GRU_B = 2,   // 1 gate  (h~):      h = Z*act(h~) + (1-Z)*h_prev   with h~ over [x | RH]       (phase B)
    REC_PEEP_A = 3,  // 4 gates (a,i,f,o): c' = tanh(a)*sig(i) + sig(f)*c ; O = o   over [x | h | c]   (phase A of StatefulPeepholeLSTM)
    REC_PEEP_B = 4,  // 1 gate  (peep_o):  h = sig(O + peep_o c') * tanh(c')                          (phase B)
    REC_XPART = 5    // no gates: the accumulators of the input part of a phase's product, stored as they are (rec_stream_kernel only)
};
struct RecLayerStep {
    const float* A[3]; int K[3];   // K segments of the concatenated operand, each [>= 16-row tiles, K[s]] rows (ld = K[s]); K multiples of 64
    const float* W; int ldw;       // packed weights [G*U_pad, ldw]: row = 16*tile + gate*(16/G) + unit-in-tile, columns = the segments in order
    const float* Ws;               // the same weights in stream order (rec_stream_pack_weights): what rec_wide_kernel reads
    const float* bias;             // [G*U_pad] in the same row order
    const float* E0; const float* E1;  // element-wise inputs  [slots, U_pad] (see RecEpilogue)
    float* O0; float* O1;              // element-wise outputs [slots, U_pad]
    int n_active; int first;
    int block0, tpb;               // set by the launcher of rec_wide_kernel
};
struct RecStepArgs {
    RecLayerStep L[LSTM_MAX_LAYERS];
    int n; int U_pad; int act;     // act: AM_ACT_* of the candidate state (GRU: tanh, MGRU-relu: relu)
    int U_real;                    // units >= U_real are padding: their outputs are forced to 0 (see gemm_epilogue)
    int bt_groups;                 // set by the launcher of rec_wide_kernel
};
hipError_t launch_rec_step(hipStream_t s, int epilogue, const RecStepArgs& a);

// The same cells as ONE persistent launch (rec_stream_kernel, am_cells_stream.hip): a role = one phase of one layer.
constexpr int RS_MAX_ROLES = 16;
struct RecStreamSeg {
    const float* A; int ld;      // time-major plane [rows, ld] the segment's rows come from
    int nb;                      // 128-k batches (0: no such segment)
    int prev;                    // 1: rows of step t - 1; 0: rows of step t
    int skip_first;              // 1: the segment does not exist at t == 0 (the state Linears of a cell without a previous state)
    int dep; unsigned dep_mult;  // the role (counter index) that produces those rows in this launch, and its arrivals per step; -1: the plane was complete before the launch
    int wb0;                     // the segment's first 128-k batch in the role's weights
};
struct RecStreamRole {
    RecStreamSeg seg[3];
    const float* W; int ldw;                       // packed weights [G U_pad, ldw] in stream order (rec_stream_pack_weights)
    const float* bias_first; const float* bias_rest;
    const float* E0; const float* E1;              // element-wise inputs, time-major planes [rows, U_pad] (rec_cell_math); E0 of MGRU / GRU / PEEP_A: rows of step t - 1
    float* O0; float* O1;                          // element-wise outputs
    int edep; unsigned edep_mult;                  // role whose step-t output the epilogue reads when no tile wait covers it (GRU B at t == 0: Z); -1: none
    int epi, n_col, wg0, counter;                  // RecEpilogue; 64-column workgroups of the role; first workgroup (within a group); counter index
    // A phase split over two roles: the input part (x W_x^T, epi = REC_XPART, no dependency on the recurrence: it runs ahead) leaves its accumulators
    // -- per K half, in fragment order -- in `xplane`; the state part starts its accumulators from them, so the sums are folded in the same order
    // as by one role over both segments.  [tile_off[t] + tile][n_col][8 waves][64 lanes][4] floats.
    float* xplane;                                 // REC_XPART: where the accumulators go
    const float* init; int idep; unsigned idep_mult;   // state part: the plane its accumulators start from (nullptr: zeros), the role that writes it
};
struct RecStreamArgs {
    const RecStreamRole* roles;                    // device [n_roles]
    int n_roles, NG, wgs_per_group, T, U_pad, U_real, act, max_wgs;
    const int* n_active; const int* step_off;      // device [T], [T + 1]
    const int* tile_off;                           // device [T]: 16-row tiles of the steps before t (the split phases' accumulator planes)
    unsigned* counters;                            // device [n_roles][NG], zero before the launch
    int* abort_flag;
    int dbg;                                       // measurement switches (AM_REC_DBG): 1 = every wait polls, 2 = groups in workgroup-id order; with stamps:
                                                   // 128 = lane 0's accumulator quads of the split phases instead of the poll times, >> 8 = the step whose batch loop is traced
    unsigned long long* stamps;                    // diagnostic timeline [n_roles][T][8] of 100 MHz ticks, or nullptr (the normal case)
};
int rec_stream_resident_wgs(int n_cus);
void rec_stream_pack_weights(const float* w, int rows, int ld, float* out);   // host: row-major [rows, ld] -> the kernel's stream order
hipError_t launch_rec_stream(hipStream_t s, const RecStreamArgs& a);

struct ViterbiArgs {
    int P, B, T;                       // phones (start/end nodes), biphone nodes, triphones
    int compact;                       // 1: the records are in the compact layout of viterbi_compact_kernel (tri_idx = c0, tri_self = c1)
    int perm, inv0;                    // perm = 1: the compact records (and all per-triphone state) are stored in arrival-list order, internal
                                       // index s <-> reference triphone outVal[s]; inv0 = internal index of the reference's triphone 0
    const int* endIdx;                 // [P]
    // one 16-B record per triphone n (packed at load for coalesced access):
    const __attribute__((ext_vector_type(4))) int* tri_idx;     // {node it leaves from, posterior column of state 0, 1, 2}
    const __attribute__((ext_vector_type(4))) float* tri_self;  // {self-loop log-prob of state 0, 1, 2, 0}
    const __attribute__((ext_vector_type(4))) float* tri_trans; // {transition log-prob of state 0, 1, 2, 0}
    const int* outOff; const int* outVal;        // CSR: triphones arriving at each node
    const float* lm;                   // [B] already scaled by LMW
    const float* p0;                   // [B] initial node scores
    float PIP;
    const float* lp; long long ldl;    // log-probabilities [N, ldl]
    const long long* offsets;          // [n_utts + 1] device
    float* pp; int* tt;                // per-utterance state: [n_utts][T] x 16 B (3 state scores / 3 entry times)
    float* pout; float* p;             // per-utterance: [n_utts][T] triphone exit scores, [n_utts][B] node scores
    int* bestI; int* bestT;            // [N][B] back-pointers
    int* seg_begin; int* seg_end; int* seg_tri;  // [N] segments of utterance u at [offsets[u], offsets[u] + seg_count[u]), last first
    int* seg_count;                    // [n_utts]
    const int* slot_utt;               // frame-synchronous form: utterance of each slot (slots sorted by length, longest first)
    const long long* slot_row0;        // ... and the row of its first frame, offsets[slot_utt[slot]] (one scalar load instead of two dependent ones)
    int n_cus;                         // CUs of the device (viterbi_fs_stream_kernel sizes its grid to the blocks the device holds at once)
    // frame-synchronous form: the node scores ping-pong between `p` (read by even frames, written by odd ones) and `p2` (the reverse), so that a
    // launch that updates triphones AND takes the node maxima never overwrites a score another block of the same frame still has to read
    float* p2;
    // fused form (viterbi_fs_fused_kernel; perm == 1): block b covers the whole arrival lists of nodes [blk_node0[b], blk_node0[b+1]) =
    // triphones [blk_tri0[b], blk_tri0[b+1]) (at most 2,048); inv0_state: [n_utts][2][8] ints, the state of triphone inv0 BEFORE frame k at parity k & 1
    const int* blk_tri0; const int* blk_node0; int n_blk;
    int blk_nodes_max;                 // most nodes any block covers (viterbi_fs_stream_kernel keeps their list bounds in LDS)
    int* inv0_state;
};
hipError_t launch_viterbi(hipStream_t s, const ViterbiArgs& a, int n_utts);
// frame-synchronous form (compact records only): init, one (triphone update + node maxima) launch pair per frame k over the
// first n_active slots, back-trace of every utterance
hipError_t launch_viterbi_fs_init(hipStream_t s, const ViterbiArgs& a, int n_utts);
hipError_t launch_viterbi_fs_step(hipStream_t s, const ViterbiArgs& a, int k, int n_active);
hipError_t launch_viterbi_fs_trace(hipStream_t s, const ViterbiArgs& a, int n_utts);

}  // namespace am
