/*
 * amfwd.h -- C ABI of libamfwd.so: MI355X (gfx950) acoustic-model posterior engine.
 *
 * Drop-in boundary for the posterior-compute path of OrcusCZ/NNAcousticModeling.  The reference has no
 * FFI of its own: the functions below replace, for that path only, what the reference does
 * through Chainer/numpy calls.  Each entry point cites the reference code it stands in for (paths relative
 * to the reference repo root).  Plain C types only; no framework types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative AM_E* code; am_last_error() gives a thread-local
 *     message for the calling thread's last failure;
 *   - all host arrays are caller-owned, row-major float32 unless stated; the library copies what it keeps;
 *   - a handle (engine/model/pipeline) is bound to one device and is NOT re-entrant; different engines may be
 *     driven from different host threads (one per GPU);
 *   - there is no CPU fallback: without a usable gfx950 device am_engine_create() fails.
 */
#ifndef AMFWD_H
#define AMFWD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AM_OK 0
#define AM_EINVAL (-22)  /* bad argument / bad state          */
#define AM_ENODEV (-19)  /* no usable GPU                     */
#define AM_ENOMEM (-12)  /* host or device allocation failed  */
#define AM_EHIP (-5)     /* a HIP runtime call failed         */

typedef struct am_engine am_engine;     /* one device: stream, events, workspace            */
typedef struct am_model am_model;       /* one network's weights, resident on the device    */
typedef struct am_pipeline am_pipeline; /* frontend + member models + epilogue + I/O buffers */

/* network kinds: chainer_networks.py:163-184 (get_nn) -- only the in-scope ones */
enum { AM_NET_FF = 0, AM_NET_LSTM = 1 };
/* FF activation: evaluate.py:96-103 / predict_folds.py:144-152 */
enum { AM_ACT_RELU = 0, AM_ACT_SIGMOID = 1, AM_ACT_TANH = 2 };
/* how member outputs are combined */
enum {
    AM_COMBINE_LOGITS = 0,   /* NNWithRPL.__call__, evaluate.py:35-51: raw-logit averaging          */
    AM_COMBINE_LOGPOST = 1   /* predict_folds.py:199-220: mean of per-member log-posteriors, renormalised */
};

/* ---- discovery / errors ------------------------------------------------------------------------- */
int am_device_count(void);            /* number of visible HIP devices, 0 if none (never negative) */
/* PCI bus id ("0000:c1:00.0") of a visible device: two ranks that were handed the same physical card (predict_folds.py:158-161 /
 * evaluate.py's --gpu names a card by index only) can tell.  buflen >= 16.  No engine needed. */
int am_device_pci_bus_id(int device, char* buf, int buflen);
const char* am_last_error(void);      /* thread-local, never NULL                                   */
/* Host-only helper, exposed so that the layout can be pinned by a test without a GPU: the order in which the kernels that stream weight
 * fragments every step (rec_wide_kernel, lstm_stream_kernel, rec_stream_kernel) read a packed recurrent weight matrix w[rows, ld]
 * (rows a multiple of 64 = 4 MFMA column tiles, ld a multiple of 128 = one k batch): out[((((rows/64 block)*ld/128 + batch)*8 + wave)*4 + sub-step)*64 + lane][4]
 * = w[block*64 + (wave&3)*16 + (lane&15)][batch*128 + (wave>>2)*64 + sub-step*16 + (lane>>4)*4 ..+4] -- the fragment of one wave's
 * v_mfma_f32_16x16x4_f32 B operand, so that one load instruction covers 1 KB of consecutive memory.  No reference counterpart (Chainer keeps
 * chainer.links.Linear.W as [out, in], chainer_networks.py:33-52; this is a device-side copy).  Returns AM_EINVAL on a shape it does not take. */
int am_pack_stream_order(const float* w, int rows, int ld, float* out);
const char* am_version(void);

/* ---- engine --------------------------------------------------------------------------------------- */
/* replaces chainer.cuda.get_device_from_id(gpu).use() / model.to_gpu() (predict_folds.py:158-161,
 * evaluateModelForTest.py:47-50).  Fails with AM_ENODEV when `device` is not a gfx950 GPU. */
int am_engine_create(int device, am_engine** out);
/* Models, pipelines and decoders keep their engine alive: destroying the engine first only closes it (nothing new can be
 * created on it); the device resources go when its last child is destroyed.  Handles may be destroyed in any order. */
int am_engine_destroy(am_engine* e);
int am_engine_synchronize(am_engine* e);
int am_engine_device_name(am_engine* e, char* buf, int buflen);

/* Arithmetic of the feed-forward Linear layers.
 *   AM_PRECISION_F32 (default): exact f32 MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit an fmaf chain.
 *   AM_PRECISION_SPLIT_F16   : every f32 operand is carried as two f16 halves (x = hi + lo) and each product is formed as
 *       lo*wh + hi*wl + hi*wh on the f16 matrix pipe with f32 accumulation -- f32-accurate (error against an f64 reference
 *       not larger than the exact path's), ~2.4x faster.  If an activation does not fit f16 the kernels raise a
 *       flag: am_forward_ff and am_ff_run then repeat the pass on the exact path by themselves (in this mode am_ff_run
 *       synchronises to read the flag), so the resident posteriors are always valid; am_pipeline_overflowed() only
 *       reports that this happened.  The recurrent path always uses AM_PRECISION_F32. */
enum { AM_PRECISION_F32 = 0, AM_PRECISION_SPLIT_F16 = 1 };
int am_engine_set_precision(am_engine* e, int precision);
int am_pipeline_overflowed(am_pipeline* p, int* flag);   /* synchronises; flag = 1 if the last run overflowed f16 (and was repeated exactly) */

/* HIP events on the engine's compute stream (the stream every kernel below is launched on).
 * slot in [0, 64).  Used by bench.py for the timed region. */
int am_event_record(am_engine* e, int slot);
int am_event_elapsed_ms(am_engine* e, int slot_start, int slot_stop, float* ms); /* synchronises on slot_stop */
/* Device time of the launch sequence of this engine's most recent am_decode_host / am_decode_pipeline call (first launch -> last
 * launch done, gaps included; two events of the engine's own, no caller slot is touched).  Synchronises on the second event. */
int am_decode_elapsed_ms(am_engine* e, float* ms);

/* Per-kernel timing: when on, every kernel launch is bracketed by a HIP event pair on the compute stream.
 * am_profile_read synchronises and returns, per kernel id, the number of launches and the summed duration
 * since the last reset. */
#define AM_KERNEL_ID_MAX 16
int am_profile_enable(am_engine* e, int on);
int am_profile_reset(am_engine* e);
int am_profile_read(am_engine* e, int64_t calls[AM_KERNEL_ID_MAX], double ms[AM_KERNEL_ID_MAX]);
const char* am_kernel_name(int kernel_id); /* NULL when out of range */

/* Page-locked host memory visible to every GPU (hipHostMallocPortable): posteriors land in it at PCIe line rate and
 * am_forward_ff overlaps the copies with compute.  Optional: every entry point also accepts ordinary (pageable) arrays. */
int am_host_alloc(size_t bytes, void** out);
int am_host_free(void* ptr);

/* ---- models --------------------------------------------------------------------------------------- */
/* replaces get_nn + L.Classifier + serializers.load_npz (chainer_networks.py:163-184, evaluate.py:134-136,
 * predict_folds.py:153-157,206).  Weights in Chainer layout:
 *   AM_NET_FF  : n_layers hidden Linear + 1 output Linear -> (n_layers+1) entries of W[out,in], b[out];
 *                W_lat must be NULL.
 *   AM_NET_LSTM: n_layers L.LSTM + 1 output Linear.  For l < n_layers: W[l] = upward.W [4U,in],
 *                b[l] = upward.b [4U], W_lat[l] = lateral.W [4U,U], gates interleaved (column 4u+k, k=a,i,f,o);
 *                entry n_layers is the output Linear.  W_lat has n_layers entries.
 * in_dims/out_dims have n_layers+1 entries (out_dims[l] = 4U for LSTM layers).  Layer input sizes must chain. */
int am_model_create(am_engine* e, int kind, int n_layers, int activation,
                    const int* in_dims, const int* out_dims,
                    const float* const* W, const float* const* b, const float* const* W_lat,
                    am_model** out);
/* The other recurrent cells get_nn can build (chainer_networks.py:103-161,174-181): they run on the
 * recurrent path (am_lstm_upload / am_lstm_run / am_forward_lstm) like AM_NET_LSTM models.
 *   AM_CELL_GRU      "gru" = chainer L.GRU (activation AM_ACT_TANH) and "mgrurelur" = the reference's MGRU with reset gate
 *                    (MGRU.py:67-85, activation AM_ACT_RELU):  r = sig(W_r x + U_r h), z = sig(W_z x + U_z h),
 *                    h~ = act(W x + U (r*h)), h' = z*h~ + (1-z)*h; with no previous state: z = sig(W_z x), h' = z*act(W x)
 *                    (the state Linears and their biases are skipped, MGRU.py:70-84).
 *                    12 arrays per layer: W_r.W, W_r.b, U_r.W, U_r.b, W_z.W, W_z.b, U_z.W, U_z.b, W.W, W.b, U.W, U.b
 *   AM_CELL_MGRU     "mgrurelu": the same without reset gate, h~ = act(W x + U h).  8 arrays: W_z.W ... U.b as above.
 *   AM_CELL_PEEPHOLE "peepholelstm" = chainer StatefulPeepholeLSTM: i,f see peep_i(c), peep_f(c); o sees peep_o(c').
 *                    6 arrays: upward.W [4U,in], upward.b [4U], lateral.W [4U,U] (gates interleaved a,i,f,o),
 *                    peep_i.W, peep_f.W, peep_o.W [U,U].
 * params: n_layers * (12 | 8 | 6) pointers, layer-major, Chainer layout W [out,in], b [out]; layer 0 takes in_dim
 * inputs, the others `units`.  out_W [n_classes, units], out_b [n_classes]: the output Linear. */
/* "tdnn" (chainer_networks.py:24-42): n_layers Convolution2D (1, ksize[l]) over the window of
 * win = sum(ksize) - n_layers + 1 positions, then the output Linear on the one remaining position.  A feed-forward model:
 * it runs through am_ff_upload / am_ff_run / am_forward_ff behind a frontend of `win` shifts and no i-vectors.  The
 * reference views each spliced row of win*D values as [D channels][win positions] (F.reshape(x, (B, -1, 1, win))): channel
 * c, position p is element c*win + p of the row -- reproduced here.  W[l]: [units[l], in_channels_l, 1, ksize[l]]
 * (in_channels_0 = in_channels = D, then units[l-1]), b[l]: [units[l]]; out_W: [n_classes, units[n_layers-1]]. */
int am_model_create_tdnn(am_engine* e, int n_layers, int activation, int in_channels, const int* units, const int* ksize,
                         const float* const* W, const float* const* b, const float* out_W, const float* out_b, int n_classes,
                         am_model** out);
enum { AM_CELL_LSTM = 0, AM_CELL_GRU = 1, AM_CELL_MGRU = 2, AM_CELL_PEEPHOLE = 3 };
int am_model_create_rnn(am_engine* e, int cell, int n_layers, int activation, int in_dim, int units, int n_classes,
                        const float* const* params, const float* out_W, const float* out_b, am_model** out);
int am_model_destroy(am_model* m);

/* ---- pipeline ------------------------------------------------------------------------------------- */
int am_pipeline_create(am_engine* e, am_pipeline** out);
int am_pipeline_destroy(am_pipeline* p);

/* Fused splice + Kaldi feature transform (+ i-vector passthrough columns).
 * replaces splicing (kw_utils.py:24-36) / prepareBatch (kw_nn_utils.py:19-43) +
 * applyKaldiFeatureTransform (kw_nn_utils.py:13-17) + np.concatenate((data, ivectors)) (evaluate.py:169-171).
 * shifts: n_shifts frame offsets (e.g. -5..5; {0} = no splice).  add_shift/rescale: [n_shifts*D] or both NULL
 * (no transform).  d_iv: number of i-vector columns appended AFTER the transform (0 = none). */
int am_pipeline_set_frontend(am_pipeline* p, int D, int n_shifts, const int* shifts,
                             const float* add_shift, const float* rescale, int d_iv);

/* Member networks and how they are combined.  All members must be of the same kind and output size.
 * AM_COMBINE_LOGITS follows NNWithRPL (evaluate.py:35-51): has_master=1 -> members[0] is the master:
 *   K=n-1 folds: K==0 -> master(x); else (K*master + sum folds)/(2K);  has_master=0 -> (sum folds)/n.
 * AM_COMBINE_LOGPOST follows predict_folds.py:199-220 (has_master must be 0). */
int am_pipeline_set_members(am_pipeline* p, int n_members, am_model* const* members, int has_master, int combine);

/* RPL4 epilogue (RPL.py:58-74): W, b, lb each [C]; all NULL disables it. */
int am_pipeline_set_rpl4(am_pipeline* p, const float* W, const float* b, const float* lb);

/* Prior subtraction y - ap (evaluateModelForTest.py:75-76,110-111); ap [C] already scaled by --ap-coef
 * (evaluate.py:174); NULL disables it (surface B, predict_folds). */
int am_pipeline_set_prior(am_pipeline* p, const float* ap);

/* Whether the final y - logsum(y) (kw_utils.py:38-43) is applied (default 1). 0 returns the combined logits. */
int am_pipeline_set_normalise(am_pipeline* p, int on);

/* ---- feed-forward path ---------------------------------------------------------------------------- */
/* One contiguous frame shard.  x: [halo_lo + n + halo_hi, D] -- the shard's n frames plus the context rows
 * that exist around it in the file; the splice clamps at the ends of THIS array, so pass halo = max|shift|
 * unless the shard touches the start/end of the file (global clamp, kw_utils.py:29-33).
 * ivec: [n, d_iv] or NULL.  am_ff_upload copies to HBM; am_ff_run launches the whole device path on the
 * compute stream (asynchronous); am_ff_download copies y [n, C] back and synchronises.
 * am_forward_ff = upload, then the device path in 16,384-row chunks (AM_FF_PIPE_ROWS) with the download of chunk c overlapping the
 * compute of chunk c+1 on a separate stream (same results).  Replaces the FF branch of predict (predict_folds.py:69-94) and of
 * evaluateModelTestTri (evaluateModelForTest.py:98-112).
 * Kernel form: networks whose hidden widths pad to multiples of 256 (<= 8 hidden layers) run their whole Linear stack as ONE
 * launch, a work queue over (layer, row tile, column tile) with write-through hand-off of the hidden activations between
 * workgroups (gemm_stack_kernel); everything else, and AM_FF_STACK=0 in the environment, one launch per layer.
 * Same bits either way.  The inter-workgroup waits are bounded: a launch that gives up is repeated on the per-layer launches
 * inside am_ff_download / am_forward_ff / am_decode_pipeline (counted by am_lstm_fallback_count), never a wrong result. */
int am_ff_upload(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi);
int am_ff_run(am_pipeline* p);
int am_ff_download(am_pipeline* p, float* y);
int am_forward_ff(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi, float* y);

/* ---- recurrent path ------------------------------------------------------------------------------- */
/* x: [N, D] frames of n_utts whole utterances, offsets: [n_utts+1] (offsets[0]=0, offsets[n_utts]=N).
 * timedelay >= 0: output frame j of an utterance is the network output at step j + timedelay
 * (predict_folds.py:59-61).  The reference edge-pads each utterance by `timedelay` rows (predict_folds.py:39) but
 * stores a step only for utterances whose un-padded length exceeds t (predict_folds.py:50,61), so the last
 * `timedelay` output frames of every utterance are exactly 0 (or -log C after the dev-mode renormalisation of
 * AM_COMBINE_LOGPOST, predict_folds.py:219) -- reproduced here.  Pass 0 for the evaluate
 * behaviour (evaluateModelForTest.py:67-80).  y: [N, C] in file order.
 * Replaces the time loops of predict (predict_folds.py:28-68) and evaluateModelTestTri (evaluateModelForTest.py:52-94).
 * am_lstm_run is asynchronous.  LSTM stacks of 256 or 512 units whose row groups fit the chip (<= 192 utterances at three
 * 512-unit layers) run as ONE persistent launch with the weights resident in registers (lstm_persist_kernel); every other
 * shape, and the GRU / MGRU / peephole cells, run one launch per time step of the (layer, time) wavefront.  Both forms give
 * the same bits.  The persistent kernel's inter-workgroup waits are bounded (2 s): if one ever times out, every workgroup leaves and
 * am_lstm_download / am_decode_pipeline repeat the pass on the per-step kernels (a line on stderr says so) -- never a hung GPU --
 * and the device stays on the per-step kernels for the rest of the process (am_lstm_persist_enabled / _reenable below).
 * Environment switches (measurement only): AM_LSTM_PERSIST=0 per-step kernels only; AM_LSTM_OVERLAP=0 no second-stream
 * overlap of the output layer with the recurrence's tail; AM_LSTM_GRAPH=0|1 hipGraph replay of the per-step form. */
int am_lstm_upload(am_pipeline* p, const float* x, const float* ivec, const int64_t* offsets, int n_utts, int timedelay);
int am_lstm_run(am_pipeline* p);
int am_lstm_download(am_pipeline* p, float* y);
int am_forward_lstm(am_pipeline* p, const float* x, const float* ivec, const int64_t* offsets, int n_utts,
                    int timedelay, float* y);

/* Diagnostic (AM_LSTM_STAMPS=1 in the environment): the timeline of the last persistent recurrent launch, 8 stamps of the
 * GPU's 100 MHz counter per (row group, time step): step start, after the upward wait / product, after the lateral wait /
 * product, stores issued, published.  out: [n_groups][n_steps][8] or NULL to query the sizes. */
int am_lstm_stamps(am_pipeline* p, unsigned long long* out, int64_t capacity, int* n_groups, int* n_steps);
/* Diagnostic (AM_FF_STAMPS=1): per workgroup of the last single-launch Linear stack (gemm_stack_kernel) 16 words: start and end
 * (100 MHz ticks), tiles computed, ticks spent in dependency waits, number of waits, tiles taken from another XCD's queue, start
 * of its last tile, XCC id; then ticks summed over its tiles: item pull + dependency check, cold start (to the first stage
 * barrier), the product loop, the epilogue; 4 unused.  out: [256][16]. */
int am_ff_stack_stamps(am_pipeline* p, unsigned long long* out);
/* How often, in this process, a persistent launch (the recurrent kernel, or the single-launch Linear stack) gave up on a bounded
 * wait and the pass was repeated on the per-step / per-layer kernels (0 in a healthy run; the tests and the randomised sweep
 * assert it). */
int am_lstm_fallback_count(void);
/* The persistent recurrent kernel needs all its workgroups resident at once.  Inside one process its launches on a device are
 * chained (each waits for the previous one's completion, whichever engine launched it), and its grid is sized from the device's
 * real CU count; what the library cannot see is another process holding CUs.  So a give-up is sticky: after the first one the
 * device stays on the per-step kernels for the rest of the process (one 2-s stall, not one per upload).
 * am_lstm_persist_enabled: 1 while uploads on `device` may still choose the persistent form (0 also with AM_LSTM_PERSIST=0);
 * am_lstm_persist_reenable lifts the latch (tests; an operator who knows the card is free again).
 * am_pipeline_recurrent_form: how many of the uploaded pipeline's members run as persistent launches. */
int am_lstm_persist_enabled(int device);
int am_lstm_persist_reenable(int device);
int am_pipeline_recurrent_form(am_pipeline* p, int* n_persistent, int* n_members);

/* Wait for everything the pipeline has enqueued and make sure its resident posteriors are valid: a persistent launch that gave
 * up on a bounded wait is repeated on the per-step / per-layer kernels here (as am_*_download and am_decode_pipeline do).  Callers
 * that time am_ff_run / am_lstm_run loops without downloading call this once after the loop and compare am_lstm_fallback_count(). */
int am_pipeline_synchronize(am_pipeline* p);
/* number of frames / classes of the resident output (after *_upload) */
int am_pipeline_output_shape(am_pipeline* p, int64_t* n, int* c);

/* ---- stand-alone stage entry points (parity tests + roofline measurements) ------------------------ */
/* splice+transform only: out [n, n_shifts*D + d_iv] (host). */
int am_frontend_apply(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi, float* out);
/* y - logsum(y) in place on host data through the device kernel (a14), with optional prior. lp: [n, c]. */
int am_lognormalise(am_engine* e, float* lp, int64_t n, int c, const float* ap);
/* single Linear (+activation) through the MFMA GEMM kernel: y[n,out] = act(x[n,in] W[out,in]^T + b); act<0 = none. */
int am_linear(am_engine* e, const float* x, int64_t n, int in_dim, const float* W, const float* b, int out_dim,
              int act, float* y);

/* ---- Viterbi phone decoder (consumer of surface A) ------------------------------------------------- */
/* replaces the PhoneRecog subprocess (evaluateModelForTest.py:124-127): CPhoneRecog::Init / RecogOne
 * (recog_src/source/phone_recog.cpp:84-136, 138-254), one workgroup per utterance (main.cpp:51 parallelises over
 * utterances with OpenMP).  The arrays are the fields of the decoding-graph image (kaldiTri1909.img) in file order;
 * lm is scaled by LMW here (phone_recog.cpp:112); PIP is added to every triphone entry (callers pass -abs(PIP)).
 * Segmentations are bit-identical to the reference: same float adds, same strict comparisons. */
typedef struct am_decoder am_decoder;
int am_decoder_create(am_engine* e, int numPhones, int numBiphones, int numTriphones, int numStates,
                      const int* startIdxs, const int* endIdxs, const int* biphoneIdxIn, const int* stateIdxs,
                      const float* tpSelfloop, const float* tpTransition, const int* outOffsets, const int* outValues,
                      const float* lm, float PIP, float LMW, am_decoder** out);
int am_decoder_destroy(am_decoder* d);
/* Which of the three kernel forms am_decode_* runs; all give the same segmentations.  AUTO (default) takes the
 * frame-synchronous form when the graph fits 16-bit node / state ids and entry times, else one workgroup per utterance
 * (main.cpp:51's parallelisation) on compact or, for larger graphs, 32-bit records.  A form that does not fit the graph
 * or batch makes am_decode_* return AM_EINVAL (parity tests walk through all three). */
#define AM_VITERBI_AUTO 0
#define AM_VITERBI_FRAME_SYNC 1
#define AM_VITERBI_PER_UTT_COMPACT 2
#define AM_VITERBI_PER_UTT_WIDE 3
int am_decoder_set_form(am_decoder* d, int form);
/* lp: [offsets[n_utts], numStates] log-probabilities (what the .lab files hold), utterance u = rows
 * [offsets[u], offsets[u+1]).  Output: utterance u's segments in time order at index offsets[u] .. offsets[u] +
 * seg_count[u] - 1 of seg_begin / seg_end (frames, as written to the MLF) / seg_tri (triphone id -> phone table).
 * Utterances that cannot reach a triphone exit (< 3 frames), for which the reference never terminates, get 0 segments. */
int am_decode_host(am_decoder* d, const float* lp, const int64_t* offsets, int n_utts,
                   int* seg_begin, int* seg_end, int* seg_tri, int* seg_count);
/* Same, on the posteriors a pipeline has just produced, without leaving the GPU (no .lab round trip). */
int am_decode_pipeline(am_decoder* d, am_pipeline* p, const int64_t* offsets, int n_utts,
                       int* seg_begin, int* seg_end, int* seg_tri, int* seg_count);

#ifdef __cplusplus
}
#endif
#endif /* AMFWD_H */
