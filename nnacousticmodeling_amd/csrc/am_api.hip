// C-ABI host layer of libamfwd.so (see include/amfwd.h).  Owns device memory, streams and the launch sequence;
// every arithmetic step runs in the gfx950 kernels of am_kernels.hip.  There is no CPU fallback.
#include "../../include/amfwd.h"
#include "am_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <numeric>
#include <string>
#include <chrono>
#include <vector>

using namespace am;

// ----------------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------------
static thread_local std::string g_err = "";

static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(_e == hipErrorOutOfMemory ? AM_ENOMEM : AM_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                \
    } while (0)

#define AM_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r != AM_OK) return _r; \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ----------------------------------------------------------------------------------------------------
// objects
// ----------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return AM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return AM_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct am_engine {
    int device = -1;
    hipStream_t stream = nullptr;    // compute stream: every GEMM / frontend / recurrent kernel
    hipStream_t stream2 = nullptr;   // copy stream: D2H of chunk c overlaps the compute of chunk c+1 (am_forward_ff); second row chain
    hipStream_t chain[2] = {nullptr, nullptr};   // third / fourth row chain of a feed-forward chunk (ff_run_chunks)
    hipEvent_t s2_tail = nullptr;    // last work enqueued on stream2
    hipEvent_t persist_done = nullptr;   // behind this engine's last persistent recurrent launch (PersistGate)
    int n_cus = 0;                   // CUs of the device (hipDeviceProp_t::multiProcessorCount): grid sizes of the resident-workgroup kernels
    bool s2_dirty = false;
    int precision = AM_PRECISION_F32;  // AM_PRECISION_* for the feed-forward Linear layers
    DevBuf d_overflow;                 // int flag raised by split-f16 kernels when an activation does not fit f16
    hipEvent_t ev[64] = {};
    hipEvent_t dec_ev[2] = {nullptr, nullptr};   // around the launch sequence of this engine's last am_decode_* call (am_decode_elapsed_ms)
    bool profile = false;
    struct Rec { int id; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    int64_t prof_calls[AM_KERNEL_ID_MAX] = {};
    double prof_ms[AM_KERNEL_ID_MAX] = {};
    char name[256] = {};
    DevBuf scratch[4];  // stand-alone stage entry points
    // models, pipelines and decoders keep their engine alive: am_engine_destroy before the last of them only marks the
    // engine closed, the last child's destroy then tears it down (host bindings finalise objects in no particular order)
    int children = 0;
    bool closed = false;
};

struct Layer {
    int in = 0, out = 0, in_pad = 0, out_pad = 0;
    int U = 0, U_pad = 0;  // LSTM layers only
    DevBuf W, b, Wlat;
    DevBuf Ws, Wlats;      // LSTM layers of 768 / 1024 units: W (layers >= 1) and Wlat in the order lstm_stream_kernel streams them (rec_stream_pack_weights)
    // split-f16 precision mode (feed-forward only), built on first use
    float wmax = 0.f;      // max |W| (host scan at creation)
    float w_scale = 1.f;   // exact power of two applied to W before splitting
    int out_pad_s = 0;     // output width padded to 256
    DevBuf Whi, Wlo, b_s;
    // other recurrent cells (am_model_create_rnn): packed phase weights and their bias vectors, see RecEpilogue
    DevBuf WA, WB, bA_first, bA_rest, bB_first, bB_rest;
    DevBuf WAs, WBs;       // WA, WB in the order the persistent cell kernel streams them (rec_stream_pack_weights)
    int Kx = 0;            // padded input width of the cell (multiple of 64)
};

struct am_model {
    am_engine* e = nullptr;
    int kind = 0, n_layers = 0, act = 0;
    int cell = AM_CELL_LSTM;   // recurrent models: which cell the layers are
    // TDNN (am_model_create_tdnn): a feed-forward model whose hidden layers are (1,k) convolutions over a window of
    // `win` positions; layer l maps pos[l] positions x chan[l] channels to pos[l+1] x chan[l+1] (chan = padded widths)
    bool tdnn = false;
    int win = 0, tdnn_D = 0;
    std::vector<int> ksize, pos, chan;
    std::vector<Layer> L;  // n_layers + 1 (last = output Linear)
    int in_dim = 0, in_pad = 0, C = 0, C_pad = 0;
    int max_width = 0;     // widest padded hidden activation
    int split_state = 0;   // 0 not built, 1 ready, -1 unsupported shape (falls back to the exact path)
};

struct am_pipeline {
    am_engine* e = nullptr;
    // frontend
    int D = 0, n_shifts = 0, d_iv = 0;
    bool has_ft = false;
    DevBuf d_shifts, d_add, d_scale;
    // members
    std::vector<am_model*> members;
    int has_master = 0, combine = AM_COMBINE_LOGITS;
    // epilogue
    bool has_rpl = false, has_prior = false;
    DevBuf d_rplw, d_rplb, d_rpllb, d_prior;
    int normalise = 1;
    // resident state
    int mode = 0;  // 0 nothing uploaded, 1 ff, 2 lstm
    bool results_valid = false;  // d_out holds the posteriors of the current upload (set by the run calls, cleared by upload / set_*)
    int64_t n = 0;
    int halo_lo = 0, halo_hi = 0;
    bool has_ivec = false;
    DevBuf d_x, d_ivec, d_in, d_in_s, d_act0, d_act1, d_acc, d_acc2, d_out;
    // the Linear stack as one work-queue launch (gemm_stack_kernel): one activation plane per hidden layer + its counters
    DevBuf d_stack, d_ffsync;          // float [planes][stack_rows * stack_width]; unsigned {abort flag, 8 queues, done[hidden][row tiles]}
    // Ensembles: two members in flight (alternating streams), so that one member's stack launch fills the CUs the previous one's ragged end
    // leaves idle: a second set of planes / counters, and one logit buffer per member (combined in member order by finish_kernel).
    DevBuf d_stack2, d_ffsync2, d_accs;
    int64_t accs_members = 0;          // member buffers d_accs holds (0: the members run one after the other into d_acc)
    int64_t stack_rows = 0; int stack_width = 0, stack_planes = 0;
    bool used_stack = false;           // the current results came from gemm_stack_kernel (its abort flag is checked before they are handed out)
    bool stack_off = false;            // a stack launch gave up on a wait: this pipeline stays on the per-layer launches
    int acc_ld = 0;        // leading dimension of d_acc / d_acc2 in the current run (C_pad, or C padded to 256 in split mode)
    bool ran_split = false;
    bool fell_back = false;  // the last split-f16 run overflowed f16 and was repeated on the exact path
    std::vector<hipEvent_t> chunk_done;  // one-shot path: chunk c's posteriors are final (compute stream -> copy stream)
    // lstm
    DevBuf d_src, d_dst, d_G, d_H, d_c;   // d_H / d_c: one packed H plane and one cell-state array per layer
    DevBuf d_cplane;                      // persistent form: the cell state as one time-major packed plane per layer, like d_H
    DevBuf d_recplanes;                   // persistent form of the other cells: two more time-major planes per layer (GRU: Z, r.h; peephole: c, the o pre-activation)
    std::vector<int64_t> step_off;
    std::vector<int> n_active;
    int64_t n_packed = 0;
    int n_slots = 0;
    int timedelay = 0;
    // persistent form of the LSTM stack (lstm_persist_kernel): per member the group plan + arrival table, shared step tables
    struct PersistPlan {
        int form = 0;                       // 0 per-step kernels, 1 lstm_persist_kernel
        std::vector<LstmPersistArgs> chunks;   // one launch per (layer segment, chunk of utterance slots); utterances are independent, longest first
        std::vector<int> layer0;               // the model layer that is L[0] of chunks[i] (a deep stack runs as several launches over layer ranges)
        std::vector<DevBuf> cums;
        RecStreamArgs rsa{};                   // form 3: the cells' persistent launch (rec_stream_kernel) and its role table on the device
        DevBuf d_roles;
        std::vector<RecStreamRole> h_roles;    // (the host copy the asynchronous upload reads from: lives as long as the plan)
        bool split = false;                    // form 3: the upper layers' phases run as two roles each (input part / state part)
        DevBuf d_xplanes;                      // ... and these are the accumulator planes between them
        void release() { for (auto& c : cums) c.release(); cums.clear(); chunks.clear(); layer0.clear(); d_roles.release(); d_xplanes.release(); form = 0; split = false; }
    };
    std::vector<PersistPlan> persist;
    DevBuf d_nact, d_stepoff, d_sync;   // int [T], int [T + 1], unsigned [groups + 1] (arrival counters, abort flag last)
    DevBuf d_tileoff;                   // int [T]: 16-row tiles of the steps before t (rec_stream_kernel's split phases)
    int64_t total_tiles = 0;
    DevBuf d_stamps;                    // AM_LSTM_STAMPS=1: timeline of the last member's persistent launch (am_lstm_stamps)
    int stamps_groups = 0, stamps_T = 0;
    bool used_persist = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // second-stream overlap of the output layer with the recurrence's tail; row chains
    hipEvent_t ev_chain[2] = {nullptr, nullptr};
    // the recurrent launch sequence (thousands of ~2 us kernels) is replayed from a hipGraph from the 2nd run on
    int lstm_runs = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        graph_exec = nullptr; graph = nullptr; lstm_runs = 0;
        results_valid = false;   // every caller is an upload or a configuration change
    }
};

// One persistent recurrent launch at a time per device.  lstm_persist_kernel needs ALL its workgroups resident (they wait for
// each other); two of them launched side by side -- two engines on one card, PosteriorRunner([(0, 0), (0, 1)]) -- can each end up
// partly resident, and then both run into the 2-s bound.  So every such launch first makes its stream wait (on the device) for
// the previous one's completion event, whichever engine of this process launched it: a linear chain, no cycle, and other kernels
// (the output layer beside the tail, another engine's GEMMs) still overlap.  Another PROCESS on the same card cannot be chained;
// for that case a give-up is sticky: after the first one the device stays on the per-step kernels for the rest of the process
// (am_lstm_persist_reenable lifts it), instead of stalling 2 s again on every upload.
struct PersistGate {
    std::mutex mu;
    hipEvent_t last = nullptr;        // completion of the most recent persistent launch on this device ...
    const am_engine* owner = nullptr; // ... and the engine whose event that is
    std::atomic<bool> off{false};     // a launch gave up on this device: no persistent form until re-enabled
};
constexpr int AM_MAX_DEVICES = 64;
static PersistGate g_gate[AM_MAX_DEVICES];
static PersistGate& gate_of(const am_engine* e) { return g_gate[e->device >= 0 && e->device < AM_MAX_DEVICES ? e->device : 0]; }

// Rows per feed-forward chunk of the device-resident path (bounds the activation workspace); tunable for experiments.
static int64_t env_int(const char* name, int64_t dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoll(v) : dflt;
}
static const int64_t FF_CHUNK_ROWS = std::max<int64_t>(GEMM_BM, env_int("AM_FF_CHUNK_ROWS", 1 << 18) / GEMM_BM * GEMM_BM);
// chunk of the one-shot host-to-host call.  That path is bound by the download of the posteriors (7.6 KB per frame over
// PCIe), so the chunk is sized for overlap, not for GEMM efficiency: 16,384 rows measured best (196,608 frames host to host:
// 29.3 ms, against 30.9 at 32,768, 33.8 at 65,536 and 31.8 at 8,192 rows)
static const int64_t FF_PIPE_ROWS = std::max<int64_t>(GEMM_BM, env_int("AM_FF_PIPE_ROWS", 1 << 14) / GEMM_BM * GEMM_BM);

// Whether am_lstm_run replays the recurrent launch sequence from a hipGraph (AM_LSTM_GRAPH=1 / 0 forces it on / off).
// Default: on, except when a queue-intercepting tool (rocprofv3 / rocprof / an HSA tools library) is loaded in the
// process.  ROCm 7.2's intercept path reads past the end of the 16,384-packet AQL ring when a graph's pre-built packet
// batch straddles the ring's wrap point: SIGSEGV inside hipGraphLaunch at a 2^20-aligned address, reproduced to the
// exact pass by tools/repro_graph_wrap.py.  Eager launches write one packet at a time and are
// not affected; a tracing profiler costs tens of microseconds per kernel anyway, so nothing is lost.
static bool lstm_use_graph()
{
    static const int mode = [] {
        const int64_t forced = env_int("AM_LSTM_GRAPH", -1);
        if (forced >= 0) return forced != 0 ? 1 : 0;
        for (const char* v : {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY"}) {
            const char* s = getenv(v);
            if (s && *s) return 0;
        }
        if (dlsym(RTLD_DEFAULT, "rocprofiler_configure")) return 0;   // a rocprofiler-sdk tool was preloaded
        return 1;
    }();
    return mode != 0;
}

// ----------------------------------------------------------------------------------------------------
// kernel launch with optional event bracketing
// ----------------------------------------------------------------------------------------------------
template <class F>
static int run_kernel(am_engine* e, int id, F&& launch, hipStream_t on = nullptr)
{
    hipStream_t st = on ? on : e->stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (e->profile) {
        for (hipEvent_t* ev : {&a, &b}) {
            if (!e->pool.empty()) { *ev = e->pool.back(); e->pool.pop_back(); }
            else HIP_TRY(hipEventCreate(ev));
        }
        HIP_TRY(hipEventRecord(a, st));
    }
    hipError_t r = launch();
    if (r != hipSuccess) return fail(AM_EHIP, "launch of %s failed: %s", am_kernel_name(id), hipGetErrorString(r));
    if (e->profile) {
        HIP_TRY(hipEventRecord(b, st));
        e->recs.push_back({id, a, b});
    }
    return AM_OK;
}

// make the compute stream wait (on the device) for everything enqueued on the copy stream
static int join_streams(am_engine* e)
{
    if (e->s2_dirty) { HIP_TRY(hipStreamWaitEvent(e->stream, e->s2_tail, 0)); e->s2_dirty = false; }
    return AM_OK;
}

static int sync_all(am_engine* e)
{
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream2));
    for (auto c : e->chain) HIP_TRY(hipStreamSynchronize(c));
    e->s2_dirty = false;
    return AM_OK;
}

static int drain_profile(am_engine* e)
{
    if (e->recs.empty()) return AM_OK;
    AM_TRY(sync_all(e));
    for (auto& r : e->recs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        e->prof_calls[r.id] += 1;
        e->prof_ms[r.id] += ms;
        e->pool.push_back(r.a);
        e->pool.push_back(r.b);
    }
    e->recs.clear();
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// discovery / engine
// ----------------------------------------------------------------------------------------------------
extern "C" int am_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" const char* am_last_error(void) { return g_err.c_str(); }
extern "C" const char* am_version(void) { return "amfwd 0.1 (gfx950)"; }

extern "C" const char* am_kernel_name(int id)
{
    static const char* names[K_COUNT] = {"splice_ft", "gemm_hidden", "gemm_out", "finish", "gather_ft", "lstm_upward", "lstm_step", "viterbi", "split_convert", "gemm_split", "gemm_stack"};
    return (id >= 0 && id < K_COUNT) ? names[id] : nullptr;
}

extern "C" int am_engine_create(int device, am_engine** out)
{
    if (!out) return fail(AM_EINVAL, "am_engine_create: out is NULL");
    *out = nullptr;
    int n = am_device_count();
    if (n <= 0) return fail(AM_ENODEV, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(AM_ENODEV, "device %d out of range (0..%d)", device, n - 1);
    if (device >= AM_MAX_DEVICES) return fail(AM_ENODEV, "device %d: at most %d devices are supported", device, AM_MAX_DEVICES);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(AM_ENODEV, "device %d is %s; libamfwd is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    am_engine* e = new am_engine();
    e->device = device;
    e->n_cus = prop.multiProcessorCount;
    snprintf(e->name, sizeof(e->name), "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName, prop.multiProcessorCount);
    // the compute stream outranks the others: where the tail of a recurrent pass and the output layer of its finished rows (second
    // stream) become runnable together, the recurrence's workgroups are placed first and the GEMM fills what they leave
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&e->chain[0], hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&e->chain[1], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->s2_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->persist_done, hipEventDisableTiming) != hipSuccess) { delete e; return fail(AM_EHIP, "hipStreamCreate failed"); }
    hipError_t r = init_kernels();
    if (r != hipSuccess) { (void)hipStreamDestroy(e->stream); delete e; return fail(AM_EHIP, "kernel init failed: %s", hipGetErrorString(r)); }
    *out = e;
    return AM_OK;
}

static void engine_teardown(am_engine* e);

extern "C" int am_engine_destroy(am_engine* e)
{
    if (!e) return AM_OK;
    if (e->closed) return AM_OK;
    e->closed = true;
    if (e->children == 0) engine_teardown(e);
    return AM_OK;
}

static void engine_child_gone(am_engine* e)
{
    if (--e->children == 0 && e->closed) engine_teardown(e);
}

static void engine_teardown(am_engine* e)
{
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamSynchronize(e->stream2);
    (void)hipEventDestroy(e->s2_tail);
    {   // the stream is idle: nobody needs to wait for this engine's last persistent launch any more
        PersistGate& G = gate_of(e);
        std::lock_guard<std::mutex> lk(G.mu);
        if (G.owner == e) { G.last = nullptr; G.owner = nullptr; }
    }
    if (e->persist_done) (void)hipEventDestroy(e->persist_done);
    (void)hipStreamDestroy(e->stream2);
    for (auto c : e->chain) if (c) { (void)hipStreamSynchronize(c); (void)hipStreamDestroy(c); }
    for (auto& r : e->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    for (auto ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : e->dec_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& s : e->scratch) s.release();
    e->d_overflow.release();
    (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int am_engine_synchronize(am_engine* e)
{
    if (!e) return fail(AM_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    return sync_all(e);
}

extern "C" int am_engine_device_name(am_engine* e, char* buf, int buflen)
{
    if (!e || !buf || buflen <= 0) return fail(AM_EINVAL, "bad argument");
    snprintf(buf, buflen, "%s", e->name);
    return AM_OK;
}

extern "C" int am_event_record(am_engine* e, int slot)
{
    if (!e || slot < 0 || slot >= 64) return fail(AM_EINVAL, "bad event slot");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->ev[slot]) HIP_TRY(hipEventCreate(&e->ev[slot]));
    AM_TRY(join_streams(e));  // the event covers the epilogue stream's pending work too
    HIP_TRY(hipEventRecord(e->ev[slot], e->stream));
    return AM_OK;
}

extern "C" int am_event_elapsed_ms(am_engine* e, int a, int b, float* ms)
{
    if (!e || !ms || a < 0 || a >= 64 || b < 0 || b >= 64 || !e->ev[a] || !e->ev[b]) return fail(AM_EINVAL, "bad event slot");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->ev[b]));
    HIP_TRY(hipEventElapsedTime(ms, e->ev[a], e->ev[b]));
    return AM_OK;
}

extern "C" int am_decode_elapsed_ms(am_engine* e, float* ms)
{
    if (!e || !ms) return fail(AM_EINVAL, "am_decode_elapsed_ms: bad argument");
    if (!e->dec_ev[0] || !e->dec_ev[1]) return fail(AM_EINVAL, "am_decode_elapsed_ms: no am_decode_* call has run on this engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventSynchronize(e->dec_ev[1]));
    HIP_TRY(hipEventElapsedTime(ms, e->dec_ev[0], e->dec_ev[1]));
    return AM_OK;
}

extern "C" int am_pack_stream_order(const float* w, int rows, int ld, float* out)
{
    if (!w || !out || rows <= 0 || ld <= 0 || rows % 64 || ld % LSTM_KB) return fail(AM_EINVAL, "am_pack_stream_order: rows must be a multiple of 64 and ld of %d", LSTM_KB);
    rec_stream_pack_weights(w, rows, ld, out);
    return AM_OK;
}

extern "C" int am_device_pci_bus_id(int device, char* buf, int buflen)
{
    if (!buf || buflen < 16) return fail(AM_EINVAL, "am_device_pci_bus_id: the buffer must hold at least 16 bytes");
    const int n = am_device_count();
    if (device < 0 || device >= n) return fail(AM_ENODEV, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipDeviceGetPCIBusId(buf, buflen, device));
    return AM_OK;
}

extern "C" int am_engine_set_precision(am_engine* e, int precision)
{
    if (!e || (precision != AM_PRECISION_F32 && precision != AM_PRECISION_SPLIT_F16)) return fail(AM_EINVAL, "am_engine_set_precision: bad argument");
    e->precision = precision;
    return AM_OK;
}

extern "C" int am_pipeline_overflowed(am_pipeline* p, int* flag)
{
    if (!p || !flag) return fail(AM_EINVAL, "am_pipeline_overflowed: bad argument");
    *flag = p->fell_back ? 1 : 0;
    if (!p->ran_split) return AM_OK;
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(join_streams(e));
    int raised = 0;
    HIP_TRY(hipMemcpyAsync(&raised, e->d_overflow.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (raised) *flag = 1;
    return AM_OK;
}

extern "C" int am_profile_enable(am_engine* e, int on)
{
    if (!e) return fail(AM_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(drain_profile(e));
    e->profile = on != 0;
    return AM_OK;
}

extern "C" int am_profile_reset(am_engine* e)
{
    if (!e) return fail(AM_EINVAL, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(drain_profile(e));
    memset(e->prof_calls, 0, sizeof(e->prof_calls));
    memset(e->prof_ms, 0, sizeof(e->prof_ms));
    return AM_OK;
}

extern "C" int am_profile_read(am_engine* e, int64_t calls[AM_KERNEL_ID_MAX], double ms[AM_KERNEL_ID_MAX])
{
    if (!e || !calls || !ms) return fail(AM_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(drain_profile(e));
    memcpy(calls, e->prof_calls, sizeof(e->prof_calls));
    memcpy(ms, e->prof_ms, sizeof(e->prof_ms));
    return AM_OK;
}

extern "C" int am_host_alloc(size_t bytes, void** out)
{
    if (!out) return fail(AM_EINVAL, "am_host_alloc: out is NULL");
    *out = nullptr;
    if (am_device_count() <= 0) return fail(AM_ENODEV, "no HIP device visible");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocPortable));
    return AM_OK;
}

extern "C" int am_host_free(void* ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// models
// ----------------------------------------------------------------------------------------------------
// All copies and memsets are stream-ordered on the engine's compute stream (a non-blocking stream does not
// synchronise with the null stream), followed by a stream synchronise where the host buffer must be reusable.
static int upload(am_engine* e, DevBuf& d, const std::vector<float>& h)
{
    AM_TRY(d.ensure(h.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    AM_TRY(sync_all(e));
    return AM_OK;
}

// a second copy of a cell's packed weights [rows, ld], in the order the persistent cell kernel streams them
static int upload_stream_order(am_engine* e, DevBuf& d, const std::vector<float>& w, int rows, int ld)
{
    std::vector<float> s(w.size());
    rec_stream_pack_weights(w.data(), rows, ld, s.data());
    return upload(e, d, s);
}

// destination row of (unit u, gate k) in the gate-tiled layout used by lstm_wavefront_kernel
static inline int gate_tiled_row(int u, int k) { return (u / 16) * 64 + k * 16 + (u % 16); }

extern "C" int am_model_create(am_engine* e, int kind, int n_layers, int activation, const int* in_dims, const int* out_dims,
                               const float* const* W, const float* const* b, const float* const* W_lat, am_model** out)
{
    if (!out) return fail(AM_EINVAL, "am_model_create: out is NULL");
    *out = nullptr;
    if (!e || !in_dims || !out_dims || !W || !b) return fail(AM_EINVAL, "am_model_create: NULL argument");
    if (e->closed) return fail(AM_EINVAL, "the engine has been destroyed");
    if (kind != AM_NET_FF && kind != AM_NET_LSTM) return fail(AM_EINVAL, "am_model_create: unknown network kind %d", kind);
    if (n_layers < 0 || n_layers > 64) return fail(AM_EINVAL, "am_model_create: bad layer count %d", n_layers);
    if (kind == AM_NET_FF && (activation < AM_ACT_RELU || activation > AM_ACT_TANH)) return fail(AM_EINVAL, "Wrong activation function specified");
    if (kind == AM_NET_LSTM && (!W_lat || n_layers < 1)) return fail(AM_EINVAL, "am_model_create: LSTM needs lateral weights and >= 1 layer");
    if (kind == AM_NET_LSTM && n_layers > LSTM_MAX_LAYERS) return fail(AM_EINVAL, "am_model_create: at most %d LSTM layers are supported", LSTM_MAX_LAYERS);
    if (kind == AM_NET_LSTM)
        for (int l = 1; l < n_layers; ++l)
            if (out_dims[l] != out_dims[0]) return fail(AM_EINVAL, "am_model_create: all LSTM layers must have the same number of units (chainer_networks.py:44-52)");
    for (int l = 0; l <= n_layers; ++l) {
        if (!W[l] || !b[l] || in_dims[l] <= 0 || out_dims[l] <= 0) return fail(AM_EINVAL, "am_model_create: layer %d has a NULL/empty parameter", l);
        if (kind == AM_NET_LSTM && l < n_layers && (out_dims[l] % 4 != 0 || !W_lat[l])) return fail(AM_EINVAL, "am_model_create: LSTM layer %d malformed", l);
        if (l > 0) {
            const int prev = (kind == AM_NET_LSTM) ? out_dims[l - 1] / 4 : out_dims[l - 1];
            if (in_dims[l] != prev) return fail(AM_EINVAL, "am_model_create: layer %d expects %d inputs but layer %d produces %d", l, in_dims[l], l - 1, prev);
        }
    }
    HIP_TRY(hipSetDevice(e->device));
    am_model* m = new am_model();
    m->e = e; ++e->children; m->kind = kind; m->n_layers = n_layers; m->act = activation;
    m->L.resize(n_layers + 1);
    m->in_dim = in_dims[0];
    m->in_pad = (int)round_up(in_dims[0], kind == AM_NET_LSTM ? 128 : GEMM_BK);   // recurrent members of one ensemble (any cell) share the input plane
    m->C = out_dims[n_layers];
    m->C_pad = gemm_n_pad(m->C);
    int rc = AM_OK;
    for (int l = 0; l <= n_layers && rc == AM_OK; ++l) {
        Layer& Ly = m->L[l];
        Ly.in = in_dims[l]; Ly.out = out_dims[l];
        const bool lstm_layer = (kind == AM_NET_LSTM && l < n_layers);
        if (l == 0) Ly.in_pad = m->in_pad;
        else Ly.in_pad = (kind == AM_NET_LSTM) ? m->L[l - 1].U_pad : m->L[l - 1].out_pad;
        if (lstm_layer) {
            Ly.U = Ly.out / 4;
            Ly.U_pad = (int)round_up(Ly.U, 128);  // multiple of the step kernel's k batch; 4*U_pad is then a GEMM tile multiple
            Ly.out_pad = 4 * Ly.U_pad;
            std::vector<float> w((size_t)Ly.out_pad * Ly.in_pad, 0.f), bb(Ly.out_pad, 0.f), wl((size_t)Ly.out_pad * Ly.U_pad, 0.f);
            for (int u = 0; u < Ly.U; ++u)
                for (int k = 0; k < 4; ++k) {
                    const int src = 4 * u + k, dst = gate_tiled_row(u, k);
                    memcpy(&w[(size_t)dst * Ly.in_pad], W[l] + (size_t)src * Ly.in, sizeof(float) * Ly.in);
                    bb[dst] = b[l][src];
                    memcpy(&wl[(size_t)dst * Ly.U_pad], W_lat[l] + (size_t)src * Ly.U, sizeof(float) * Ly.U);
                }
            if ((rc = upload(e, Ly.W, w)) != AM_OK) break;
            if ((rc = upload(e, Ly.b, bb)) != AM_OK) break;
            if ((rc = upload(e, Ly.Wlat, wl)) != AM_OK) break;
            if (Ly.U_pad == 768 || Ly.U_pad == 1024) {   // the widths lstm_stream_kernel takes
                if (Ly.in_pad == Ly.U_pad && (rc = upload_stream_order(e, Ly.Ws, w, Ly.out_pad, Ly.in_pad)) != AM_OK) break;
                if ((rc = upload_stream_order(e, Ly.Wlats, wl, Ly.out_pad, Ly.U_pad)) != AM_OK) break;
            }
            m->max_width = std::max(m->max_width, Ly.U_pad);
        } else {
            Ly.out_pad = gemm_n_pad(Ly.out);
            Ly.out_pad_s = (int)round_up(Ly.out, 256);
            for (size_t i = 0; i < (size_t)Ly.out * Ly.in; ++i) Ly.wmax = std::max(Ly.wmax, fabsf(W[l][i]));
            std::vector<float> w((size_t)Ly.out_pad * Ly.in_pad, 0.f), bb(Ly.out_pad, 0.f);
            for (int o = 0; o < Ly.out; ++o) memcpy(&w[(size_t)o * Ly.in_pad], W[l] + (size_t)o * Ly.in, sizeof(float) * Ly.in);
            memcpy(bb.data(), b[l], sizeof(float) * Ly.out);
            if ((rc = upload(e, Ly.W, w)) != AM_OK) break;
            if ((rc = upload(e, Ly.b, bb)) != AM_OK) break;
            if (l < n_layers) m->max_width = std::max(m->max_width, Ly.out_pad);
        }
    }
    if (rc != AM_OK) { am_model_destroy(m); return rc; }
    *out = m;
    return AM_OK;
}

extern "C" int am_model_destroy(am_model* m)
{
    if (!m) return AM_OK;
    (void)hipSetDevice(m->e->device);
    for (auto& l : m->L) {
        l.W.release(); l.b.release(); l.Wlat.release(); l.Ws.release(); l.Wlats.release(); l.Whi.release(); l.Wlo.release(); l.b_s.release();
        l.WA.release(); l.WB.release(); l.WAs.release(); l.WBs.release(); l.bA_first.release(); l.bA_rest.release(); l.bB_first.release(); l.bB_rest.release();
    }
    am_engine* e = m->e;
    delete m;
    engine_child_gone(e);
    return AM_OK;
}

// Recurrent cells other than L.LSTM.  Chainer layout in, packed per phase for rec_step_kernel:
// row(unit u, gate g) = 16*(u / NU) + g*NU + u % NU with NU = 16 / gates, columns = [input | state | second state].
extern "C" int am_model_create_tdnn(am_engine* e, int n_layers, int activation, int in_channels, const int* units, const int* ksize,
                                    const float* const* W, const float* const* b, const float* out_W, const float* out_b, int n_classes,
                                    am_model** out)
{
    if (!out) return fail(AM_EINVAL, "am_model_create_tdnn: out is NULL");
    *out = nullptr;
    if (!e || !units || !ksize || !W || !b || !out_W || !out_b) return fail(AM_EINVAL, "am_model_create_tdnn: NULL argument");
    if (e->closed) return fail(AM_EINVAL, "the engine has been destroyed");
    if (n_layers < 1 || n_layers > 64) return fail(AM_EINVAL, "am_model_create_tdnn: bad layer count %d", n_layers);
    if (activation < AM_ACT_RELU || activation > AM_ACT_TANH) return fail(AM_EINVAL, "Wrong activation function specified");
    if (in_channels <= 0 || n_classes <= 0) return fail(AM_EINVAL, "am_model_create_tdnn: bad dimension");
    int win = 1;
    for (int l = 0; l < n_layers; ++l) {
        if (units[l] <= 0 || ksize[l] <= 0 || !W[l] || !b[l]) return fail(AM_EINVAL, "am_model_create_tdnn: layer %d malformed", l);
        win += ksize[l] - 1;   // input_win_size = sum(ksize) - len(ksize) + 1 (chainer_networks.py:32): the last layer sees one position
    }
    HIP_TRY(hipSetDevice(e->device));
    am_model* m = new am_model();
    m->e = e; ++e->children; m->kind = AM_NET_FF; m->tdnn = true; m->n_layers = n_layers; m->act = activation;
    m->win = win; m->tdnn_D = in_channels;
    m->in_dim = in_channels * win;
    m->in_pad = (int)round_up(in_channels, 32);   // channels per position of the channel-last input
    m->C = n_classes; m->C_pad = gemm_n_pad(n_classes);
    m->L.resize(n_layers + 1);
    m->ksize.assign(ksize, ksize + n_layers);
    m->pos.assign(1, win); m->chan.assign(1, m->in_pad);
    int rc = AM_OK, cin = in_channels;
    for (int l = 0; l < n_layers && rc == AM_OK; ++l) {
        Layer& Ly = m->L[l];
        const int k = ksize[l], cp = m->chan[l];
        Ly.in = cin * k; Ly.out = units[l]; Ly.in_pad = cp * k; Ly.out_pad = gemm_n_pad(units[l]);
        Ly.out_pad_s = (int)round_up(units[l], 256);
        for (size_t i = 0; i < (size_t)units[l] * cin * k; ++i) Ly.wmax = std::max(Ly.wmax, fabsf(W[l][i]));
        // Convolution2D weight [out, in, 1, k] (cross-correlation) -> GEMM weight over the window's taps: column j*cp + c
        std::vector<float> w((size_t)Ly.out_pad * Ly.in_pad, 0.f), bb(Ly.out_pad, 0.f);
        for (int o = 0; o < units[l]; ++o)
            for (int c = 0; c < cin; ++c)
                for (int j = 0; j < k; ++j) w[(size_t)o * Ly.in_pad + (size_t)j * cp + c] = W[l][((size_t)o * cin + c) * k + j];
        memcpy(bb.data(), b[l], sizeof(float) * units[l]);
        if ((rc = upload(e, Ly.W, w)) != AM_OK || (rc = upload(e, Ly.b, bb)) != AM_OK) break;
        m->pos.push_back(m->pos[l] - k + 1);
        m->chan.push_back(Ly.out_pad);
        m->max_width = std::max(m->max_width, Ly.out_pad);
        cin = units[l];
    }
    if (rc == AM_OK) {
        Layer& Lo = m->L[n_layers];   // out Linear on the flattened (units, 1, 1) map
        Lo.in = cin; Lo.out = n_classes; Lo.in_pad = m->chan[n_layers]; Lo.out_pad = gemm_n_pad(n_classes);
        Lo.out_pad_s = (int)round_up(n_classes, 256);
        for (size_t i = 0; i < (size_t)n_classes * cin; ++i) Lo.wmax = std::max(Lo.wmax, fabsf(out_W[i]));
        std::vector<float> w((size_t)Lo.out_pad * Lo.in_pad, 0.f), bb(Lo.out_pad, 0.f);
        for (int o = 0; o < n_classes; ++o) memcpy(&w[(size_t)o * Lo.in_pad], out_W + (size_t)o * cin, sizeof(float) * cin);
        memcpy(bb.data(), out_b, sizeof(float) * n_classes);
        if ((rc = upload(e, Lo.W, w)) == AM_OK) rc = upload(e, Lo.b, bb);
    }
    if (rc != AM_OK) { am_model_destroy(m); return rc; }
    *out = m;
    return AM_OK;
}

static int cell_param_count(int cell)
{
    return cell == AM_CELL_GRU ? 12 : cell == AM_CELL_MGRU ? 8 : cell == AM_CELL_PEEPHOLE ? 6 : 0;
}

extern "C" int am_model_create_rnn(am_engine* e, int cell, int n_layers, int activation, int in_dim, int units, int n_classes,
                                   const float* const* params, const float* out_W, const float* out_b, am_model** out)
{
    if (!out) return fail(AM_EINVAL, "am_model_create_rnn: out is NULL");
    *out = nullptr;
    const int np = cell_param_count(cell);
    if (!e || !params || !out_W || !out_b) return fail(AM_EINVAL, "am_model_create_rnn: NULL argument");
    if (e->closed) return fail(AM_EINVAL, "the engine has been destroyed");
    if (np == 0) return fail(AM_EINVAL, "am_model_create_rnn: unknown cell %d", cell);
    if (n_layers < 1 || n_layers > LSTM_MAX_LAYERS) return fail(AM_EINVAL, "am_model_create_rnn: 1..%d layers are supported", LSTM_MAX_LAYERS);
    if (in_dim <= 0 || units <= 0 || n_classes <= 0) return fail(AM_EINVAL, "am_model_create_rnn: bad dimension");
    if (activation < AM_ACT_RELU || activation > AM_ACT_TANH) return fail(AM_EINVAL, "Wrong activation function specified");
    for (int i = 0; i < n_layers * np; ++i)
        if (!params[i]) return fail(AM_EINVAL, "am_model_create_rnn: parameter %d of layer %d is NULL", i % np, i / np);
    HIP_TRY(hipSetDevice(e->device));
    am_model* m = new am_model();
    m->e = e; ++e->children; m->kind = AM_NET_LSTM; m->cell = cell; m->n_layers = n_layers; m->act = activation;
    m->L.resize(n_layers + 1);
    m->in_dim = in_dim;
    m->in_pad = (int)round_up(in_dim, 128);   // K segments in units of the step kernels' 128-k batches
    m->C = n_classes;
    m->C_pad = gemm_n_pad(n_classes);
    const int U = units, U_pad = (int)round_up(U, 128);
    int rc = AM_OK;
    for (int l = 0; l < n_layers && rc == AM_OK; ++l) {
        Layer& Ly = m->L[l];
        const float* const* P = params + (size_t)l * np;
        Ly.in = l == 0 ? in_dim : U; Ly.out = U; Ly.U = U; Ly.U_pad = U_pad; Ly.out_pad = U_pad;
        Ly.Kx = l == 0 ? m->in_pad : U_pad; Ly.in_pad = Ly.Kx;
        const int in = Ly.in, Kx = Ly.Kx;
        auto row_of = [](int u, int g, int G) { const int NU = 16 / G; return 16 * (u / NU) + g * NU + u % NU; };
        if (cell == AM_CELL_GRU || cell == AM_CELL_MGRU) {
            // Linear pairs (input, state) per gate; GRU: r, z, candidate; MGRU without reset gate: z, candidate
            const int o = cell == AM_CELL_GRU ? 4 : 0;   // offset of W_z in the parameter list
            const int ld = Kx + U_pad;
            auto fill = [&](std::vector<float>& w, std::vector<float>& bf, std::vector<float>& br, int row, int u, const float* Wi, const float* bi, const float* Ws, const float* bs) {
                memcpy(&w[(size_t)row * ld], Wi + (size_t)u * in, sizeof(float) * in);
                memcpy(&w[(size_t)row * ld + Kx], Ws + (size_t)u * U, sizeof(float) * U);
                bf[row] = bi[u];             // no previous state: the state Linear (and its bias) is not applied (MGRU.py:70-76)
                br[row] = bi[u] + bs[u];
            };
            if (cell == AM_CELL_GRU) {
                std::vector<float> wa((size_t)2 * U_pad * ld, 0.f), af(2 * U_pad, 0.f), ar(2 * U_pad, 0.f);
                std::vector<float> wb((size_t)U_pad * ld, 0.f), bf(U_pad, 0.f), br(U_pad, 0.f);
                for (int u = 0; u < U; ++u) {
                    fill(wa, af, ar, row_of(u, 0, 2), u, P[0], P[1], P[2], P[3]);       // r
                    fill(wa, af, ar, row_of(u, 1, 2), u, P[4], P[5], P[6], P[7]);       // z
                    fill(wb, bf, br, u, u, P[8], P[9], P[10], P[11]);                   // candidate: [W | U]
                }
                if ((rc = upload(e, Ly.WA, wa)) != AM_OK || (rc = upload(e, Ly.bA_first, af)) != AM_OK || (rc = upload(e, Ly.bA_rest, ar)) != AM_OK) break;
                if ((rc = upload(e, Ly.WB, wb)) != AM_OK || (rc = upload(e, Ly.bB_first, bf)) != AM_OK || (rc = upload(e, Ly.bB_rest, br)) != AM_OK) break;
                if ((rc = upload_stream_order(e, Ly.WAs, wa, 2 * U_pad, ld)) != AM_OK || (rc = upload_stream_order(e, Ly.WBs, wb, U_pad, ld)) != AM_OK) break;
            } else {
                std::vector<float> wa((size_t)2 * U_pad * ld, 0.f), af(2 * U_pad, 0.f), ar(2 * U_pad, 0.f);
                for (int u = 0; u < U; ++u) {
                    fill(wa, af, ar, row_of(u, 0, 2), u, P[o + 0], P[o + 1], P[o + 2], P[o + 3]);   // z
                    fill(wa, af, ar, row_of(u, 1, 2), u, P[o + 4], P[o + 5], P[o + 6], P[o + 7]);   // candidate
                }
                if ((rc = upload(e, Ly.WA, wa)) != AM_OK || (rc = upload(e, Ly.bA_first, af)) != AM_OK || (rc = upload(e, Ly.bA_rest, ar)) != AM_OK) break;
                if ((rc = upload_stream_order(e, Ly.WAs, wa, 2 * U_pad, ld)) != AM_OK) break;
            }
        } else {   // AM_CELL_PEEPHOLE: upward.W [4U,in], upward.b [4U], lateral.W [4U,U] (gates interleaved 4u+g), peep_i/f/o.W [U,U]
            const int ld = Kx + 2 * U_pad;
            std::vector<float> wa((size_t)4 * U_pad * ld, 0.f), ab(4 * U_pad, 0.f), wb((size_t)U_pad * U_pad, 0.f), zb(U_pad, 0.f);
            for (int u = 0; u < U; ++u) {
                for (int g = 0; g < 4; ++g) {
                    const int row = row_of(u, g, 4), src = 4 * u + g;
                    memcpy(&wa[(size_t)row * ld], P[0] + (size_t)src * in, sizeof(float) * in);
                    memcpy(&wa[(size_t)row * ld + Kx], P[2] + (size_t)src * U, sizeof(float) * U);
                    if (g == 1) memcpy(&wa[(size_t)row * ld + Kx + U_pad], P[3] + (size_t)u * U, sizeof(float) * U);   // peep_i(c)
                    if (g == 2) memcpy(&wa[(size_t)row * ld + Kx + U_pad], P[4] + (size_t)u * U, sizeof(float) * U);   // peep_f(c)
                    ab[row] = P[1][src];
                }
                memcpy(&wb[(size_t)u * U_pad], P[5] + (size_t)u * U, sizeof(float) * U);                               // peep_o(c')
            }
            if ((rc = upload(e, Ly.WA, wa)) != AM_OK || (rc = upload(e, Ly.bA_first, ab)) != AM_OK) break;
            if ((rc = upload(e, Ly.WB, wb)) != AM_OK || (rc = upload(e, Ly.bB_first, zb)) != AM_OK) break;
            if ((rc = upload_stream_order(e, Ly.WAs, wa, 4 * U_pad, ld)) != AM_OK || (rc = upload_stream_order(e, Ly.WBs, wb, U_pad, U_pad)) != AM_OK) break;
        }
        m->max_width = std::max(m->max_width, U_pad);
    }
    if (rc == AM_OK) {
        Layer& Lo = m->L[n_layers];
        Lo.in = U; Lo.out = n_classes; Lo.in_pad = U_pad; Lo.out_pad = gemm_n_pad(n_classes); Lo.out_pad_s = (int)round_up(n_classes, 256);
        std::vector<float> w((size_t)Lo.out_pad * Lo.in_pad, 0.f), bb(Lo.out_pad, 0.f);
        for (int o = 0; o < n_classes; ++o) memcpy(&w[(size_t)o * Lo.in_pad], out_W + (size_t)o * U, sizeof(float) * U);
        memcpy(bb.data(), out_b, sizeof(float) * n_classes);
        if ((rc = upload(e, Lo.W, w)) == AM_OK) rc = upload(e, Lo.b, bb);
    }
    if (rc != AM_OK) { am_model_destroy(m); return rc; }
    *out = m;
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// pipeline configuration
// ----------------------------------------------------------------------------------------------------
extern "C" int am_pipeline_create(am_engine* e, am_pipeline** out)
{
    if (!e || !out) return fail(AM_EINVAL, "am_pipeline_create: NULL argument");
    if (e->closed) return fail(AM_EINVAL, "the engine has been destroyed");
    am_pipeline* p = new am_pipeline();
    p->e = e; ++e->children;
    *out = p;
    return AM_OK;
}

extern "C" int am_pipeline_destroy(am_pipeline* p)
{
    if (!p) return AM_OK;
    (void)hipSetDevice(p->e->device);
    (void)hipStreamSynchronize(p->e->stream);
    (void)hipStreamSynchronize(p->e->stream2);
    for (auto ev : p->chunk_done) (void)hipEventDestroy(ev);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    for (auto ev : p->ev_chain) if (ev) (void)hipEventDestroy(ev);
    p->drop_graph();
    for (DevBuf* d : {&p->d_shifts, &p->d_add, &p->d_scale, &p->d_rplw, &p->d_rplb, &p->d_rpllb, &p->d_prior, &p->d_x, &p->d_ivec, &p->d_in, &p->d_in_s,
                      &p->d_act0, &p->d_act1, &p->d_acc, &p->d_acc2, &p->d_out, &p->d_stack, &p->d_ffsync, &p->d_stack2, &p->d_ffsync2, &p->d_accs, &p->d_src, &p->d_dst, &p->d_G, &p->d_H, &p->d_c, &p->d_cplane, &p->d_recplanes, &p->d_nact, &p->d_stepoff, &p->d_tileoff, &p->d_sync, &p->d_stamps})
        d->release();
    for (auto& pl : p->persist) pl.release();
    am_engine* e = p->e;
    delete p;
    engine_child_gone(e);
    return AM_OK;
}

extern "C" int am_pipeline_set_frontend(am_pipeline* p, int D, int n_shifts, const int* shifts, const float* add_shift, const float* rescale, int d_iv)
{
    if (!p || D <= 0 || n_shifts <= 0 || !shifts || d_iv < 0) return fail(AM_EINVAL, "am_pipeline_set_frontend: bad argument");
    if ((add_shift == nullptr) != (rescale == nullptr)) return fail(AM_EINVAL, "am_pipeline_set_frontend: add_shift and rescale must both be given or both NULL");
    HIP_TRY(hipSetDevice(p->e->device));
    AM_TRY(sync_all(p->e));
    p->D = D; p->n_shifts = n_shifts; p->d_iv = d_iv; p->has_ft = add_shift != nullptr;
    AM_TRY(p->d_shifts.ensure(sizeof(int) * n_shifts));
    HIP_TRY(hipMemcpyAsync(p->d_shifts.p, shifts, sizeof(int) * n_shifts, hipMemcpyHostToDevice, p->e->stream));
    if (p->has_ft) {
        const size_t bytes = sizeof(float) * (size_t)n_shifts * D;
        AM_TRY(p->d_add.ensure(bytes));
        AM_TRY(p->d_scale.ensure(bytes));
        HIP_TRY(hipMemcpyAsync(p->d_add.p, add_shift, bytes, hipMemcpyHostToDevice, p->e->stream));
        HIP_TRY(hipMemcpyAsync(p->d_scale.p, rescale, bytes, hipMemcpyHostToDevice, p->e->stream));
    }
    AM_TRY(sync_all(p->e));
    p->mode = 0; p->drop_graph();
    return AM_OK;
}

extern "C" int am_pipeline_set_members(am_pipeline* p, int n_members, am_model* const* members, int has_master, int combine)
{
    if (!p || n_members <= 0 || !members) return fail(AM_EINVAL, "am_pipeline_set_members: bad argument");
    if (combine != AM_COMBINE_LOGITS && combine != AM_COMBINE_LOGPOST) return fail(AM_EINVAL, "am_pipeline_set_members: unknown combine mode");
    if (combine == AM_COMBINE_LOGPOST && has_master) return fail(AM_EINVAL, "am_pipeline_set_members: log-posterior averaging has no master");
    for (int i = 0; i < n_members; ++i) {
        if (!members[i] || members[i]->e != p->e) return fail(AM_EINVAL, "am_pipeline_set_members: member %d is NULL or lives on another engine", i);
        if (members[i]->kind != members[0]->kind || members[i]->C != members[0]->C || members[i]->in_dim != members[0]->in_dim ||
            members[i]->in_pad != members[0]->in_pad || members[i]->tdnn != members[0]->tdnn || members[i]->win != members[0]->win)
            return fail(AM_EINVAL, "am_pipeline_set_members: member %d differs in kind / input / output size", i);
    }
    if (members[0]->C > 2048) return fail(AM_EINVAL, "am_pipeline_set_members: at most 2048 classes are supported");
    p->members.assign(members, members + n_members);
    p->has_master = has_master ? 1 : 0;
    p->combine = combine;
    p->mode = 0; p->drop_graph();
    return AM_OK;
}

static int set_vec(am_pipeline* p, DevBuf& d, const float* src, int c)
{
    AM_TRY(d.ensure(sizeof(float) * c));
    HIP_TRY(hipMemcpyAsync(d.p, src, sizeof(float) * c, hipMemcpyHostToDevice, p->e->stream));
    AM_TRY(sync_all(p->e));
    return AM_OK;
}

extern "C" int am_pipeline_set_rpl4(am_pipeline* p, const float* W, const float* b, const float* lb)
{
    if (!p) return fail(AM_EINVAL, "pipeline is NULL");
    if (!W && !b && !lb) { p->has_rpl = false; p->drop_graph(); return AM_OK; }
    if (!W || !b || !lb) return fail(AM_EINVAL, "am_pipeline_set_rpl4: W, b, lb must all be given");
    if (p->members.empty()) return fail(AM_EINVAL, "am_pipeline_set_rpl4: set the members first");
    HIP_TRY(hipSetDevice(p->e->device));
    AM_TRY(sync_all(p->e));
    const int c = p->members[0]->C;
    AM_TRY(set_vec(p, p->d_rplw, W, c));
    AM_TRY(set_vec(p, p->d_rplb, b, c));
    AM_TRY(set_vec(p, p->d_rpllb, lb, c));
    p->has_rpl = true; p->drop_graph();
    return AM_OK;
}

extern "C" int am_pipeline_set_prior(am_pipeline* p, const float* ap)
{
    if (!p) return fail(AM_EINVAL, "pipeline is NULL");
    if (!ap) { p->has_prior = false; p->drop_graph(); return AM_OK; }
    if (p->members.empty()) return fail(AM_EINVAL, "am_pipeline_set_prior: set the members first");
    HIP_TRY(hipSetDevice(p->e->device));
    AM_TRY(sync_all(p->e));
    AM_TRY(set_vec(p, p->d_prior, ap, p->members[0]->C));
    p->has_prior = true; p->drop_graph();
    return AM_OK;
}

extern "C" int am_pipeline_set_normalise(am_pipeline* p, int on)
{
    if (!p) return fail(AM_EINVAL, "pipeline is NULL");
    p->normalise = on ? 1 : 0; p->drop_graph();
    return AM_OK;
}

static int check_ready(am_pipeline* p, int kind)
{
    if (!p) return fail(AM_EINVAL, "pipeline is NULL");
    if (p->D <= 0) return fail(AM_EINVAL, "pipeline has no frontend (am_pipeline_set_frontend)");
    if (p->members.empty()) return fail(AM_EINVAL, "pipeline has no member network (am_pipeline_set_members)");
    if (p->members[0]->kind != kind) return fail(AM_EINVAL, "pipeline members are %s networks", p->members[0]->kind == AM_NET_FF ? "feed-forward" : "recurrent");
    const int width = p->n_shifts * p->D + p->d_iv;
    if (width != p->members[0]->in_dim)
        return fail(AM_EINVAL, "frontend produces %d features per frame but the network expects %d", width, p->members[0]->in_dim);
    return AM_OK;
}

// combine / finish shared by both paths ----------------------------------------------------------------
static int epilogue_mode(const am_pipeline* p, int mi, float* alpha)
{
    *alpha = 1.f;
    if (p->combine == AM_COMBINE_LOGPOST) return EPI_LINEAR;
    const int n = (int)p->members.size();
    if (mi > 0) return EPI_ACCUM;
    if (p->has_master && n > 1) { *alpha = (float)(n - 1); return EPI_SCALE; }
    return EPI_LINEAR;
}

static float final_divisor(const am_pipeline* p)
{
    const int n = (int)p->members.size();
    if (p->combine == AM_COMBINE_LOGPOST) return (float)n;
    if (p->has_master) return n > 1 ? 2.f * (n - 1) : 0.f;
    return n > 1 ? (float)n : 0.f;
}

static int member_finish(am_pipeline* p, int mi, int64_t rows, int64_t row0 = 0, hipStream_t on = nullptr)
{
    // log-posterior averaging: normalise this member and add it to the running sum (predict_folds.py:207-212); rows [row0, row0 + rows)
    am_model* m0 = p->members[0];
    FinishArgs f{};
    f.in = p->d_acc.as<float>() + row0 * p->acc_ld; f.ldin = p->acc_ld;
    f.out = p->d_acc2.as<float>() + row0 * p->acc_ld; f.ldout = p->acc_ld;
    f.n = rows; f.C = m0->C; f.normalise = 1; f.accumulate = mi > 0;
    hipStream_t st = on ? on : p->e->stream;
    return run_kernel(p->e, K_FINISH, [&] { return launch_finish(st, f); }, st);
}

static int final_finish(am_pipeline* p, int64_t rows, float* out, const int* row_map, int64_t row0 = 0, hipStream_t on = nullptr)
{
    // rows [row0, row0 + rows) of the accumulated logits; with a row_map the output row is row_map[row], else out + row (caller offsets `out`)
    am_model* m0 = p->members[0];
    FinishArgs f{};
    f.in = (p->combine == AM_COMBINE_LOGPOST ? p->d_acc2 : p->d_acc).as<float>() + row0 * p->acc_ld; f.ldin = p->acc_ld;
    f.out = out; f.ldout = m0->C; f.row_map = row_map ? row_map + row0 : nullptr;
    f.n = rows; f.C = m0->C;
    f.divisor = final_divisor(p);
    if (p->has_rpl) { f.rpl_w = p->d_rplw.as<float>(); f.rpl_b = p->d_rplb.as<float>(); f.rpl_lb = p->d_rpllb.as<float>(); }
    if (p->has_prior) f.prior = p->d_prior.as<float>();
    f.normalise = p->normalise;
    hipStream_t st = on ? on : p->e->stream;
    return run_kernel(p->e, K_FINISH, [&] { return launch_finish(st, f); }, st);
}

// ----------------------------------------------------------------------------------------------------
// feed-forward path
// ----------------------------------------------------------------------------------------------------
// TDNN geometry.  Layer l's input is a channel-last matrix of (frames * pos[l]) rows x chan[l]; its GEMM computes
// round_up(frames * pos[l+1], 256) output rows, row R reading input rows [R + (R / pos[l+1]) * (k-1), ... + k).
static int64_t tdnn_chunk_frames(const am_model* m)
{
    const int64_t f = FF_CHUNK_ROWS / std::max(m->win, 1) / GEMM_BM * GEMM_BM;
    return std::max<int64_t>(GEMM_BM, f);
}

static size_t tdnn_rows_needed(const am_model* m, int l, int64_t frames)
{
    // rows of layer l's INPUT buffer (l = 0: the frontend's output; l = n_layers: the out Linear's input)
    size_t rows = (size_t)round_up(frames * m->pos[l], GEMM_BM);
    if (l < m->n_layers) {
        const int64_t R = round_up(frames * m->pos[l + 1], GEMM_BM) - 1;
        rows = std::max(rows, (size_t)(R + R / m->pos[l + 1] * (m->ksize[l] - 1) + m->ksize[l] + 1));
    }
    return rows + 1;
}

static int tdnn_member_forward(am_pipeline* p, int mi, int64_t frames, float* acc)
{
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    const float* A = p->d_in.as<float>();
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        GemmArgs g{};
        g.A = A; g.lda = m->chan[l]; g.W = Ly.W.as<float>(); g.ldw = Ly.in_pad; g.bias = Ly.b.as<float>();
        g.N_pad = Ly.out_pad; g.K_pad = Ly.in_pad; g.act = m->act;
        int id;
        if (l < m->n_layers) {
            g.M_pad = (int)round_up(frames * m->pos[l + 1], GEMM_BM);
            g.row_p = m->pos[l + 1]; g.row_skip = m->ksize[l] - 1;
            g.C = ((l & 1) ? p->d_act1 : p->d_act0).as<float>(); g.ldc = Ly.out_pad; g.epi = EPI_ACT; g.alpha = 1.f; g.n_real = Ly.out;
            id = K_GEMM_HIDDEN;
        } else {
            g.M_pad = (int)round_up(frames, GEMM_BM);   // pos[n_layers] == 1: one row per frame
            g.C = acc; g.ldc = m->C_pad; g.epi = epilogue_mode(p, mi, &g.alpha);
            id = K_GEMM_OUT;
        }
        AM_TRY(run_kernel(e, id, [&] { return launch_gemm(e->stream, g); }));
        A = g.C;
    }
    return AM_OK;
}

// the same in the split-f16 precision mode: every activation plane is a hi half plane followed by a lo half plane of the
// same (rows incl. window slack) x channels shape, i.e. the bytes of the f32 plane it replaces
static int tdnn_member_forward_split(am_pipeline* p, int mi, int64_t frames, float* acc)
{
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    int* flag = e->d_overflow.as<int>();
    const int64_t cap = std::min<int64_t>(round_up(std::max<int64_t>(p->n, 1), GEMM_BM), tdnn_chunk_frames(m));   // frames the buffers were sized for
    // frontend output -> halves (the f32 rows the splice kernel wrote; slack rows stay zero)
    const size_t in_plane = tdnn_rows_needed(m, 0, cap) * (size_t)m->chan[0];
    _Float16* Ahi = (_Float16*)p->d_in_s.p;
    _Float16* Alo = Ahi + in_plane;
    AM_TRY(run_kernel(e, K_SPLIT_CONVERT, [&] { return launch_split_f32(e->stream, p->d_in.as<float>(), Ahi, Alo, (size_t)frames * m->pos[0] * m->chan[0], 1.0f, flag); }));
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        SplitGemmArgs g{};
        g.Ahi = Ahi; g.Alo = Alo; g.lda = m->chan[l]; g.Whi = Ly.Whi.p; g.Wlo = Ly.Wlo.p; g.ldw = Ly.in_pad; g.bias = Ly.b_s.as<float>();
        g.N_pad = Ly.out_pad_s; g.K_pad = Ly.in_pad; g.act = m->act; g.out_scale = 1.0f / Ly.w_scale; g.overflow = flag;
        if (l < m->n_layers) {
            g.M_pad = (int)round_up(frames * m->pos[l + 1], GEMM_BM);
            g.row_p = m->pos[l + 1]; g.row_skip = m->ksize[l] - 1;
            _Float16* Chi = (_Float16*)((l & 1) ? p->d_act1 : p->d_act0).p;
            g.C = nullptr; g.Chi = Chi; g.Clo = Chi + tdnn_rows_needed(m, l + 1, cap) * (size_t)m->chan[l + 1]; g.ldc = Ly.out_pad_s; g.epi = EPI_ACT; g.alpha = 1.f;
            AM_TRY(run_kernel(e, K_GEMM_SPLIT, [&] { return launch_gemm_split(e->stream, g); }));
            Ahi = (_Float16*)g.Chi; Alo = (_Float16*)g.Clo;
        } else {
            g.M_pad = (int)round_up(frames, GEMM_BM);
            g.C = acc; g.ldc = p->acc_ld; g.epi = epilogue_mode(p, mi, &g.alpha);
            AM_TRY(run_kernel(e, K_GEMM_SPLIT, [&] { return launch_gemm_split(e->stream, g); }));
        }
    }
    return AM_OK;
}

static std::atomic<int> g_lstm_fallbacks{0};   // passes a persistent kernel (recurrent, or the Linear stack) gave up on and the per-launch kernels repeated (am_lstm_fallback_count)

// The whole Linear stack of a member as ONE work-queue launch (gemm_stack_kernel) instead of one launch per layer (AM_FF_STACK=0:
// the per-layer launches).  Measured faster at every size: 1,024 rows 1.05 against 1.24 ms, 57,919 rows 5.10 against 5.30 (two row
// chains), 196,608 rows 16.30 against 16.63 ms (4x1024).
constexpr int FF_SYNC_QUEUES = 32, FF_SYNC_DONE = FF_SYNC_QUEUES + 8 * GEMM_STACK_QSTRIDE;   // words of d_ffsync: [0] abort flag, the queue counters, the arrival counters
static int ff_stack_mode() { return (int)env_int("AM_FF_STACK", 1); }   // read on every call: the tests switch it
static bool ff_stack_shape_ok(const am_model* m)
{
    if (m->tdnn || m->n_layers < 1 || m->n_layers > GEMM_STACK_MAX_HIDDEN) return false;
    // (hidden widths that need padding stay on the per-layer launches: their epilogue forces the pad columns to 0, which this kernel has no registers for --
    //  an infinite input value would otherwise become inf x 0 = NaN there and spread; every width of the reference's recipes is a multiple of 256)
    for (int l = 0; l < m->n_layers; ++l) if (m->L[l].out_pad % 256 || m->L[l].in_pad % 32 || m->L[l].out != m->L[l].out_pad) return false;
    const Layer& out = m->L[m->n_layers];
    return out.out_pad % 256 == 0 || out.out_pad % 192 == 0;
}

extern "C" int am_ff_upload(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi)
{
    AM_TRY(check_ready(p, AM_NET_FF));
    if ((!x && halo_lo + n + halo_hi > 0) || n < 0 || halo_lo < 0 || halo_hi < 0) return fail(AM_EINVAL, "am_ff_upload: bad argument");
    if ((p->d_iv > 0) != (ivec != nullptr)) return fail(AM_EINVAL, "am_ff_upload: i-vectors %s", p->d_iv > 0 ? "expected but NULL" : "given but the frontend has d_iv = 0");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    am_model* m0 = p->members[0];
    const int64_t ext = halo_lo + n + halo_hi;
    if (n > 0 && ext <= 0) return fail(AM_EINVAL, "am_ff_upload: empty input");
    const int64_t rows = std::min<int64_t>(round_up(std::max<int64_t>(n, 1), GEMM_BM), std::max(FF_CHUNK_ROWS, FF_PIPE_ROWS));
    int width = 0;
    for (auto* m : p->members) width = std::max(width, m->max_width);
    AM_TRY(p->d_x.ensure(std::max<size_t>(sizeof(float) * ext * p->D, 16)));
    if (ext > 0) HIP_TRY(hipMemcpyAsync(p->d_x.p, x, sizeof(float) * ext * p->D, hipMemcpyHostToDevice, e->stream));
    if (ivec) {
        AM_TRY(p->d_ivec.ensure(std::max<size_t>(sizeof(float) * n * p->d_iv, 16)));
        if (n > 0) HIP_TRY(hipMemcpyAsync(p->d_ivec.p, ivec, sizeof(float) * n * p->d_iv, hipMemcpyHostToDevice, e->stream));
    }
    if (m0->tdnn && (p->D != m0->tdnn_D || p->n_shifts != m0->win))
        return fail(AM_EINVAL, "am_ff_upload: the TDNN expects %d channels x %d positions, the frontend splices %d x %d", m0->tdnn_D, m0->win, p->D, p->n_shifts);
    if (m0->tdnn) {
        // frames per chunk are cut so that the widest activation ((frames * positions) rows) stays near the FF chunk size;
        // every buffer gets slack rows for the windows of the padded GEMM rows (tdnn_rows_needed)
        if (p->d_iv > 0) return fail(AM_EINVAL, "am_ff_upload: TDNN members take no i-vectors");
        const int64_t frames = std::min<int64_t>(round_up(std::max<int64_t>(n, 1), GEMM_BM), tdnn_chunk_frames(m0));
        size_t in_rows = 0, act_elems = 0;
        for (auto* m : p->members) {
            in_rows = std::max(in_rows, tdnn_rows_needed(m, 0, frames));
            for (int l = 1; l <= m->n_layers; ++l) act_elems = std::max(act_elems, tdnn_rows_needed(m, l, frames) * (size_t)m->chan[l]);
        }
        AM_TRY(p->d_in.ensure(sizeof(float) * in_rows * m0->in_pad));
        AM_TRY(p->d_in_s.ensure(sizeof(float) * in_rows * m0->in_pad));   // split-f16 mode: the same rows as hi / lo half planes
        HIP_TRY(hipMemsetAsync(p->d_in_s.p, 0, sizeof(float) * in_rows * m0->in_pad, e->stream));
        AM_TRY(p->d_act0.ensure(sizeof(float) * act_elems));
        AM_TRY(p->d_act1.ensure(sizeof(float) * act_elems));
        HIP_TRY(hipMemsetAsync(p->d_in.p, 0, sizeof(float) * in_rows * m0->in_pad, e->stream));   // slack rows are read (x 0 weights): keep them finite
        HIP_TRY(hipMemsetAsync(p->d_act0.p, 0, sizeof(float) * act_elems, e->stream));
        HIP_TRY(hipMemsetAsync(p->d_act1.p, 0, sizeof(float) * act_elems, e->stream));
    } else {
    AM_TRY(p->d_in.ensure(sizeof(float) * rows * m0->in_pad));
    if (width > 0) {
        AM_TRY(p->d_act0.ensure(sizeof(float) * rows * width));
        AM_TRY(p->d_act1.ensure(sizeof(float) * rows * width));
    }
    // one plane per hidden layer for the single-launch form of the stack (a plane re-written inside a launch could be read stale)
    int planes = 0;
    for (auto* m : p->members) if (ff_stack_shape_ok(m)) planes = std::max(planes, m->n_layers);
    if (planes > 0 && ff_stack_mode() > 0) {
        if (p->d_stack.ensure(sizeof(float) * (size_t)planes * rows * width) == AM_OK &&
            p->d_ffsync.ensure(sizeof(unsigned) * (FF_SYNC_DONE + (size_t)planes * (rows / GEMM_BM))) == AM_OK) {
            p->stack_rows = rows; p->stack_width = width; p->stack_planes = planes;
            // ensembles: the second plane set and the per-member logit buffers (4.5 GB for ten 1909-class members at the core-test size);
            // where that does not fit -- or is over 48 GB -- the members simply run one after the other
            p->accs_members = 0;
            const size_t nm = p->members.size(), acc_w0 = (size_t)std::max<int>(m0->C_pad, (int)round_up(m0->C, 256));
            static const int64_t overlap_on = env_int("AM_FF_MEMBER_OVERLAP", 1);
            if (nm > 1 && overlap_on && nm * rows * acc_w0 * sizeof(float) <= ((size_t)48 << 30)) {
                if (p->d_stack2.ensure(sizeof(float) * (size_t)planes * rows * width) == AM_OK &&
                    p->d_ffsync2.ensure(sizeof(unsigned) * (FF_SYNC_DONE + (size_t)planes * (rows / GEMM_BM))) == AM_OK &&
                    p->d_accs.ensure(sizeof(float) * nm * rows * acc_w0) == AM_OK) p->accs_members = (int64_t)nm;
                else { (void)hipGetLastError(); p->d_stack2.release(); p->d_accs.release(); }
            }
        } else {   // (3 to 13 GB per pipeline at the bench sizes) no room for the planes: the per-layer launches need none
            (void)hipGetLastError();
            p->d_stack.release();
            p->stack_planes = 0;
        }
    } else p->stack_planes = 0;
    }
    const int acc_w = std::max<int>(m0->C_pad, (int)round_up(m0->C, 256));
    AM_TRY(p->d_acc.ensure(sizeof(float) * rows * acc_w));
    if (p->combine == AM_COMBINE_LOGPOST) AM_TRY(p->d_acc2.ensure(sizeof(float) * rows * acc_w));
    if (!m0->tdnn) AM_TRY(p->d_in_s.ensure(sizeof(float) * rows * m0->in_pad));
    AM_TRY(p->d_out.ensure(std::max<size_t>(sizeof(float) * n * m0->C, 16)));
    AM_TRY(sync_all(e));
    p->n = n; p->halo_lo = halo_lo; p->halo_hi = halo_hi; p->has_ivec = ivec != nullptr;
    p->mode = 1; p->results_valid = false;
    return AM_OK;
}

static int ff_member_forward_stack(am_pipeline* p, int mi, int64_t rows_pad, float* acc, int set = 0, hipStream_t on = nullptr, int epi_override = -1)
{
    // set 1 / `on` / epi_override: the overlapped form of an ensemble (its own planes and counters, the other stream, plain logits out)
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    hipStream_t st = on ? on : e->stream;
    GemmStackArgs a{};
    const int tiles_m = (int)(rows_pad / GEMM_BM);
    const size_t plane = (size_t)p->stack_rows * p->stack_width;
    int item = 0;
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        GemmStackLayer& S = a.L[l];
        S.W = Ly.W.as<float>(); S.bias = Ly.b.as<float>(); S.ldw = Ly.in_pad;
        if (l < m->n_layers) {
            S.C = (set ? p->d_stack2 : p->d_stack).as<float>() + (size_t)l * plane; S.ldc = Ly.out_pad; S.tiles_n = Ly.out_pad / 256; S.nk = Ly.in_pad / 32;
        } else {
            const bool wide = Ly.out_pad % 256 != 0;
            S.C = acc; S.ldc = m->C_pad; S.tiles_n = Ly.out_pad / (wide ? 192 : 256); S.nk = Ly.in_pad / 32;
        }
        S.cum_n = item;
        item += S.tiles_n;
    }
    a.A0 = p->d_in.as<float>(); a.lda0 = m->in_pad;
    a.n_hidden = m->n_layers; a.tiles_m = tiles_m; a.act = m->act;
    a.out_epi = epilogue_mode(p, mi, &a.out_alpha);
    if (epi_override >= 0) { a.out_epi = epi_override; a.out_alpha = 1.f; }
    unsigned* sync = (set ? p->d_ffsync2 : p->d_ffsync).as<unsigned>();
    a.abort_flag = (int*)sync; a.queues = (int*)sync + FF_SYNC_QUEUES; a.done = sync + FF_SYNC_DONE; a.cum_total = item;
    a.test_extra = env_int("AM_FF_TEST_TIMEOUT", 0) ? 1 : 0;
    if (env_int("AM_FF_STAMPS", 0)) {   // diagnostic timeline of this launch (am_ff_stack_stamps)
        AM_TRY(p->d_stamps.ensure(sizeof(unsigned long long) * 16 * 256));
        HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, sizeof(unsigned long long) * 16 * 256, st));
        a.stamps = p->d_stamps.as<unsigned long long>();
    }   // test hook: arrival targets nobody reaches -> the waits must time out
    HIP_TRY(hipMemsetAsync(sync + FF_SYNC_QUEUES, 0, sizeof(unsigned) * (FF_SYNC_DONE - FF_SYNC_QUEUES + (size_t)m->n_layers * tiles_m), st));
    const int out_pad = m->L[m->n_layers].out_pad;
    AM_TRY(run_kernel(e, K_GEMM_STACK, [&] { return launch_gemm_stack(st, a, out_pad, e->n_cus); }, st));
    p->used_stack = true;
    return AM_OK;
}

static int ff_member_forward(am_pipeline* p, int mi, int64_t rows_pad, float* acc, int64_t row0 = 0, hipStream_t on = nullptr)
{
    // rows [row0, row0 + rows_pad) of the chunk (row0 a multiple of the GEMM's row tile), on stream `on` (default: the compute stream)
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    hipStream_t st = on ? on : e->stream;
    const float* A = p->d_in.as<float>() + row0 * m->in_pad;
    acc += row0 * m->C_pad;
    int lda = m->in_pad;
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        GemmArgs g{};
        g.A = A; g.lda = lda; g.W = Ly.W.as<float>(); g.ldw = Ly.in_pad; g.bias = Ly.b.as<float>();
        g.M_pad = (int)rows_pad; g.N_pad = Ly.out_pad; g.K_pad = Ly.in_pad; g.act = m->act;
        int id;
        if (l < m->n_layers) {
            g.C = ((l & 1) ? p->d_act1 : p->d_act0).as<float>() + row0 * Ly.out_pad; g.ldc = Ly.out_pad; g.epi = EPI_ACT; g.alpha = 1.f; g.n_real = Ly.out;
            id = K_GEMM_HIDDEN;
        } else {
            g.C = acc; g.ldc = m->C_pad; g.epi = epilogue_mode(p, mi, &g.alpha);
            id = K_GEMM_OUT;
        }
        AM_TRY(run_kernel(e, id, [&] { return launch_gemm(st, g); }, st));
        A = g.C; lda = g.ldc;
    }
    return AM_OK;
}

// ---- split-f16 precision mode ---------------------------------------------------------------------
// Builds, once per model, the hi/lo half copies of the (zero-padded) weights, pre-scaled by an exact power of two so that
// max |W'| lies in [512, 1024] and the lo halves are normal f16 numbers.  Hidden widths must already be multiples of 256
// (the exact path may have padded to 384); otherwise the model stays on the exact path.
static int ensure_split_weights(am_model* m)
{
    if (m->split_state != 0) return AM_OK;
    am_engine* e = m->e;
    for (int l = 0; l < m->n_layers; ++l)
        if (m->L[l].out_pad % 256) { m->split_state = -1; return AM_OK; }
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        const float wmax = Ly.wmax > 0.f ? Ly.wmax : 1.f;
        Ly.w_scale = exp2f(floorf(log2f(1024.0f / wmax)));
        const size_t rows_s = (size_t)Ly.out_pad_s, halves = rows_s * Ly.in_pad;
        AM_TRY(Ly.Whi.ensure(halves * 2));
        AM_TRY(Ly.Wlo.ensure(halves * 2));
        AM_TRY(Ly.b_s.ensure(sizeof(float) * rows_s));
        HIP_TRY(hipMemsetAsync(Ly.Whi.p, 0, halves * 2, e->stream));
        HIP_TRY(hipMemsetAsync(Ly.Wlo.p, 0, halves * 2, e->stream));
        HIP_TRY(hipMemsetAsync(Ly.b_s.p, 0, sizeof(float) * rows_s, e->stream));
        const size_t rows_have = (size_t)std::min(Ly.out_pad, Ly.out_pad_s);   // rows beyond `out` are zero in W already
        HIP_TRY(hipMemcpyAsync(Ly.b_s.p, Ly.b.p, sizeof(float) * rows_have, hipMemcpyDeviceToDevice, e->stream));
        AM_TRY(run_kernel(e, K_SPLIT_CONVERT, [&] { return launch_split_f32(e->stream, Ly.W.as<float>(), Ly.Whi.p, Ly.Wlo.p, rows_have * Ly.in_pad, Ly.w_scale, nullptr); }));
    }
    m->split_state = 1;
    return AM_OK;
}

static bool use_split(am_pipeline* p)
{
    if (p->e->precision != AM_PRECISION_SPLIT_F16) return false;
    for (auto* m : p->members) if (m->split_state != 1) return false;
    return true;
}

static int ff_member_forward_split(am_pipeline* p, int mi, int64_t rows_pad, float* acc)
{
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    int* flag = e->d_overflow.as<int>();
    // activations travel as two half matrices inside one f32-sized buffer: hi at the base, lo right behind it
    const _Float16* Ahi = (const _Float16*)p->d_in_s.p;
    const _Float16* Alo = Ahi + (size_t)rows_pad * m->in_pad;
    int lda = m->in_pad;
    for (int l = 0; l <= m->n_layers; ++l) {
        Layer& Ly = m->L[l];
        SplitGemmArgs g{};
        g.Ahi = Ahi; g.Alo = Alo; g.lda = lda; g.Whi = Ly.Whi.p; g.Wlo = Ly.Wlo.p; g.ldw = Ly.in_pad; g.bias = Ly.b_s.as<float>();
        g.M_pad = (int)rows_pad; g.N_pad = Ly.out_pad_s; g.K_pad = Ly.in_pad; g.act = m->act; g.out_scale = 1.0f / Ly.w_scale; g.overflow = flag;
        if (l < m->n_layers) {
            _Float16* Chi = (_Float16*)((l & 1) ? p->d_act1 : p->d_act0).p;
            g.C = nullptr; g.Chi = Chi; g.Clo = Chi + (size_t)rows_pad * Ly.out_pad_s; g.ldc = Ly.out_pad_s; g.epi = EPI_ACT; g.alpha = 1.f;
            AM_TRY(run_kernel(e, K_GEMM_SPLIT, [&] { return launch_gemm_split(e->stream, g); }));
            Ahi = (const _Float16*)g.Chi; Alo = (const _Float16*)g.Clo; lda = g.ldc;
        } else {
            g.C = acc; g.ldc = p->acc_ld; g.epi = epilogue_mode(p, mi, &g.alpha);
            AM_TRY(run_kernel(e, K_GEMM_SPLIT, [&] { return launch_gemm_split(e->stream, g); }));
        }
    }
    return AM_OK;
}

// One pass over the resident shard in chunks of `chunk_rows`.  With `host_y` the posteriors of chunk c are copied back on
// the copy stream while the compute stream works on chunk c+1 (all compute is enqueued first, then the copies, so the
// overlap also holds for pageable host memory, where hipMemcpyAsync blocks the calling thread).
static int ff_run_chunks(am_pipeline* p, int64_t chunk_rows, float* host_y)
{
    am_engine* e = p->e;
    am_model* m0 = p->members[0];
    size_t n_chunks = 0;
    if (e->precision == AM_PRECISION_SPLIT_F16)
        for (auto* m : p->members) AM_TRY(ensure_split_weights(m));
    const bool split = use_split(p);
    p->ran_split = split;
    p->acc_ld = split ? (int)round_up(m0->C, 256) : m0->C_pad;
    if (split) {
        AM_TRY(e->d_overflow.ensure(16));
        HIP_TRY(hipMemsetAsync(e->d_overflow.p, 0, sizeof(int), e->stream));
    }
    if (m0->tdnn) chunk_rows = std::min<int64_t>(chunk_rows, tdnn_chunk_frames(m0));
    p->used_stack = false;
    bool stack_ok = !split && !p->stack_off && p->stack_planes > 0 && ff_stack_mode() > 0;
    for (auto* m : p->members) stack_ok = stack_ok && ff_stack_shape_ok(m) && m->n_layers <= p->stack_planes && m->max_width <= p->stack_width;
    if (stack_ok) HIP_TRY(hipMemsetAsync(p->d_ffsync.p, 0, sizeof(unsigned), e->stream));   // the abort flag: once per pass
    const bool overlap_members = stack_ok && p->members.size() > 1 && p->accs_members == (int64_t)p->members.size() && !host_y;
    if (stack_ok && p->d_ffsync2.p) HIP_TRY(hipMemsetAsync(p->d_ffsync2.p, 0, sizeof(unsigned), e->stream));   // (the second set's flag is read by ff_stack_check whenever it exists)
    for (int64_t c0 = 0; c0 < p->n; c0 += chunk_rows, ++n_chunks) {
        const int64_t nc = std::min<int64_t>(chunk_rows, p->n - c0);
        const int64_t nc_pad = round_up(nc, GEMM_BM);
        SpliceArgs s{};
        s.x = p->d_x.as<float>(); s.ext_rows = p->halo_lo + p->n + p->halo_hi; s.first = p->halo_lo + c0;
        s.ivec = p->has_ivec ? p->d_ivec.as<float>() : nullptr; s.row0 = c0; s.n = nc; s.n_pad = nc_pad;
        s.D = p->D; s.n_shifts = p->n_shifts; s.shifts = p->d_shifts.as<int>();
        s.add = p->has_ft ? p->d_add.as<float>() : nullptr; s.scale = p->has_ft ? p->d_scale.as<float>() : nullptr;
        s.d_iv = p->d_iv; s.out = p->d_in.as<float>(); s.ldo = m0->in_pad;
        if (split) {   // the frontend writes hi/lo halves directly (shared by every member)
            s.out_hi = p->d_in_s.p; s.out_lo = (_Float16*)p->d_in_s.p + (size_t)nc_pad * m0->in_pad; s.overflow = e->d_overflow.as<int>();
        }
        if (m0->tdnn) AM_TRY(run_kernel(e, K_SPLICE_FT, [&] { return launch_splice_tdnn(e->stream, s, m0->win, m0->in_pad, nc_pad); }));
        else AM_TRY(run_kernel(e, K_SPLICE_FT, [&] { return launch_splice_ft(e->stream, s); }));
        // Tile quantisation: a layer of (rows / 256) x (width / 256) tiles on 256 CUs runs in whole rounds (57,919 rows x 1024:
        // 908 tiles = 3.55 rounds, run as 4).  When the last round is mostly empty the chunk's rows are cut into two ranges (AM_FF_CHAINS, up to four; two measured best) that
        // go down the layers on as many streams: rows are independent (same bits), and the other chains' tiles fill the CUs the
        // first chain's partial rounds leave idle.  (AM_FF_CHAINS=1 switches it off.)
        int n_chains = 1;
        static const int64_t max_chains = std::min<int64_t>(4, std::max<int64_t>(1, env_int("AM_FF_CHAINS", 2)));
        if (max_chains > 1 && !host_y && !split && !m0->tdnn && nc_pad >= 16 * GEMM_BM) {
            const double rounds = (double)(nc_pad / GEMM_BM) * std::max(1, m0->max_width / 256) / 256.0;
            if (std::ceil(rounds) / rounds > 1.04) n_chains = (int)max_chains;
        }
        if (stack_ok && nc_pad <= p->stack_rows && overlap_members) {
            // Ensemble: member m + 1's stack launch goes to the other stream with its own planes and counters, so its workgroups take over
            // the CUs member m's ragged end leaves idle (a launch ends with ~one output tile per workgroup of imbalance: 4 % of a 4x1024
            // member at the core-test size, ten times per 10-fold ensemble).  No member waits for another: each writes plain logits to a
            // buffer of its own, and the finish pass adds them up in member order -- the float operations of the one-after-the-other
            // form (C = C + ...), so the same bits.  Streams alternate, so member m + 2 follows member m and re-uses its planes.
            if (!p->ev_fork) { HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming)); }
            const int nm = (int)p->members.size();
            const int64_t stride = (int64_t)p->stack_rows * p->acc_ld;
            HIP_TRY(hipEventRecord(p->ev_fork, e->stream));      // the frontend's rows are written (and the previous chunk's finish has read the buffers)
            HIP_TRY(hipStreamWaitEvent(e->stream2, p->ev_fork, 0));
            for (int mi = 0; mi < nm; ++mi)
                AM_TRY(ff_member_forward_stack(p, mi, nc_pad, p->d_accs.as<float>() + mi * stride, mi & 1, (mi & 1) ? e->stream2 : e->stream, EPI_LINEAR));
            HIP_TRY(hipEventRecord(p->ev_join, e->stream2));
            HIP_TRY(hipStreamWaitEvent(e->stream, p->ev_join, 0));
            FinishArgs f{};
            f.in = p->d_accs.as<float>(); f.ldin = p->acc_ld; f.n_in = nm; f.in_stride = stride;
            f.out = p->d_out.as<float>() + c0 * m0->C; f.ldout = m0->C; f.n = nc; f.C = m0->C;
            if (p->combine == AM_COMBINE_LOGPOST) f.member_norm = 1;
            else if (p->has_master) f.first_scale = (float)(nm - 1);
            f.divisor = final_divisor(p);
            if (p->has_rpl) { f.rpl_w = p->d_rplw.as<float>(); f.rpl_b = p->d_rplb.as<float>(); f.rpl_lb = p->d_rpllb.as<float>(); }
            if (p->has_prior) f.prior = p->d_prior.as<float>();
            f.normalise = p->normalise;
            AM_TRY(run_kernel(e, K_FINISH, [&] { return launch_finish(e->stream, f); }));
        } else if (stack_ok && nc_pad <= p->stack_rows) {
            // one launch per member: a work queue over (layer, row tile, column tile) -- the layers' tails overlap, one partial round per network
            for (int mi = 0; mi < (int)p->members.size(); ++mi) {
                AM_TRY(ff_member_forward_stack(p, mi, nc_pad, p->d_acc.as<float>()));
                if (p->combine == AM_COMBINE_LOGPOST) AM_TRY(member_finish(p, mi, nc));
            }
            AM_TRY(final_finish(p, nc, p->d_out.as<float>() + c0 * m0->C, nullptr));
        } else if (n_chains > 1) {
            if (!p->ev_fork) { HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming)); }
            for (auto& ev : p->ev_chain) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipStream_t streams[4] = {e->stream, e->stream2, e->chain[0], e->chain[1]};
            hipEvent_t joins[4] = {nullptr, p->ev_join, p->ev_chain[0], p->ev_chain[1]};
            HIP_TRY(hipEventRecord(p->ev_fork, e->stream));
            const int64_t tiles_m = nc_pad / GEMM_BM;
            for (int br = 0; br < n_chains; ++br) {
                hipStream_t st = streams[br];
                const int64_t t0 = tiles_m * br / n_chains, t1 = tiles_m * (br + 1) / n_chains;
                const int64_t r0 = t0 * GEMM_BM, rp = (t1 - t0) * GEMM_BM, nr = std::max<int64_t>(0, std::min(nc, r0 + rp) - r0);
                if (br > 0) HIP_TRY(hipStreamWaitEvent(st, p->ev_fork, 0));
                for (int mi = 0; mi < (int)p->members.size(); ++mi) {
                    AM_TRY(ff_member_forward(p, mi, rp, p->d_acc.as<float>(), r0, st));
                    if (p->combine == AM_COMBINE_LOGPOST && nr > 0) AM_TRY(member_finish(p, mi, nr, r0, st));
                }
                if (nr > 0) AM_TRY(final_finish(p, nr, p->d_out.as<float>() + (c0 + r0) * m0->C, nullptr, r0, st));
                if (br > 0) HIP_TRY(hipEventRecord(joins[br], st));
            }
            for (int br = 1; br < n_chains; ++br) HIP_TRY(hipStreamWaitEvent(e->stream, joins[br], 0));
        } else {
        for (int mi = 0; mi < (int)p->members.size(); ++mi) {
            if (m0->tdnn && split) AM_TRY(tdnn_member_forward_split(p, mi, nc_pad, p->d_acc.as<float>()));
            else if (m0->tdnn) AM_TRY(tdnn_member_forward(p, mi, nc_pad, p->d_acc.as<float>()));
            else if (split) AM_TRY(ff_member_forward_split(p, mi, nc_pad, p->d_acc.as<float>()));
            else AM_TRY(ff_member_forward(p, mi, nc_pad, p->d_acc.as<float>()));
            if (p->combine == AM_COMBINE_LOGPOST) AM_TRY(member_finish(p, mi, nc));
        }
        AM_TRY(final_finish(p, nc, p->d_out.as<float>() + c0 * m0->C, nullptr));
        }
        if (host_y) {
            if (p->chunk_done.size() <= n_chunks) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                p->chunk_done.push_back(ev);
            }
            HIP_TRY(hipEventRecord(p->chunk_done[n_chunks], e->stream));
        }
    }
    if (host_y) {
        size_t c = 0;
        for (int64_t c0 = 0; c0 < p->n; c0 += chunk_rows, ++c) {
            const int64_t nc = std::min<int64_t>(chunk_rows, p->n - c0);
            HIP_TRY(hipStreamWaitEvent(e->stream2, p->chunk_done[c], 0));
            HIP_TRY(hipMemcpyAsync(host_y + c0 * m0->C, p->d_out.as<float>() + c0 * m0->C, sizeof(float) * nc * m0->C, hipMemcpyDeviceToHost, e->stream2));
        }
        HIP_TRY(hipEventRecord(e->s2_tail, e->stream2));
        e->s2_dirty = true;
    }
    return AM_OK;
}

// After a synchronisation: did a gemm_stack_kernel launch give up on a wait?  Then the pass is invalid: repeat it on the
// per-layer launches (same bits), keep this pipeline on them, and say so -- never a wrong result, never a hung GPU.
static int ff_stack_check(am_pipeline* p, bool* reran = nullptr, int64_t chunk_rows = 0, float* host_y = nullptr)
{
    if (reran) *reran = false;
    if (p->mode != 1 || !p->used_stack || !p->d_ffsync.p) return AM_OK;
    int flag = 0, flag2 = 0;
    HIP_TRY(hipMemcpy(&flag, p->d_ffsync.p, sizeof(int), hipMemcpyDeviceToHost));
    if (p->d_ffsync2.p) HIP_TRY(hipMemcpy(&flag2, p->d_ffsync2.p, sizeof(int), hipMemcpyDeviceToHost));
    if (!flag && !flag2) return AM_OK;
    fprintf(stderr, "libamfwd: the single-launch Linear stack gave up on an inter-workgroup wait; repeating the pass on the per-layer launches\n");
    ++g_lstm_fallbacks;
    p->stack_off = true;
    const int rc = ff_run_chunks(p, chunk_rows > 0 ? chunk_rows : FF_CHUNK_ROWS, host_y);
    if (rc != AM_OK) { p->results_valid = false; return rc; }
    AM_TRY(sync_all(p->e));
    if (reran) *reran = true;
    return AM_OK;
}

extern "C" int am_ff_stack_stamps(am_pipeline* p, unsigned long long* out)
{
    if (!p || !out || p->mode != 1 || !p->used_stack || !p->d_stamps.p) return fail(AM_EINVAL, "am_ff_stack_stamps: no stamped single-launch pass (AM_FF_STAMPS=1, am_ff_run)");
    HIP_TRY(hipSetDevice(p->e->device));
    AM_TRY(sync_all(p->e));
    HIP_TRY(hipMemcpy(out, p->d_stamps.p, sizeof(unsigned long long) * 16 * 256, hipMemcpyDeviceToHost));
    return AM_OK;
}

extern "C" int am_ff_run(am_pipeline* p)
{
    if (!p || p->mode != 1) return fail(AM_EINVAL, "am_ff_run: nothing uploaded (am_ff_upload)");
    HIP_TRY(hipSetDevice(p->e->device));
    p->fell_back = false;
    AM_TRY(ff_run_chunks(p, FF_CHUNK_ROWS, nullptr));
    p->results_valid = true;
    if (p->ran_split) {   // split-f16 mode: an activation that does not fit f16 sends the whole pass to the exact f32 path
        int over = 0;     // (this read synchronises; the exact path stays asynchronous)
        AM_TRY(am_pipeline_overflowed(p, &over));
        if (over) {
            const int keep = p->e->precision;
            p->e->precision = AM_PRECISION_F32;
            const int rc = ff_run_chunks(p, FF_CHUNK_ROWS, nullptr);
            p->e->precision = keep;
            AM_TRY(rc);
            p->fell_back = true;
        }
    }
    return AM_OK;
}

extern "C" int am_ff_download(am_pipeline* p, float* y)
{
    if (!p || p->mode != 1 || !y) return fail(AM_EINVAL, "am_ff_download: nothing to download");
    if (!p->results_valid) return fail(AM_EINVAL, "am_ff_download: the pipeline has not been run since its last upload / configuration change");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(join_streams(e));
    if (p->n > 0) HIP_TRY(hipMemcpyAsync(y, p->d_out.p, sizeof(float) * p->n * p->members[0]->C, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    bool reran = false;
    AM_TRY(ff_stack_check(p, &reran));   // (checked after the copy: costs the normal case nothing)
    if (reran && p->n > 0) {
        HIP_TRY(hipMemcpyAsync(y, p->d_out.p, sizeof(float) * p->n * p->members[0]->C, hipMemcpyDeviceToHost, e->stream));
        AM_TRY(sync_all(e));
    }
    return AM_OK;
}

extern "C" int am_forward_ff(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi, float* y)
{
    AM_TRY(am_ff_upload(p, x, ivec, n, halo_lo, halo_hi));
    if (!y) return fail(AM_EINVAL, "am_forward_ff: y is NULL");
    p->fell_back = false;
    AM_TRY(ff_run_chunks(p, FF_PIPE_ROWS, y));   // download of chunk c overlaps the compute of chunk c+1
    p->results_valid = true;
    AM_TRY(sync_all(p->e));
    AM_TRY(ff_stack_check(p, nullptr, FF_PIPE_ROWS, y));   // (repeats the pass on the per-layer launches if a stack launch gave up)
    int over = 0;
    AM_TRY(am_pipeline_overflowed(p, &over));
    if (over) {   // an activation did not fit f16: repeat the pass on the exact f32 path
        const int keep = p->e->precision;
        p->e->precision = AM_PRECISION_F32;
        const int rc = ff_run_chunks(p, FF_PIPE_ROWS, y);
        p->e->precision = keep;
        AM_TRY(rc);
        p->fell_back = true;
        AM_TRY(sync_all(p->e));
    }
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// recurrent path
// ----------------------------------------------------------------------------------------------------
extern "C" int am_lstm_upload(am_pipeline* p, const float* x, const float* ivec, const int64_t* offsets, int n_utts, int timedelay)
{
    AM_TRY(check_ready(p, AM_NET_LSTM));
    if (!offsets || n_utts < 0 || timedelay < 0 || (!x && offsets[n_utts] > 0)) return fail(AM_EINVAL, "am_lstm_upload: bad argument");
    if (p->n_shifts != 1) return fail(AM_EINVAL, "am_lstm_upload: the recurrent path takes un-spliced frames (n_shifts must be 1)");
    if ((p->d_iv > 0) != (ivec != nullptr)) return fail(AM_EINVAL, "am_lstm_upload: i-vector argument does not match the frontend");
    if (offsets[0] != 0) return fail(AM_EINVAL, "am_lstm_upload: offsets[0] must be 0");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    am_model* m0 = p->members[0];
    const int64_t N = offsets[n_utts];
    std::vector<int64_t> len(n_utts);
    for (int u = 0; u < n_utts; ++u) {
        const int64_t L = offsets[u + 1] - offsets[u];
        if (L < 0) return fail(AM_EINVAL, "am_lstm_upload: offsets must be non-decreasing");
        // predict_folds.py:39 edge-pads every utterance by `timedelay` rows, but its time loop stores only rows of
        // utterances with (un-padded) length > t (predict_folds.py:50,61): the padded steps never reach an output and
        // the last `timedelay` output frames of every utterance stay at their initial zeros.  So the device runs L steps.
        len[u] = L;
    }
    // utterance slots: longest first (any order gives identical rows; this one keeps active slots contiguous)
    std::vector<int> order(n_utts);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });
    const int64_t T = n_utts > 0 ? len[order[0]] : 0;
    p->step_off.assign(T + 1, 0);
    p->n_active.assign(T, 0);
    int active = n_utts;
    for (int64_t t = 0; t < T; ++t) {
        while (active > 0 && len[order[active - 1]] <= t) --active;
        p->n_active[t] = active;
        p->step_off[t + 1] = p->step_off[t] + active;
    }
    const int64_t NP = p->step_off[T];
    if (NP > INT32_MAX || N > INT32_MAX) return fail(AM_EINVAL, "am_lstm_upload: shard too large");
    std::vector<int> src(std::max<int64_t>(NP, 1)), dst(std::max<int64_t>(NP, 1));
    for (int64_t t = 0; t < T; ++t)
        for (int s = 0; s < p->n_active[t]; ++s) {
            const int u = order[s];
            const int64_t r = p->step_off[t] + s;
            src[r] = (int)(offsets[u] + t);
            dst[r] = t >= timedelay ? (int)(offsets[u] + t - timedelay) : -1;  // predict_folds.py:60-61
        }
    const int64_t rows_pad = round_up(std::max<int64_t>(NP, 1), GEMM_BM);
    const int64_t rows_alloc = rows_pad + GEMM_BM;
    int upad = 0, gw = 0;
    for (auto* m : p->members)
        for (int l = 0; l < m->n_layers; ++l) { upad = std::max(upad, m->L[l].U_pad); gw = std::max(gw, m->L[l].out_pad); }
    AM_TRY(p->d_x.ensure(std::max<size_t>(sizeof(float) * N * p->D, 16)));
    if (N > 0) HIP_TRY(hipMemcpyAsync(p->d_x.p, x, sizeof(float) * N * p->D, hipMemcpyHostToDevice, e->stream));
    if (ivec) {
        AM_TRY(p->d_ivec.ensure(std::max<size_t>(sizeof(float) * N * p->d_iv, 16)));
        if (N > 0) HIP_TRY(hipMemcpyAsync(p->d_ivec.p, ivec, sizeof(float) * N * p->d_iv, hipMemcpyHostToDevice, e->stream));
    }
    AM_TRY(p->d_src.ensure(sizeof(int) * src.size()));
    AM_TRY(p->d_dst.ensure(sizeof(int) * dst.size()));
    HIP_TRY(hipMemcpyAsync(p->d_src.p, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(p->d_dst.p, dst.data(), sizeof(int) * dst.size(), hipMemcpyHostToDevice, e->stream));
    AM_TRY(p->d_in.ensure(sizeof(float) * rows_alloc * m0->in_pad));
    AM_TRY(p->d_G.ensure(sizeof(float) * rows_alloc * gw));
    int max_layers = 0;
    for (auto* m : p->members) max_layers = std::max(max_layers, m->n_layers);
    AM_TRY(p->d_H.ensure(sizeof(float) * rows_alloc * upad * max_layers));      // one H plane per layer
    HIP_TRY(hipMemsetAsync(p->d_H.p, 0, sizeof(float) * rows_alloc * upad * max_layers, e->stream));
    HIP_TRY(hipMemsetAsync(p->d_G.p, 0, sizeof(float) * rows_alloc * gw, e->stream));
    // per layer: the LSTM cell state, or three per-slot planes of the other cells (Z / r*h, or c ping-pong / o), 16 rows of slack
    const size_t c_bytes = sizeof(float) * 3 * (round_up(std::max(n_utts, 1), 16) + 16) * upad * max_layers;
    AM_TRY(p->d_c.ensure(c_bytes));
    HIP_TRY(hipMemsetAsync(p->d_c.p, 0, c_bytes, e->stream));
    AM_TRY(p->d_acc.ensure(sizeof(float) * rows_pad * m0->C_pad));
    if (p->combine == AM_COMBINE_LOGPOST) AM_TRY(p->d_acc2.ensure(sizeof(float) * rows_pad * m0->C_pad));
    AM_TRY(p->d_out.ensure(std::max<size_t>(sizeof(float) * N * m0->C, 16)));
    // persistent forms: plan the row groups per LSTM member and upload the step / arrival tables
    for (auto& pl : p->persist) pl.release();
    p->persist.clear();
    p->persist.resize(p->members.size());
    const int pmode = lstm_persist_mode();
    bool any_persist = false, any_recstream = false;
    if (pmode && !gate_of(e).off.load() && T > 0 && (size_t)rows_alloc * gw * sizeof(float) < ((size_t)1 << 31)) {
        std::vector<int> nact(p->n_active.begin(), p->n_active.end()), soff(p->step_off.begin(), p->step_off.end());
        AM_TRY(p->d_nact.ensure(sizeof(int) * nact.size()));
        AM_TRY(p->d_stepoff.ensure(sizeof(int) * soff.size()));
        HIP_TRY(hipMemcpyAsync(p->d_nact.p, nact.data(), sizeof(int) * nact.size(), hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(p->d_stepoff.p, soff.data(), sizeof(int) * soff.size(), hipMemcpyHostToDevice, e->stream));
        AM_TRY(p->d_sync.ensure(sizeof(unsigned) * 64));
        std::vector<int> toff(nact.size());
        p->total_tiles = 0;
        for (size_t t = 0; t < nact.size(); ++t) { toff[t] = (int)p->total_tiles; p->total_tiles += (nact[t] + 15) / 16; }
        AM_TRY(p->d_tileoff.ensure(sizeof(int) * std::max<size_t>(toff.size(), 1)));
        HIP_TRY(hipMemcpyAsync(p->d_tileoff.p, toff.data(), sizeof(int) * toff.size(), hipMemcpyHostToDevice, e->stream));   // (synchronised below, with the tables above)
        std::vector<std::vector<unsigned>> staging;   // alive until the copies below have been synchronised
        for (size_t mi = 0; mi < p->members.size(); ++mi) {
            am_model* m = p->members[mi];
            if (m->cell != AM_CELL_LSTM) {
                // GRU / MGRU / peephole stacks: the per-tick phase kernels, unless AM_REC_STREAM asks for the persistent form (rec_stream_kernel): measured
                // on the 3 x 512 recipes it is 17-32 % SLOWER than they are (MGRU 11.7 against 10.0 ms, GRU 22.5 / 17.5, peephole 32.9 / 24.9),
                // so it is there to be measured, not used.  1 = when its row groups fill the chip -- at least two groups, or a file of at most four row
                // tiles (1024-unit stacks: one group of 192 workgroups would take every row in three passes) --, 2 = whenever it fits.
                const int64_t rs_mode = env_int("AM_REC_STREAM", 0);   // (read at every upload, like AM_LSTM_PERSIST)
                // A phase of an upper layer can run as TWO roles: the input part (h of the layer below times W_x -- no dependency on the recurrence, it runs
                // ahead) and the state part, which starts its accumulators from what the input part left: the chain a step has to wait for is half as
                // long, at the price of more workgroups per row group, i.e. more row tiles per workgroup.  Taken when batches x tiles of the longest
                // chain come out smaller (3 x 512 MGRU, 192 utterances: 8 batches x 3 tiles -> 5 x 4; GRU: 16 x 4 -> 10 x 6).  AM_REC_SPLIT=0 never,
                // 2 whenever it fits.
                const int64_t split_mode = env_int("AM_REC_SPLIT", 0);   // (read at every upload: the tests compare both forms in one process)
                const int n_phase = m->cell == AM_CELL_MGRU ? 1 : 2;
                const int U_pad = m->L[0].U_pad, NL = m->n_layers;
                const int ncA = (m->cell == AM_CELL_PEEPHOLE ? 4 : 2) * U_pad / 64, ncB = U_pad / 64;
                const int swgs = rs_mode ? rec_stream_resident_wgs(e->n_cus) : 0;
                const int tiles = (n_utts + 15) / 16;
                struct Opt { int per_group = 0, n_roles = 0, NG = 0; long cost = 0; };
                auto option = [&](bool split) {
                    Opt o;
                    const int per_layer = ncA + (n_phase == 2 ? ncB : 0);
                    o.per_group = split ? per_layer * (1 + 2 * (NL - 1)) : NL * per_layer;
                    o.n_roles = split ? n_phase * (1 + 2 * (NL - 1)) : NL * n_phase;
                    o.NG = o.per_group > 0 ? swgs / o.per_group : 0;
                    o.NG = std::min(o.NG, std::max(tiles, 1));
                    while (o.NG > 0 && o.n_roles * o.NG > 62) --o.NG;   // arrival counters: d_sync holds 63
                    if (o.n_roles > RS_MAX_ROLES) o.NG = 0;
                    if (o.NG < 1) return o;
                    const int nbx0 = m->L[0].Kx / LSTM_KB, nbu = U_pad / LSTM_KB;
                    const int state = m->cell == AM_CELL_PEEPHOLE ? 3 * nbu : n_phase * nbu;       // k-batches of a step's chain without / with the input part
                    const int chain0 = state + n_phase * nbx0, chain = NL > 1 ? (split ? state : state + (m->cell == AM_CELL_PEEPHOLE ? 1 : n_phase) * nbu) : 0;
                    o.cost = (long)std::max(chain0, chain) * ((tiles + o.NG - 1) / o.NG);
                    return o;
                };
                Opt o = option(false);
                bool split = false;
                if (split_mode && NL > 1 && m->cell != AM_CELL_PEEPHOLE) {
                    const Opt os = option(true);
                    if (os.NG >= 1 && (o.NG < 1 || split_mode == 2 || os.cost < o.cost)) { o = os; split = true; }
                }
                const int NG = o.NG;
                if (NG < 1 || (rs_mode != 2 && NG < 2 && tiles > 4)) continue;
                auto& pl = p->persist[mi];
                RecStreamArgs& ra = pl.rsa;
                ra = RecStreamArgs{};
                ra.n_roles = o.n_roles; ra.NG = NG; ra.wgs_per_group = o.per_group; ra.T = (int)T; ra.U_pad = U_pad; ra.U_real = m->L[0].U; ra.act = m->act; ra.max_wgs = swgs;
                AM_TRY(pl.d_roles.ensure(sizeof(RecStreamRole) * RS_MAX_ROLES));
                pl.split = split;
                if (split)   // per upper layer: [total tiles][64-column workgroups of its phases][8 KB]
                    AM_TRY(pl.d_xplanes.ensure(sizeof(float) * 2048 * (size_t)p->total_tiles * (size_t)(ncA + (n_phase == 2 ? ncB : 0)) * (size_t)(NL - 1)));
                pl.form = 3; any_persist = true; any_recstream = true;
                continue;
            }
            bool fits = m->L[0].out_pad == 4 * m->L[0].U_pad;
            for (int l = 0; l < m->n_layers && fits; ++l) fits = m->L[l].U_pad == m->L[0].U_pad && (l == 0 || m->L[l].in_pad == m->L[l].U_pad);
            if (!fits) continue;
            auto& pl = p->persist[mi];
            const int U_pad = m->L[0].U_pad;
            const int max_wgs = lstm_persist_resident_wgs(U_pad, e->n_cus);
            if (lstm_persist_capacity(U_pad, 1, max_wgs) <= 0) {
                // no register-resident form for this width: the streamed persistent form (lstm_stream_kernel: 768 / 1024 units), every utterance in
                // one launch per segment of as many layers as the device holds workgroups for (4 x 1024: one launch of 256 workgroups)
                static const bool stream_ok = env_int("AM_LSTM_STREAM", 1) != 0;
                const int swgs = stream_ok ? lstm_stream_resident_wgs(U_pad, e->n_cus) : 0;
                const int lmax = lstm_stream_layers(U_pad, swgs);
                if (lmax < 1) continue;
                bool packed = true;   // (the model holds its weights in the kernel's stream order: am_model_create)
                for (int l = 0; l < m->n_layers; ++l) packed = packed && m->L[l].Wlats.p && (l == 0 || m->L[l].Ws.p);
                if (!packed) continue;
                for (int l0 = 0; l0 < m->n_layers; l0 += lmax) {
                    LstmPersistArgs pa{};
                    pa.n_layers = std::min(lmax, m->n_layers - l0); pa.U_pad = U_pad; pa.T = (int)T; pa.slot_base = 0; pa.slot_cap = n_utts;
                    pa.max_wgs = swgs; pa.ext_upward = l0 > 0 ? 1 : 0;
                    pl.chunks.push_back(pa);
                    pl.layer0.push_back(l0);
                }
                pl.form = 2; any_persist = true;
                continue;
            }
            // Layer segments.  A launch holds every layer of its segment at once, so the slots it can take shrink with the layer count (3 x 512:
            // 192 utterances, 4 x 512: 128, 5 x 512: 64), and a file that does not fit runs as several launches over chunks of its utterances, each
            // walking the whole time axis of its longest utterance.  A deep stack is cheaper cut the other way: the first launch runs as many layers
            // as still take ALL the utterances, the next one continues from its (complete) H plane -- its first layer multiplies that plane
            // like any upper layer (ext_upward) --, and so on: one time axis per segment instead of one per chunk (5 x 512, 192 utterances:
            // 3 chunks x 5 layers -> segments of 3 + 2 layers).  AM_LSTM_SEGMENTS=0 keeps one segment.
            std::vector<std::pair<int, int>> segs;   // (first layer, layers)
            {
                static const bool seg_ok = env_int("AM_LSTM_SEGMENTS", 1) != 0;
                int l0 = 0;
                while (l0 < m->n_layers) {
                    int nl = m->n_layers - l0;
                    if (seg_ok) while (nl > 1 && lstm_persist_capacity(U_pad, nl, max_wgs, l0 > 0) < n_utts) --nl;
                    segs.emplace_back(l0, nl);
                    l0 += nl;
                }
                // (a cut buys nothing when every segment still needs as many chunks as the whole stack would: keep the stack whole then)
                const int cap_all = lstm_persist_capacity(U_pad, m->n_layers, max_wgs);
                if (segs.size() > 1 && cap_all > 0) {
                    size_t launches = 0;
                    for (auto& sg : segs) { const int c = lstm_persist_capacity(U_pad, sg.second, max_wgs, sg.first > 0); launches += c > 0 ? (n_utts + c - 1) / c : 1u << 20; }
                    if (launches >= (size_t)(n_utts + cap_all - 1) / cap_all) { segs.clear(); segs.emplace_back(0, m->n_layers); }
                }
            }
            bool ok = true;
            for (auto& sg : segs) {
                const int cap = lstm_persist_capacity(U_pad, sg.second, max_wgs, sg.first > 0);
                if (cap <= 0) { ok = false; break; }
                for (int base = 0; base < n_utts && ok; base += cap) {
                    LstmPersistArgs pa{};
                    pa.n_layers = sg.second; pa.U_pad = U_pad; pa.T = (int)T; pa.slot_base = base; pa.slot_cap = std::min(cap, n_utts - base);
                    pa.max_wgs = max_wgs; pa.ext_upward = sg.first > 0 ? 1 : 0;
                    if (!plan_lstm_persist(pa, pa.slot_cap, nact.data(), nullptr)) { ok = false; break; }
                    staging.emplace_back(lstm_persist_cum_size(pa));
                    plan_lstm_persist(pa, pa.slot_cap, nact.data(), staging.back().data());
                    if (env_int("AM_LSTM_TEST_TIMEOUT", 0) != 0)   // test hook: an arrival target nobody can reach -> the bounded waits must give up
                        for (int64_t t = 0; t < T; ++t) staging.back()[(size_t)(pa.n_groups_total - 1) * T + t] += 1;
                    pl.cums.emplace_back();
                    AM_TRY(pl.cums.back().ensure(sizeof(unsigned) * staging.back().size()));
                    HIP_TRY(hipMemcpyAsync(pl.cums.back().p, staging.back().data(), sizeof(unsigned) * staging.back().size(), hipMemcpyHostToDevice, e->stream));
                    pl.chunks.push_back(pa);
                    pl.layer0.push_back(sg.first);
                }
                if (!ok) break;
            }
            if (ok) { pl.form = 1; any_persist = true; } else pl.release();
        }
        AM_TRY(sync_all(e));   // the staging vectors go out of scope here
    }
    if (any_persist) AM_TRY(p->d_cplane.ensure(sizeof(float) * rows_alloc * upad * max_layers));   // written before it is read: no memset
    if (any_recstream) {   // pad units are never stored: their columns must read as zeros
        AM_TRY(p->d_recplanes.ensure(sizeof(float) * 2 * rows_alloc * upad * max_layers));
        HIP_TRY(hipMemsetAsync(p->d_recplanes.p, 0, sizeof(float) * 2 * rows_alloc * upad * max_layers, e->stream));
    }
    AM_TRY(sync_all(e));
    p->drop_graph();
    p->n = N; p->n_packed = NP; p->n_slots = n_utts; p->has_ivec = ivec != nullptr; p->timedelay = timedelay;
    p->mode = 2;
    return AM_OK;
}

static int lstm_enqueue(am_pipeline* p);
extern "C" int am_lstm_fallback_count(void) { return g_lstm_fallbacks.load(); }

extern "C" int am_lstm_persist_enabled(int device)
{
    if (device < 0 || device >= AM_MAX_DEVICES) return 0;
    return (lstm_persist_mode() != 0 && !g_gate[device].off.load()) ? 1 : 0;
}

extern "C" int am_lstm_persist_reenable(int device)
{
    if (device < 0 || device >= AM_MAX_DEVICES) return fail(AM_EINVAL, "am_lstm_persist_reenable: bad device %d", device);
    g_gate[device].off.store(false);
    return AM_OK;
}

extern "C" int am_pipeline_recurrent_form(am_pipeline* p, int* n_persistent, int* n_members)
{
    if (!p || !n_persistent || !n_members || p->mode != 2) return fail(AM_EINVAL, "am_pipeline_recurrent_form: nothing uploaded (am_lstm_upload)");
    *n_members = (int)p->members.size();
    *n_persistent = 0;
    for (auto& pl : p->persist) *n_persistent += pl.form != 0;
    return AM_OK;
}

// After a synchronisation: did the persistent recurrent kernel give up on a wait (lstm_persist_kernel's abort flag)?
static int persist_check(am_pipeline* p)
{
    if (p->mode != 2 || !p->used_persist || !p->d_sync.p) return AM_OK;
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, p->d_sync.as<int>() + 63, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        // A wait ran into its 2 s bound (the kernel's workgroups were not all resident -- e.g. another process holds CUs -- or a
        // hand-off was lost).  Every workgroup has left; the pass is invalid.  Repeat it once on the per-step kernels (same
        // bits), keep this upload on them, and say so: never a wrong result, never a hung GPU.
        fprintf(stderr, "libamfwd: the persistent recurrent kernel gave up on an inter-workgroup wait; repeating the pass on the per-step kernels\n"
                        "libamfwd: device %d stays on the per-step recurrent kernels for the rest of this process (am_lstm_persist_reenable lifts that)\n", p->e->device);
        ++g_lstm_fallbacks;
        gate_of(p->e).off.store(true);
        for (auto& pl : p->persist) pl.release();
        p->drop_graph();
        p->results_valid = true;
        const int rc = lstm_enqueue(p);
        if (rc != AM_OK) { p->results_valid = false; return rc; }
        AM_TRY(sync_all(p->e));
    }
    return AM_OK;
}

extern "C" int am_lstm_run(am_pipeline* p)
{
    if (!p || p->mode != 2) return fail(AM_EINVAL, "am_lstm_run: nothing uploaded (am_lstm_upload)");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    p->results_valid = true;   // (cleared again below if the enqueue fails)
    if (p->n_packed == 0) return AM_OK;
    // a give-up on this device is sticky for EVERY pipeline, also those that planned the persistent form before it happened (another
    // engine or shard on the same card): they drop their plan here instead of each paying its own 2-s stall
    if (gate_of(e).off.load())
        for (auto& pl : p->persist) if (pl.form != 0) pl.release();
    auto eager = [&] { const int rc = lstm_enqueue(p); if (rc != AM_OK) p->results_valid = false; return rc; };
    // with the persistent form a pass is a handful of launches on two streams: nothing to gain from a graph, and a replayed
    // graph serialises the second stream's branch (measured: 10.2 ms replayed against 8.8 ms eager)
    // (and a pass with a persistent launch in it waits for another engine's event, which a stream capture does not admit)
    bool any_persistent = false;
    for (auto& pl : p->persist) any_persistent = any_persistent || pl.form != 0;
    if (e->profile || any_persistent || !lstm_use_graph()) return eager();   // per-kernel events need eager launches
    if (p->graph_exec) { HIP_TRY(hipGraphLaunch(p->graph_exec, e->stream)); return AM_OK; }
    if (p->lstm_runs++ == 0) return eager();          // one-shot use: eager (no capture / instantiate cost)
    // second run on the same upload: capture the launch sequence once, then replay it
    HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    const int rc = lstm_enqueue(p);
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(e->stream, &g);
    if (rc != AM_OK) { if (g) (void)hipGraphDestroy(g); p->results_valid = false; return rc; }
    if (ce != hipSuccess) { p->results_valid = false; return fail(AM_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce)); }
    p->graph = g;
    HIP_TRY(hipGraphInstantiate(&p->graph_exec, p->graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(p->graph_exec, e->stream));
    return AM_OK;
}

// GRU / MGRU / peephole-LSTM stack: wavefront over (layer, time) like the LSTM one, with one or two dependent phases per
// tick (launch_rec_step).  Tick d runs layer l at t = d - l; cells with a reset gate or an output peephole run phase A of
// every active layer, then phase B.
static int rec_stack(am_pipeline* p, am_model* m, size_t h_stride)
{
    am_engine* e = p->e;
    const int NL = m->n_layers, U_pad = m->L[0].U_pad;
    const int64_t T = (int64_t)p->n_active.size();
    const size_t plane = (size_t)(round_up(std::max(p->n_slots, 1), 16) + 16) * U_pad;
    float* Hall = p->d_H.as<float>();
    float* S = p->d_c.as<float>();
    const int phases = m->cell == AM_CELL_MGRU ? 1 : 2;
    for (int64_t d = 0; d < T + NL - 1; ++d)
        for (int ph = 0; ph < phases; ++ph) {
            RecStepArgs w{};
            w.U_pad = U_pad; w.act = m->act; w.U_real = m->L[0].U;
            int epi = REC_MGRU;
            for (int l = 0; l < NL; ++l) {
                const int64_t t = d - l;
                if (t < 0 || t >= T) continue;
                Layer& Ly = m->L[l];
                RecLayerStep& s = w.L[w.n++];
                const bool first = t == 0;
                float* H = Hall + l * h_stride;
                const float* x = l == 0 ? p->d_in.as<float>() + p->step_off[t] * (int64_t)m->in_pad : Hall + (l - 1) * h_stride + p->step_off[t] * (int64_t)U_pad;
                const float* hprev = first ? H : H + p->step_off[t - 1] * (int64_t)U_pad;
                float* hout = H + p->step_off[t] * (int64_t)U_pad;
                float* S0 = S + (size_t)(3 * l) * plane; float* S1 = S0 + plane; float* S2 = S1 + plane;
                s.n_active = p->n_active[t]; s.first = first;
                s.A[0] = x; s.K[0] = Ly.Kx;
                if (m->cell == AM_CELL_MGRU) {
                    epi = REC_MGRU;
                    s.A[1] = hprev; s.K[1] = first ? 0 : U_pad;
                    s.W = Ly.WA.as<float>(); s.Ws = Ly.WAs.as<float>(); s.ldw = Ly.Kx + U_pad;
                    s.bias = (first ? Ly.bA_first : Ly.bA_rest).as<float>();
                    s.E0 = hprev; s.O0 = hout;
                } else if (m->cell == AM_CELL_GRU) {
                    float* Z = S0; float* RH = S1;
                    s.ldw = Ly.Kx + U_pad;
                    if (ph == 0) {
                        epi = REC_GRU_A;
                        s.A[1] = hprev; s.K[1] = first ? 0 : U_pad;
                        s.W = Ly.WA.as<float>(); s.Ws = Ly.WAs.as<float>(); s.bias = (first ? Ly.bA_first : Ly.bA_rest).as<float>();
                        s.E0 = hprev; s.O0 = Z; s.O1 = RH;
                    } else {
                        epi = REC_GRU_B;
                        s.A[1] = RH; s.K[1] = first ? 0 : U_pad;
                        s.W = Ly.WB.as<float>(); s.Ws = Ly.WBs.as<float>(); s.bias = (first ? Ly.bB_first : Ly.bB_rest).as<float>();
                        s.E0 = hprev; s.E1 = Z; s.O0 = hout;
                    }
                } else {   // AM_CELL_PEEPHOLE: c ping-pongs between two planes by time parity (other blocks read c[t-1] as a K segment)
                    float* Cprev = (t & 1) ? S0 : S1; float* Cnext = (t & 1) ? S1 : S0; float* Opre = S2;
                    if (ph == 0) {
                        epi = REC_PEEP_A;
                        s.A[1] = hprev; s.K[1] = first ? 0 : U_pad;
                        s.A[2] = Cprev; s.K[2] = first ? 0 : U_pad;
                        s.W = Ly.WA.as<float>(); s.Ws = Ly.WAs.as<float>(); s.ldw = Ly.Kx + 2 * U_pad; s.bias = Ly.bA_first.as<float>();
                        s.E0 = Cprev; s.O0 = Cnext; s.O1 = Opre;
                    } else {
                        epi = REC_PEEP_B;
                        s.A[0] = Cnext; s.K[0] = U_pad;
                        s.W = Ly.WB.as<float>(); s.Ws = Ly.WBs.as<float>(); s.ldw = U_pad; s.bias = Ly.bB_first.as<float>();
                        s.E0 = Cnext; s.E1 = Opre; s.O0 = hout;
                    }
                }
            }
            AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_rec_step(e->stream, epi, w); }));
        }
    return AM_OK;
}

// The same stack as ONE persistent launch (rec_stream_kernel): roles = phases of layers, planes time-major and written once.
static int rec_stack_stream(am_pipeline* p, int mi, size_t h_stride)
{
    am_engine* e = p->e;
    am_model* m = p->members[mi];
    auto& pl = p->persist[mi];
    RecStreamArgs ra = pl.rsa;
    const int NL = m->n_layers, U_pad = ra.U_pad, nbu = U_pad / LSTM_KB;
    const int n_phase = m->cell == AM_CELL_MGRU ? 1 : 2;
    const int ncA = (m->cell == AM_CELL_PEEPHOLE ? 4 : 2) * U_pad / 64, ncB = U_pad / 64;
    float* Hall = p->d_H.as<float>();
    float* X = p->d_recplanes.as<float>();
    std::vector<RecStreamRole>& roles = pl.h_roles;
    roles.assign(ra.n_roles, RecStreamRole{});
    // Role numbering.  Whole phases: layer l -> n_phase l (+ 1 for phase B).  Split (pl.split; upper layers only): layer 0 as before, layer l >= 1 ->
    // n_phase + 2 n_phase (l - 1) + {input part of A, [input part of B,] state part of A [, state part of B]}.
    const bool split = pl.split;
    auto role_of = [&](int l, int phase, bool input_part) {
        if (!split || l == 0) return n_phase * l + phase;
        return n_phase + 2 * n_phase * (l - 1) + (input_part ? 0 : n_phase) + phase;
    };
    const size_t xplane_cols = (size_t)(ncA + (n_phase == 2 ? ncB : 0));   // 64-column workgroups of a layer's phases: the layer's share of d_xplanes
    RecStreamSeg none{}; none.dep = -1;
    int wg = 0;
    auto place = [&](RecStreamRole& R, int idx, int n_col) { R.n_col = n_col; R.wg0 = wg; R.counter = idx; wg += n_col; };
    for (int l = 0; l < NL; ++l) {
        Layer& Ly = m->L[l];
        float* H = Hall + l * h_stride;
        float* P1 = X + (size_t)(2 * l) * h_stride; float* P2 = X + (size_t)(2 * l + 1) * h_stride;
        const bool sp = split && l > 0;
        const int rA = role_of(l, 0, false), rB = role_of(l, 1, false);
        const int hw = n_phase == 1 ? rA : rB;                                    // the role that writes this layer's h
        const int below = l > 0 ? role_of(l - 1, n_phase - 1, false) : -1;        // ... and the layer below's
        const unsigned below_mult = (unsigned)(n_phase == 1 ? ncA : ncB), hw_mult = below_mult;
        const int nbx = Ly.Kx / LSTM_KB;
        float* xp = sp ? pl.d_xplanes.as<float>() + (size_t)2048 * p->total_tiles * xplane_cols * (size_t)(l - 1) : nullptr;   // this layer's accumulator planes: A, then B
        RecStreamSeg sx{};   // the cell's input: the gathered frames (layer 0) or h of the layer below at the same step
        sx.A = l == 0 ? p->d_in.as<float>() : Hall + (l - 1) * h_stride; sx.ld = l == 0 ? m->in_pad : U_pad; sx.nb = nbx;
        sx.prev = 0; sx.skip_first = 0; sx.dep = below; sx.dep_mult = below_mult; sx.wb0 = 0;
        RecStreamSeg sh{};   // h of this layer at the previous step
        sh.A = H; sh.ld = U_pad; sh.nb = nbu; sh.prev = 1; sh.skip_first = 1; sh.dep = hw; sh.dep_mult = hw_mult; sh.wb0 = nbx;
        // the input part of phase `ph` as a role of its own
        auto input_role = [&](int ph, const float* W, int ldw, int n_col, float* plane) {
            const int idx = role_of(l, ph, true);
            RecStreamRole& R = roles[idx];
            R = RecStreamRole{};
            R.seg[0] = sx; R.seg[1] = none; R.seg[2] = none;
            R.W = W; R.ldw = ldw; R.epi = REC_XPART; R.edep = -1; R.idep = -1; R.xplane = plane;
            place(R, idx, n_col);
        };
        if (sp) {
            input_role(0, Ly.WAs.as<float>(), Ly.Kx + U_pad, ncA, xp);
            if (n_phase == 2) input_role(1, Ly.WBs.as<float>(), Ly.Kx + U_pad, ncB, xp + (size_t)2048 * p->total_tiles * ncA);
        }
        RecStreamRole& A = roles[rA];
        A = RecStreamRole{};
        A.seg[0] = sp ? none : sx; A.seg[1] = none; A.seg[2] = none;
        A.edep = -1; A.idep = -1;
        if (sp) { A.init = xp; A.idep = role_of(l, 0, true); A.idep_mult = (unsigned)ncA; }
        place(A, rA, ncA);
        if (m->cell == AM_CELL_MGRU) {
            A.seg[1] = sh; A.W = Ly.WAs.as<float>(); A.ldw = Ly.Kx + U_pad;
            A.bias_first = Ly.bA_first.as<float>(); A.bias_rest = Ly.bA_rest.as<float>();
            A.E0 = H; A.O0 = H; A.epi = REC_MGRU;
            continue;
        }
        RecStreamRole& B = roles[rB];
        B = RecStreamRole{};
        B.seg[0] = sp ? none : sx; B.seg[1] = none; B.seg[2] = none;
        B.edep = -1; B.idep = -1;
        if (sp) { B.init = xp + (size_t)2048 * p->total_tiles * ncA; B.idep = role_of(l, 1, true); B.idep_mult = (unsigned)ncB; }
        place(B, rB, ncB);
        if (m->cell == AM_CELL_GRU) {
            A.seg[1] = sh; A.W = Ly.WAs.as<float>(); A.ldw = Ly.Kx + U_pad;
            A.bias_first = Ly.bA_first.as<float>(); A.bias_rest = Ly.bA_rest.as<float>();
            A.E0 = H; A.O0 = P1; A.O1 = P2; A.epi = REC_GRU_A;                    // Z, r.h
            RecStreamSeg srh{};
            srh.A = P2; srh.ld = U_pad; srh.nb = nbu; srh.prev = 0; srh.skip_first = 1; srh.dep = rA; srh.dep_mult = (unsigned)ncA; srh.wb0 = nbx;
            B.seg[1] = srh; B.W = Ly.WBs.as<float>(); B.ldw = Ly.Kx + U_pad;
            B.bias_first = Ly.bB_first.as<float>(); B.bias_rest = Ly.bB_rest.as<float>();
            B.E0 = H; B.E1 = P1; B.O0 = H; B.epi = REC_GRU_B; B.edep = rA; B.edep_mult = (unsigned)ncA;
        } else {   // AM_CELL_PEEPHOLE (never split)
            RecStreamSeg sc{};   // the cell state of the previous step
            sc.A = P1; sc.ld = U_pad; sc.nb = nbu; sc.prev = 1; sc.skip_first = 1; sc.dep = rA; sc.dep_mult = (unsigned)ncA; sc.wb0 = nbx + nbu;
            A.seg[1] = sh; A.seg[2] = sc; A.W = Ly.WAs.as<float>(); A.ldw = Ly.Kx + 2 * U_pad;
            A.bias_first = Ly.bA_first.as<float>(); A.bias_rest = Ly.bA_first.as<float>();
            A.E0 = P1; A.O0 = P1; A.O1 = P2; A.epi = REC_PEEP_A;                   // c', the o pre-activation
            RecStreamSeg scn{};  // c' of this step
            scn.A = P1; scn.ld = U_pad; scn.nb = nbu; scn.prev = 0; scn.skip_first = 0; scn.dep = rA; scn.dep_mult = (unsigned)ncA; scn.wb0 = 0;
            B.seg[0] = scn; B.W = Ly.WBs.as<float>(); B.ldw = U_pad;
            B.bias_first = Ly.bB_first.as<float>(); B.bias_rest = Ly.bB_first.as<float>();
            B.E0 = P1; B.E1 = P2; B.O0 = H; B.epi = REC_PEEP_B;
        }
    }
    HIP_TRY(hipMemcpyAsync(pl.d_roles.p, roles.data(), sizeof(RecStreamRole) * roles.size(), hipMemcpyHostToDevice, e->stream));   // (pageable source: staged before the call returns)
    ra.roles = pl.d_roles.as<RecStreamRole>();
    ra.n_active = p->d_nact.as<int>(); ra.step_off = p->d_stepoff.as<int>(); ra.tile_off = p->d_tileoff.as<int>();
    ra.counters = p->d_sync.as<unsigned>(); ra.abort_flag = p->d_sync.as<int>() + 63;
    HIP_TRY(hipMemsetAsync(p->d_sync.p, 0, sizeof(unsigned) * 63, e->stream));
    static const bool want_stamps = env_int("AM_LSTM_STAMPS", 0) != 0;
    ra.stamps = nullptr;
    if (want_stamps && mi == 0) {   // diagnostic timeline: one row of stamps per role (am_lstm_stamps)
        const size_t bytes = sizeof(unsigned long long) * (8 * (size_t)ra.n_roles * ra.T + 64 * (size_t)ra.n_roles);   // + one step's batch loop, AM_REC_DBG >> 8
        AM_TRY(p->d_stamps.ensure(bytes));
        HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, bytes, e->stream));
        ra.stamps = p->d_stamps.as<unsigned long long>();
        p->stamps_groups = ra.n_roles; p->stamps_T = ra.T;
    }
    PersistGate& G = gate_of(e);   // one persistent launch at a time per device: every workgroup must be resident
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.last && G.owner != e) HIP_TRY(hipStreamWaitEvent(e->stream, G.last, 0));
    AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_rec_stream(e->stream, ra); }));
    HIP_TRY(hipEventRecord(e->persist_done, e->stream));
    G.last = e->persist_done; G.owner = e;
    p->used_persist = true;
    return AM_OK;
}

static int lstm_enqueue(am_pipeline* p)
{
    am_engine* e = p->e;
    am_model* m0 = p->members[0];
    const int64_t NP = p->n_packed, rows_pad = round_up(NP, GEMM_BM);
    const int64_t T = (int64_t)p->n_active.size();
    p->acc_ld = m0->C_pad;
    if (p->timedelay > 0) {
        // rows the time loop never stores (the last `timedelay` frames of every utterance, predict_folds.py:50,61) keep
        // yb's initial zeros; in dev mode those zero rows then go through the final `y - logsum(y)` (predict_folds.py:219),
        // which turns an all-zero row into -log(C)
        const bool renorm = p->combine == AM_COMBINE_LOGPOST && p->normalise;
        const float fill = renorm ? -logf((float)m0->C) : 0.0f;
        int bits;
        memcpy(&bits, &fill, sizeof(bits));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p->d_out.p, bits, (size_t)p->n * m0->C, e->stream));
    }
    p->used_persist = false;
    int64_t split_rows = 0, split_step = 0;   // > 0: the output layer of rows [0, split_rows) runs on the second stream, beside the recurrence's tail
    if (p->d_sync.p) HIP_TRY(hipMemsetAsync(p->d_sync.p, 0, sizeof(unsigned) * 64, e->stream));
    GatherArgs ga{};
    ga.x = p->d_x.as<float>(); ga.ivec = p->has_ivec ? p->d_ivec.as<float>() : nullptr; ga.src = p->d_src.as<int>();
    ga.n = NP; ga.n_pad = rows_pad; ga.D = p->D;
    ga.add = p->has_ft ? p->d_add.as<float>() : nullptr; ga.scale = p->has_ft ? p->d_scale.as<float>() : nullptr;
    ga.d_iv = p->d_iv; ga.out = p->d_in.as<float>(); ga.ldo = m0->in_pad;
    AM_TRY(run_kernel(e, K_GATHER_FT, [&] { return launch_gather_ft(e->stream, ga); }));
    for (int mi = 0; mi < (int)p->members.size(); ++mi) {
        am_model* m = p->members[mi];
        const int NL = m->n_layers;
        const int U_pad = m->L[0].U_pad;
        const size_t h_stride = (size_t)(rows_pad + GEMM_BM) * U_pad;   // per-layer H plane
        const size_t c_stride = (size_t)round_up(std::max(p->n_slots, 1), 16) * U_pad;
        if (m->cell != AM_CELL_LSTM && p->persist[mi].form == 3) AM_TRY(rec_stack_stream(p, mi, h_stride));
        else if (m->cell != AM_CELL_LSTM) AM_TRY(rec_stack(p, m, h_stride));
        else {
        // layer 0: its input is known up front -> hoist the upward projection of ALL frames into one bulk GEMM
        {
            Layer& L0 = m->L[0];
            GemmArgs g{};
            g.A = p->d_in.as<float>(); g.lda = m->in_pad; g.W = L0.W.as<float>(); g.ldw = L0.in_pad; g.bias = L0.b.as<float>();
            g.C = p->d_G.as<float>(); g.ldc = L0.out_pad; g.M_pad = (int)rows_pad; g.N_pad = L0.out_pad;
            // the input plane is padded to 128 columns for the recurrent kernels' k batches; the columns past the real input are zeros on
            // both sides, so the product stops at the GEMM's own k step (40 inputs: 64 instead of 128 -- half the launch, the same bits)
            g.K_pad = std::min(L0.in_pad, (int)round_up(L0.in, GEMM_BK));
            g.epi = EPI_LINEAR; g.alpha = 1.f;
            g.tiled_units = L0.U;   // pad units' rows of G stay 0 whatever the input holds (inf x 0 would be NaN and spread through the lateral product)
            AM_TRY(run_kernel(e, K_LSTM_UPWARD, [&] { return launch_gemm(e->stream, g); }));
        }
        float* Hall = p->d_H.as<float>();
        float* Call = p->d_c.as<float>();
        if (p->persist[mi].form == 2) {
            // the whole stack and time axis in one persistent launch per layer segment, weights streamed (lstm_stream_kernel)
            float* Cplanes = p->d_cplane.as<float>();
            const size_t n_chunks = p->persist[mi].chunks.size();
            for (size_t ci = 0; ci < n_chunks; ++ci) {
                LstmPersistArgs pa = p->persist[mi].chunks[ci];
                const int lbase = p->persist[mi].layer0[ci];
                for (int l = lbase; l < lbase + pa.n_layers; ++l) {
                    Layer& Ly = m->L[l];
                    LstmPersistLayer& s = pa.L[l - lbase];
                    s.G = l == 0 ? p->d_G.as<float>() : nullptr;
                    s.bias = Ly.b.as<float>(); s.Wup = l == 0 ? nullptr : Ly.Ws.as<float>(); s.Wlat = Ly.Wlats.as<float>();   // (stream order)
                    s.Hin = l == 0 ? nullptr : Hall + (l - 1) * h_stride;
                    s.H = Hall + l * h_stride; s.Cst = Cplanes + l * h_stride;
                }
                pa.n_active = p->d_nact.as<int>(); pa.step_off = p->d_stepoff.as<int>();
                pa.counters = p->d_sync.as<unsigned>(); pa.abort_flag = p->d_sync.as<int>() + 63;
                static const bool want_stamps2 = env_int("AM_LSTM_STAMPS", 0) != 0;
                if (want_stamps2 && ci == 0) {   // diagnostic timeline: one row of stamps per layer (am_lstm_stamps)
                    const size_t bytes = sizeof(unsigned long long) * 8 * (size_t)pa.n_layers * pa.T;
                    AM_TRY(p->d_stamps.ensure(bytes));
                    HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, bytes, e->stream));
                    pa.stamps = p->d_stamps.as<unsigned long long>();
                    p->stamps_groups = pa.n_layers; p->stamps_T = pa.T;
                }
                HIP_TRY(hipMemsetAsync(p->d_sync.p, 0, sizeof(unsigned) * 63, e->stream));   // the abort flag (last word) is cleared once per pass
                PersistGate& G = gate_of(e);   // one persistent launch at a time per device: every workgroup must be resident
                std::lock_guard<std::mutex> lk(G.mu);
                if (G.last && G.owner != e) HIP_TRY(hipStreamWaitEvent(e->stream, G.last, 0));
                AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_lstm_stream(e->stream, pa); }));
                HIP_TRY(hipEventRecord(e->persist_done, e->stream));
                G.last = e->persist_done; G.owner = e;
            }
            p->used_persist = true;
        } else if (p->persist[mi].form == 1) {
            float* Cplanes = p->d_cplane.as<float>();
            // the whole stack and time axis in one persistent launch per chunk of utterance slots (lstm_persist_kernel)
            static const bool want_stamps = env_int("AM_LSTM_STAMPS", 0) != 0;
            static const bool overlap_ok = env_int("AM_LSTM_OVERLAP", 1) != 0;
            const size_t n_chunks = p->persist[mi].chunks.size();
            for (size_t ci = 0; ci < n_chunks; ++ci) {
                LstmPersistArgs pa = p->persist[mi].chunks[ci];
                const int lbase = p->persist[mi].layer0[ci];   // (launches run in the order planned: a segment starts when the one below it is complete)
                for (int l = lbase; l < lbase + pa.n_layers; ++l) {
                    Layer& Ly = m->L[l];
                    LstmPersistLayer& s = pa.L[l - lbase];
                    s.G = l == 0 ? p->d_G.as<float>() : nullptr;
                    s.bias = Ly.b.as<float>(); s.Wup = l == 0 ? nullptr : Ly.W.as<float>(); s.Wlat = Ly.Wlat.as<float>();
                    s.Hin = l == 0 ? nullptr : Hall + (l - 1) * h_stride;
                    s.H = Hall + l * h_stride; s.Cst = Cplanes + l * h_stride;
                }
                pa.n_active = p->d_nact.as<int>(); pa.step_off = p->d_stepoff.as<int>(); pa.cum = p->persist[mi].cums[ci].as<unsigned>();
                pa.counters = p->d_sync.as<unsigned>(); pa.abort_flag = p->d_sync.as<int>() + 63;
                if (want_stamps && ci == 0) {
                    const size_t bytes = sizeof(unsigned long long) * 8 * (size_t)pa.n_groups_total * pa.T;
                    AM_TRY(p->d_stamps.ensure(bytes));
                    HIP_TRY(hipMemsetAsync(p->d_stamps.p, 0, bytes, e->stream));
                    pa.stamps = p->d_stamps.as<unsigned long long>();
                    p->stamps_groups = pa.n_groups_total; p->stamps_T = pa.T;
                }
                HIP_TRY(hipMemsetAsync(p->d_sync.p, 0, sizeof(unsigned) * 63, e->stream));   // the abort flag (last word) is cleared once per pass
                // Overlap of the output layer with the recurrence's tail (single network, logit output, one chunk): from the step
                // t_A at which at most 32 utterances are still running, two thirds of the persistent kernel's workgroups have no
                // rows left and leave.  The output GEMM + log-softmax of the rows of steps < t_A go to the second stream behind a
                // one-wave gate kernel that returns when the top layer has finished step t_A - 1; they take the CUs as those are vacated.
                if (n_chunks == 1 && overlap_ok && !e->profile && p->members.size() == 1 && p->combine == AM_COMBINE_LOGITS) {
                    int64_t tA = -1;
                    for (int64_t t = 1; t < T; ++t) if (p->n_active[t] <= 32) { tA = t; break; }
                    const int64_t RA = tA > 0 ? p->step_off[tA] / GEMM_BM * GEMM_BM : 0;
                    if (RA >= NP / 4 && RA < rows_pad) { split_rows = RA; split_step = tA; }
                }
                if (split_rows > 0) {
                    if (!p->ev_fork) { HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming)); }
                    HIP_TRY(hipEventRecord(p->ev_fork, e->stream));            // the counters are zero from here on
                    HIP_TRY(hipStreamWaitEvent(e->stream2, p->ev_fork, 0));
                }
                {   // one persistent launch at a time per device (PersistGate): wait for the previous one, whoever launched it
                    PersistGate& G = gate_of(e);
                    std::lock_guard<std::mutex> lk(G.mu);
                    if (G.last && G.owner != e) HIP_TRY(hipStreamWaitEvent(e->stream, G.last, 0));
                    if (split_rows > 0) {
                        // The output layer of the finished rows runs beside the recurrence's tail.  The time axis is cut at the step from
                        // which <= 32 utterances run: the launch for the steps before it ends there -- an event the second stream waits
                        // for; nothing spins on the GPU, so nothing can sit on a CU the recurrence needs -- and the launch for the rest
                        // finds h, c and the arrival counters where the first one left them (~10 us for the boundary and the weights'
                        // reload, once per pass).  (Rounds 1-2 ran ONE launch and a one-wave gate kernel polling its counters in front
                        // of the GEMM: that wave had to find room beside a workgroup of the recurrence on some SIMD, which happened to
                        // work with this kernel's 2 x 248 registers and deadlocked into the 2-s bound with a kernel that fills the file.)
                        LstmPersistArgs head = pa, tail = pa;
                        head.t_begin = 0; head.t_end = (int)split_step;
                        tail.t_begin = (int)split_step; tail.t_end = pa.T;
                        AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_lstm_persist(e->stream, head); }));
                        HIP_TRY(hipEventRecord(p->ev_fork, e->stream));             // (re-recorded: the second stream's branch starts here)
                        HIP_TRY(hipStreamWaitEvent(e->stream2, p->ev_fork, 0));
                        AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_lstm_persist(e->stream, tail); }));
                    } else
                    AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_lstm_persist(e->stream, pa); }));
                    HIP_TRY(hipEventRecord(e->persist_done, e->stream));
                    G.last = e->persist_done; G.owner = e;
                }
            }
            p->used_persist = true;
        } else
        // wavefront over (layer, time): launch d runs layer l at t = d - l; chain length T + NL - 1 instead of T * NL
        for (int64_t d = 0; d < T + NL - 1; ++d) {
            LstmWavefrontArgs w{};
            w.U_pad = U_pad;
            for (int l = 0; l < NL; ++l) {
                const int64_t t = d - l;
                if (t < 0 || t >= T) continue;
                Layer& Ly = m->L[l];
                LstmLayerStep& s = w.L[w.n++];
                float* H = Hall + l * h_stride;
                if (l == 0) { s.G = p->d_G.as<float>() + p->step_off[t] * Ly.out_pad; s.Hin = nullptr; }
                else { s.G = nullptr; s.Hin = Hall + (l - 1) * h_stride + p->step_off[t] * U_pad; }
                s.bias = Ly.b.as<float>(); s.Wup = Ly.W.as<float>(); s.Kin_pad = Ly.in_pad;
                s.Hprev = t > 0 ? H + p->step_off[t - 1] * U_pad : H;
                s.Wlat = Ly.Wlat.as<float>(); s.Cst = Call + l * c_stride;
                s.Hout = H + p->step_off[t] * U_pad;
                s.n_active = p->n_active[t]; s.first = t == 0;
            }
            AM_TRY(run_kernel(e, K_LSTM_STEP, [&] { return launch_lstm_wavefront(e->stream, w); }));
        }
        }
        const float* A = p->d_H.as<float>() + (size_t)(NL - 1) * h_stride;
        const int lda = U_pad;
        Layer& Lo = m->L[m->n_layers];
        GemmArgs g{};
        g.A = A; g.lda = lda; g.W = Lo.W.as<float>(); g.ldw = Lo.in_pad; g.bias = Lo.b.as<float>();
        g.C = p->d_acc.as<float>(); g.ldc = m->C_pad; g.M_pad = (int)rows_pad; g.N_pad = Lo.out_pad; g.K_pad = Lo.in_pad;
        g.epi = epilogue_mode(p, mi, &g.alpha);
        if (split_rows > 0) {
            // rows of the finished steps: second stream (gate kernel in front); the rest: compute stream, after the recurrence
            GemmArgs ga = g;
            ga.M_pad = (int)split_rows;
            ga.queue = p->d_sync.as<int>() + 62;   // zeroed with the arrival counters, before the fork
            ga.n_cus = e->n_cus;
            AM_TRY(run_kernel(e, K_GEMM_OUT, [&] { return launch_gemm(e->stream2, ga); }, e->stream2));
            AM_TRY(final_finish(p, split_rows, p->d_out.as<float>(), p->d_dst.as<int>(), 0, e->stream2));
            HIP_TRY(hipEventRecord(p->ev_join, e->stream2));
            g.A += (size_t)split_rows * lda; g.C += (size_t)split_rows * m->C_pad; g.M_pad = (int)(rows_pad - split_rows);
            AM_TRY(run_kernel(e, K_GEMM_OUT, [&] { return launch_gemm(e->stream, g); }));
            AM_TRY(final_finish(p, NP - split_rows, p->d_out.as<float>(), p->d_dst.as<int>(), split_rows));
            HIP_TRY(hipStreamWaitEvent(e->stream, p->ev_join, 0));
            return AM_OK;
        }
        AM_TRY(run_kernel(e, K_GEMM_OUT, [&] { return launch_gemm(e->stream, g); }));
        if (p->combine == AM_COMBINE_LOGPOST) AM_TRY(member_finish(p, mi, NP));
    }
    return final_finish(p, NP, p->d_out.as<float>(), p->d_dst.as<int>());
}

extern "C" int am_lstm_download(am_pipeline* p, float* y)
{
    if (!p || p->mode != 2 || !y) return fail(AM_EINVAL, "am_lstm_download: nothing to download");
    if (!p->results_valid) return fail(AM_EINVAL, "am_lstm_download: the pipeline has not been run since its last upload / configuration change");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(join_streams(e));
    if (p->used_persist) { AM_TRY(sync_all(e)); AM_TRY(persist_check(p)); }   // (repeats the pass on the per-step kernels if the persistent one gave up)
    if (p->n > 0) HIP_TRY(hipMemcpyAsync(y, p->d_out.p, sizeof(float) * p->n * p->members[0]->C, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    return AM_OK;
}

extern "C" int am_lstm_stamps(am_pipeline* p, unsigned long long* out, int64_t capacity, int* n_groups, int* n_steps)
{
    if (!p || !n_groups || !n_steps) return fail(AM_EINVAL, "am_lstm_stamps: bad argument");
    *n_groups = p->stamps_groups; *n_steps = p->stamps_T;
    const int64_t need = (int64_t)8 * p->stamps_groups * p->stamps_T;
    if (!out || need == 0) return AM_OK;
    if (capacity < need) return fail(AM_EINVAL, "am_lstm_stamps: buffer too small");
    HIP_TRY(hipSetDevice(p->e->device));
    AM_TRY(sync_all(p->e));
    // (a caller that offers more room also gets what the launch wrote behind the table: the persistent cell kernel's trace of one step's batch loop)
    const int64_t have = (int64_t)(p->d_stamps.cap / sizeof(unsigned long long));
    HIP_TRY(hipMemcpy(out, p->d_stamps.p, sizeof(unsigned long long) * std::min(capacity, have), hipMemcpyDeviceToHost));
    return AM_OK;
}

extern "C" int am_forward_lstm(am_pipeline* p, const float* x, const float* ivec, const int64_t* offsets, int n_utts, int timedelay, float* y)
{
    AM_TRY(am_lstm_upload(p, x, ivec, offsets, n_utts, timedelay));
    AM_TRY(am_lstm_run(p));
    return am_lstm_download(p, y);
}

extern "C" int am_pipeline_synchronize(am_pipeline* p)
{
    if (!p) return fail(AM_EINVAL, "pipeline is NULL");
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(join_streams(e));
    AM_TRY(sync_all(e));
    if (!p->results_valid) return AM_OK;
    if (p->mode == 2 && p->used_persist) AM_TRY(persist_check(p));
    if (p->mode == 1 && p->used_stack) AM_TRY(ff_stack_check(p));
    return AM_OK;
}

extern "C" int am_pipeline_output_shape(am_pipeline* p, int64_t* n, int* c)
{
    if (!p || p->mode == 0 || !n || !c) return fail(AM_EINVAL, "am_pipeline_output_shape: nothing uploaded");
    *n = p->n; *c = p->members[0]->C;
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// stand-alone stage entry points
// ----------------------------------------------------------------------------------------------------
extern "C" int am_frontend_apply(am_pipeline* p, const float* x, const float* ivec, int64_t n, int halo_lo, int halo_hi, float* out)
{
    if (!p || p->D <= 0) return fail(AM_EINVAL, "am_frontend_apply: pipeline has no frontend");
    if (!x || !out || n < 0 || halo_lo < 0 || halo_hi < 0) return fail(AM_EINVAL, "am_frontend_apply: bad argument");
    if ((p->d_iv > 0) != (ivec != nullptr)) return fail(AM_EINVAL, "am_frontend_apply: i-vector argument does not match the frontend");
    if (n == 0) return AM_OK;
    am_engine* e = p->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    const int width = p->n_shifts * p->D + p->d_iv;
    const int ldo = (int)round_up(width, GEMM_BK);
    const int64_t ext = halo_lo + n + halo_hi;
    const int64_t n_pad = round_up(n, GEMM_BM);
    AM_TRY(e->scratch[0].ensure(sizeof(float) * ext * p->D));
    HIP_TRY(hipMemcpyAsync(e->scratch[0].p, x, sizeof(float) * ext * p->D, hipMemcpyHostToDevice, e->stream));
    if (ivec) {
        AM_TRY(e->scratch[1].ensure(sizeof(float) * n * p->d_iv));
        HIP_TRY(hipMemcpyAsync(e->scratch[1].p, ivec, sizeof(float) * n * p->d_iv, hipMemcpyHostToDevice, e->stream));
    }
    AM_TRY(e->scratch[2].ensure(sizeof(float) * n_pad * ldo));
    SpliceArgs s{};
    s.x = e->scratch[0].as<float>(); s.ext_rows = ext; s.first = halo_lo; s.ivec = ivec ? e->scratch[1].as<float>() : nullptr;
    s.row0 = 0; s.n = n; s.n_pad = n_pad; s.D = p->D; s.n_shifts = p->n_shifts; s.shifts = p->d_shifts.as<int>();
    s.add = p->has_ft ? p->d_add.as<float>() : nullptr; s.scale = p->has_ft ? p->d_scale.as<float>() : nullptr;
    s.d_iv = p->d_iv; s.out = e->scratch[2].as<float>(); s.ldo = ldo;
    AM_TRY(run_kernel(e, K_SPLICE_FT, [&] { return launch_splice_ft(e->stream, s); }));
    HIP_TRY(hipMemcpy2DAsync(out, sizeof(float) * width, e->scratch[2].p, sizeof(float) * ldo, sizeof(float) * width, n, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    return AM_OK;
}

extern "C" int am_lognormalise(am_engine* e, float* lp, int64_t n, int c, const float* ap)
{
    if (!e || !lp || n < 0 || c <= 0 || c > 2048) return fail(AM_EINVAL, "am_lognormalise: bad argument");
    if (n == 0) return AM_OK;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    AM_TRY(e->scratch[0].ensure(sizeof(float) * n * c));
    HIP_TRY(hipMemcpyAsync(e->scratch[0].p, lp, sizeof(float) * n * c, hipMemcpyHostToDevice, e->stream));
    if (ap) {
        AM_TRY(e->scratch[1].ensure(sizeof(float) * c));
        HIP_TRY(hipMemcpyAsync(e->scratch[1].p, ap, sizeof(float) * c, hipMemcpyHostToDevice, e->stream));
    }
    FinishArgs f{};
    f.in = e->scratch[0].as<float>(); f.ldin = c; f.out = e->scratch[0].as<float>(); f.ldout = c;
    f.n = n; f.C = c; f.prior = ap ? e->scratch[1].as<float>() : nullptr; f.normalise = 1;
    AM_TRY(run_kernel(e, K_FINISH, [&] { return launch_finish(e->stream, f); }));
    HIP_TRY(hipMemcpyAsync(lp, e->scratch[0].p, sizeof(float) * n * c, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    return AM_OK;
}

extern "C" int am_linear(am_engine* e, const float* x, int64_t n, int in_dim, const float* W, const float* b, int out_dim, int act, float* y)
{
    if (!e || !x || !W || !y || n < 0 || in_dim <= 0 || out_dim <= 0 || act > AM_ACT_TANH) return fail(AM_EINVAL, "am_linear: bad argument");
    if (n == 0) return AM_OK;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    const int kp = (int)round_up(in_dim, GEMM_BK), np = gemm_n_pad(out_dim);
    const int64_t mp = round_up(n, GEMM_BM);
    AM_TRY(e->scratch[0].ensure(sizeof(float) * mp * kp));
    AM_TRY(e->scratch[1].ensure(sizeof(float) * (size_t)np * kp));
    AM_TRY(e->scratch[2].ensure(sizeof(float) * np));
    AM_TRY(e->scratch[3].ensure(sizeof(float) * mp * np));
    HIP_TRY(hipMemsetAsync(e->scratch[0].p, 0, sizeof(float) * mp * kp, e->stream));
    HIP_TRY(hipMemsetAsync(e->scratch[1].p, 0, sizeof(float) * (size_t)np * kp, e->stream));
    HIP_TRY(hipMemsetAsync(e->scratch[2].p, 0, sizeof(float) * np, e->stream));
    HIP_TRY(hipMemcpy2DAsync(e->scratch[0].p, sizeof(float) * kp, x, sizeof(float) * in_dim, sizeof(float) * in_dim, n, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpy2DAsync(e->scratch[1].p, sizeof(float) * kp, W, sizeof(float) * in_dim, sizeof(float) * in_dim, out_dim, hipMemcpyHostToDevice, e->stream));
    if (b) HIP_TRY(hipMemcpyAsync(e->scratch[2].p, b, sizeof(float) * out_dim, hipMemcpyHostToDevice, e->stream));
    GemmArgs g{};
    g.A = e->scratch[0].as<float>(); g.lda = kp; g.W = e->scratch[1].as<float>(); g.ldw = kp; g.bias = e->scratch[2].as<float>();
    g.C = e->scratch[3].as<float>(); g.ldc = np; g.M_pad = (int)mp; g.N_pad = np; g.K_pad = kp;
    g.act = act < 0 ? 0 : act; g.epi = act < 0 ? EPI_LINEAR : EPI_ACT; g.alpha = 1.f;
    AM_TRY(run_kernel(e, K_GEMM_HIDDEN, [&] { return launch_gemm(e->stream, g); }));
    HIP_TRY(hipMemcpy2DAsync(y, sizeof(float) * out_dim, e->scratch[3].p, sizeof(float) * np, sizeof(float) * out_dim, n, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    return AM_OK;
}

// ----------------------------------------------------------------------------------------------------
// Viterbi phone decoder
// ----------------------------------------------------------------------------------------------------
struct am_decoder {
    am_engine* e = nullptr;
    int P = 0, B = 0, T = 0, S = 0;
    float PIP = 0.f;
    DevBuf endIdx, tri_idx, tri_self, tri_trans, outOff, outVal, lm, p0;
    DevBuf d_slot;       // frame-synchronous form: slot -> utterance
    DevBuf d_slot_row;   // ... and slot -> first row of that utterance
    DevBuf cmp0, cmp1;   // compact 32-B graph records (nodes and posterior columns < 65536), see viterbi_compact_kernel
    bool has_compact = false;
    int form = AM_VITERBI_AUTO;   // am_decoder_set_form
    int perm = 0, inv0 = 0;   // compact records stored in arrival-list order (see ViterbiArgs::perm)
    DevBuf d_lp, d_off, pp, tt, pout, p, p2, inv0_state, bestI, bestT, seg_begin, seg_end, seg_tri, seg_count;
    DevBuf blk_tri0, blk_node0;   // fused frame-synchronous form (viterbi_fs_fused_kernel): blocks of whole arrival lists, <= 2,048 triphones each
    int n_blk = 0, blk_nodes_max = 0;
};

template <class Tt>
static int upload_raw(am_engine* e, DevBuf& d, const Tt* src, size_t count)
{
    AM_TRY(d.ensure(std::max<size_t>(sizeof(Tt) * count, 16)));
    if (count) HIP_TRY(hipMemcpyAsync(d.p, src, sizeof(Tt) * count, hipMemcpyHostToDevice, e->stream));
    return AM_OK;
}

extern "C" int am_decoder_create(am_engine* e, int numPhones, int numBiphones, int numTriphones, int numStates,
                                 const int* startIdxs, const int* endIdxs, const int* biphoneIdxIn, const int* stateIdxs,
                                 const float* tpSelfloop, const float* tpTransition, const int* outOffsets, const int* outValues,
                                 const float* lm, float PIP, float LMW, am_decoder** out)
{
    if (!out) return fail(AM_EINVAL, "am_decoder_create: out is NULL");
    *out = nullptr;
    if (!e || numPhones <= 0 || numBiphones <= 0 || numTriphones <= 0 || numStates <= 0 || !startIdxs || !endIdxs || !biphoneIdxIn ||
        !stateIdxs || !tpSelfloop || !tpTransition || !outOffsets || !outValues || !lm)
        return fail(AM_EINVAL, "am_decoder_create: bad argument");
    if (e->closed) return fail(AM_EINVAL, "the engine has been destroyed");
    const int P = numPhones, B = numBiphones, T = numTriphones;
    // validate every index once on the host: the kernel then needs no bounds checks
    for (int i = 0; i < P; ++i)
        if (startIdxs[i] < 0 || startIdxs[i] >= B || endIdxs[i] < 0 || endIdxs[i] >= B) return fail(AM_EINVAL, "am_decoder_create: start/end node out of range");
    for (int n = 0; n < T; ++n) {
        if (biphoneIdxIn[n] < 0 || biphoneIdxIn[n] >= B) return fail(AM_EINVAL, "am_decoder_create: biphoneIdxIn[%d] out of range", n);
        for (int s3 = 0; s3 < 3; ++s3)
            if (stateIdxs[3 * n + s3] < 0 || stateIdxs[3 * n + s3] >= numStates) return fail(AM_EINVAL, "am_decoder_create: stateIdxs[%d] out of range", 3 * n + s3);
    }
    if (outOffsets[0] != 0) return fail(AM_EINVAL, "am_decoder_create: outOffsets[0] must be 0");
    for (int i = 0; i < B; ++i)
        if (outOffsets[i + 1] < outOffsets[i]) return fail(AM_EINVAL, "am_decoder_create: outOffsets must be non-decreasing");
    for (int i = 0; i < outOffsets[B]; ++i)
        if (outValues[i] < 0 || outValues[i] >= T) return fail(AM_EINVAL, "am_decoder_create: outValues[%d] out of range", i);
    HIP_TRY(hipSetDevice(e->device));
    am_decoder* d = new am_decoder();
    d->e = e; ++e->children; d->P = P; d->B = B; d->T = T; d->S = numStates; d->PIP = PIP;
    std::vector<float> lms(B), p0(B, -1e20f);
    for (int i = 0; i < B; ++i) lms[i] = lm[i] * LMW;                                     // phone_recog.cpp:112
    for (int i = 0; i < P; ++i) p0[startIdxs[i]] = (float)((double)lms[startIdxs[i]] - log((double)(float)P));  // phone_recog.cpp:148
    // pack the per-triphone constants into 16-B records: one coalesced load each instead of eleven strided ones
    std::vector<int> tidx(4 * (size_t)T);
    std::vector<float> tself(4 * (size_t)T, 0.f), ttrans(4 * (size_t)T, 0.f);
    for (int n = 0; n < T; ++n) {
        tidx[4 * n] = biphoneIdxIn[n];
        for (int s3 = 0; s3 < 3; ++s3) {
            tidx[4 * n + 1 + s3] = stateIdxs[3 * n + s3];
            tself[4 * n + s3] = tpSelfloop[3 * n + s3];
            ttrans[4 * n + s3] = tpTransition[3 * n + s3];
        }
    }
    int rc = AM_OK;
    if (B < 65536 && numStates < 65536) {
        // when every triphone arrives at exactly one node (the arrival lists are a permutation of the triphones, as in a
        // triphone loop), store everything in arrival-list order: the node maxima then read contiguous exit scores
        std::vector<int> order(T);
        bool is_perm = outOffsets[B] == T;
        if (is_perm) {
            std::vector<char> seen(T, 0);
            for (int s3 = 0; s3 < T && is_perm; ++s3) { if (seen[outValues[s3]]) is_perm = false; seen[outValues[s3]] = 1; }
        }
        for (int s3 = 0; s3 < T; ++s3) order[s3] = is_perm ? outValues[s3] : s3;
        d->perm = is_perm;
        for (int s3 = 0; s3 < T; ++s3) if (order[s3] == 0) d->inv0 = s3;
        std::vector<int> c0(4 * (size_t)T);
        std::vector<float> c1(4 * (size_t)T);
        for (int m2 = 0; m2 < T; ++m2) {
            const int n = order[m2];
            c0[4 * m2] = biphoneIdxIn[n] | (stateIdxs[3 * n] << 16);
            c0[4 * m2 + 1] = stateIdxs[3 * n + 1] | (stateIdxs[3 * n + 2] << 16);
            memcpy(&c0[4 * m2 + 2], &tpSelfloop[3 * n], sizeof(float));
            memcpy(&c0[4 * m2 + 3], &tpSelfloop[3 * n + 1], sizeof(float));
            c1[4 * m2] = tpSelfloop[3 * n + 2];
            c1[4 * m2 + 1] = tpTransition[3 * n]; c1[4 * m2 + 2] = tpTransition[3 * n + 1]; c1[4 * m2 + 3] = tpTransition[3 * n + 2];
        }
        if ((rc = upload_raw(e, d->cmp0, c0.data(), c0.size())) != AM_OK || (rc = upload_raw(e, d->cmp1, c1.data(), c1.size())) != AM_OK) {
            am_decoder_destroy(d);
            return rc;
        }
        d->has_compact = true;
        if (is_perm) {
            // blocks of the fused form: whole arrival lists, greedily packed up to 2,048 triphones (a longer list: no fused form)
            std::vector<int> bt{0}, bn{0};
            bool fits = true;
            int cur = 0;
            for (int i = 0; i < B && fits; ++i) {
                const int len = outOffsets[i + 1] - outOffsets[i];
                if (len > 2048) fits = false;
                if (cur + len > 2048) { bt.push_back(outOffsets[i]); bn.push_back(i); cur = 0; }
                cur += len;
            }
            bt.push_back(T); bn.push_back(B);
            if (fits && (rc = upload_raw(e, d->blk_tri0, bt.data(), bt.size())) == AM_OK && (rc = upload_raw(e, d->blk_node0, bn.data(), bn.size())) == AM_OK)
            { d->n_blk = (int)bt.size() - 1; for (size_t bi = 0; bi + 1 < bn.size(); ++bi) d->blk_nodes_max = std::max(d->blk_nodes_max, bn[bi + 1] - bn[bi]); }
            if (rc != AM_OK) { am_decoder_destroy(d); return rc; }
        }
    }
    if ((rc = upload_raw(e, d->endIdx, endIdxs, P)) == AM_OK && (rc = upload_raw(e, d->tri_idx, tidx.data(), tidx.size())) == AM_OK &&
        (rc = upload_raw(e, d->tri_self, tself.data(), tself.size())) == AM_OK && (rc = upload_raw(e, d->tri_trans, ttrans.data(), ttrans.size())) == AM_OK &&
        (rc = upload_raw(e, d->outOff, outOffsets, (size_t)B + 1)) == AM_OK &&
        (rc = upload_raw(e, d->outVal, outValues, (size_t)outOffsets[B])) == AM_OK && (rc = upload_raw(e, d->lm, lms.data(), B)) == AM_OK &&
        (rc = upload_raw(e, d->p0, p0.data(), B)) == AM_OK)
        rc = sync_all(e);
    if (rc != AM_OK) { am_decoder_destroy(d); return rc; }
    *out = d;
    return AM_OK;
}

extern "C" int am_decoder_set_form(am_decoder* d, int form)
{
    if (!d || form < AM_VITERBI_AUTO || form > AM_VITERBI_PER_UTT_WIDE) return fail(AM_EINVAL, "am_decoder_set_form: bad argument");
    d->form = form;
    return AM_OK;
}

extern "C" int am_decoder_destroy(am_decoder* d)
{
    if (!d) return AM_OK;
    (void)hipSetDevice(d->e->device);
    (void)hipStreamSynchronize(d->e->stream);
    for (DevBuf* b : {&d->d_slot, &d->d_slot_row, &d->cmp0, &d->cmp1, &d->endIdx, &d->tri_idx, &d->tri_self, &d->tri_trans, &d->outOff, &d->outVal, &d->lm, &d->p0, &d->d_lp, &d->d_off,
                      &d->pp, &d->tt, &d->pout, &d->p, &d->p2, &d->inv0_state, &d->blk_tri0, &d->blk_node0, &d->bestI, &d->bestT, &d->seg_begin, &d->seg_end, &d->seg_tri, &d->seg_count})
        b->release();
    am_engine* e = d->e;
    delete d;
    engine_child_gone(e);
    return AM_OK;
}

// decode posteriors already on the device (lp_dev [N, ld]); results to the host arrays, chronological per utterance
static int decode_device(am_decoder* d, const float* lp_dev, int64_t ld, const int64_t* offsets, int n_utts,
                         int* seg_begin, int* seg_end, int* seg_tri, int* seg_count)
{
    am_engine* e = d->e;
    const int64_t N = offsets[n_utts];
    for (int u = 0; u < n_utts; ++u)
        if (offsets[u + 1] < offsets[u]) return fail(AM_EINVAL, "am_decode: offsets must be non-decreasing");
    if (N > INT32_MAX) return fail(AM_EINVAL, "am_decode: too many frames");
    const size_t T4 = 4 * (size_t)d->T;
    std::vector<long long> off(offsets, offsets + n_utts + 1);
    AM_TRY(upload_raw(e, d->d_off, off.data(), off.size()));
    AM_TRY(d->pp.ensure(sizeof(float) * T4 * std::max(n_utts, 1)));
    AM_TRY(d->tt.ensure(sizeof(int) * T4 * std::max(n_utts, 1)));
    AM_TRY(d->pout.ensure(sizeof(float) * (size_t)d->T * std::max(n_utts, 1)));
    AM_TRY(d->p.ensure(sizeof(float) * (size_t)d->B * std::max(n_utts, 1)));
    AM_TRY(d->p2.ensure(sizeof(float) * (size_t)d->B * std::max(n_utts, 1)));
    AM_TRY(d->inv0_state.ensure(sizeof(int) * 16 * (size_t)std::max(n_utts, 1)));
    AM_TRY(d->bestI.ensure(std::max<size_t>(sizeof(int) * (size_t)N * d->B, 16)));
    AM_TRY(d->bestT.ensure(std::max<size_t>(sizeof(int) * (size_t)N * d->B, 16)));
    for (DevBuf* b : {&d->seg_begin, &d->seg_end, &d->seg_tri}) AM_TRY(b->ensure(std::max<size_t>(sizeof(int) * (size_t)N, 16)));
    AM_TRY(d->seg_count.ensure(sizeof(int) * std::max(n_utts, 1)));
    ViterbiArgs a{};
    a.P = d->P; a.B = d->B; a.T = d->T;
    a.endIdx = d->endIdx.as<int>();
    a.tri_idx = (decltype(a.tri_idx))d->tri_idx.p; a.tri_self = (decltype(a.tri_self))d->tri_self.p; a.tri_trans = (decltype(a.tri_trans))d->tri_trans.p;
    int64_t longest = 0;
    for (int u = 0; u < n_utts; ++u) longest = std::max(longest, offsets[u + 1] - offsets[u]);
    static const int64_t compact_env = env_int("AM_VITERBI_COMPACT", 1);
    const bool compact_fits = d->has_compact && longest < 65535;   // entry times travel as 16-bit (t + 1)
    const bool fs_fits = compact_fits && n_utts > 0 && n_utts <= 65535;   // slots ride on gridDim.y
    if (d->form == AM_VITERBI_FRAME_SYNC && !fs_fits && n_utts > 0) return fail(AM_EINVAL, "am_decode: the frame-synchronous form does not fit this graph / batch");
    if (d->form == AM_VITERBI_PER_UTT_COMPACT && !compact_fits) return fail(AM_EINVAL, "am_decode: the compact form does not fit this graph / batch");
    const bool compact_ok = d->form == AM_VITERBI_AUTO ? compact_env != 0 : d->form != AM_VITERBI_PER_UTT_WIDE;
    if (compact_fits && compact_ok) {
        a.compact = 1; a.perm = d->perm; a.inv0 = d->inv0;
        a.tri_idx = (decltype(a.tri_idx))d->cmp0.p; a.tri_self = (decltype(a.tri_self))d->cmp1.p;
    }
    a.outOff = d->outOff.as<int>(); a.outVal = d->outVal.as<int>();
    a.lm = d->lm.as<float>(); a.p0 = d->p0.as<float>(); a.PIP = d->PIP;
    a.lp = lp_dev; a.ldl = ld; a.offsets = d->d_off.as<long long>();
    a.p2 = d->p2.as<float>(); a.inv0_state = d->inv0_state.as<int>();
    a.blk_tri0 = d->blk_tri0.as<int>(); a.blk_node0 = d->blk_node0.as<int>(); a.n_blk = d->n_blk; a.blk_nodes_max = d->blk_nodes_max;
    a.pp = d->pp.as<float>(); a.tt = d->tt.as<int>(); a.pout = d->pout.as<float>(); a.p = d->p.as<float>(); a.bestI = d->bestI.as<int>(); a.bestT = d->bestT.as<int>();
    a.seg_begin = d->seg_begin.as<int>(); a.seg_end = d->seg_end.as<int>(); a.seg_tri = d->seg_tri.as<int>(); a.seg_count = d->seg_count.as<int>();
    static const int64_t fs_env = env_int("AM_VITERBI_FS", 1);
    const bool fs_ok = d->form == AM_VITERBI_AUTO ? fs_env != 0 : d->form == AM_VITERBI_FRAME_SYNC;
    // two events of the engine's own bracket the decode's launch sequence (am_decode_elapsed_ms: device time of the whole pass, launch
    // gaps included, without the cost of an event pair per launch); the caller's am_event_record slots are never touched
    for (hipEvent_t& ev : e->dec_ev) if (!ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipEventRecord(e->dec_ev[0], e->stream));
    const auto host_t0 = std::chrono::steady_clock::now();
    if (a.compact && fs_ok && fs_fits) {
        // frame-synchronous: slots = utterances by length (longest first); step k advances the first n_active slots
        std::vector<int> order(n_utts);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return offsets[x + 1] - offsets[x] > offsets[y + 1] - offsets[y]; });
        AM_TRY(upload_raw(e, d->d_slot, order.data(), order.size()));
        a.slot_utt = d->d_slot.as<int>();
        std::vector<long long> slot_row(n_utts);
        for (int sl = 0; sl < n_utts; ++sl) slot_row[sl] = offsets[order[sl]];
        AM_TRY(upload_raw(e, d->d_slot_row, slot_row.data(), slot_row.size()));
        a.slot_row0 = d->d_slot_row.as<long long>(); a.n_cus = e->n_cus;
        // (the launch sequence replayed as ONE hipGraph launch takes the same device time -- 78.7 ms both ways on the round-2 kernels: the host
        // enqueues the ~1,280 launches in 3 ms, the boundaries are not the cost)
        int rc = run_kernel(e, K_VITERBI, [&] { return launch_viterbi_fs_init(e->stream, a, n_utts); });
        int active = n_utts;
        struct SegEvents {                                    // diagnostic (AM_VITERBI_TIMING=2): device time per 64 frames
            std::vector<hipEvent_t> v;
            ~SegEvents() { for (hipEvent_t ev : v) (void)hipEventDestroy(ev); }   // also on the early returns below
            void push_back(hipEvent_t ev) { v.push_back(ev); }
            size_t size() const { return v.size(); }
            hipEvent_t operator[](size_t i) const { return v[i]; }
        } seg_ev;
        std::vector<int> seg_active;
        const bool seg_timing = env_int("AM_VITERBI_TIMING", 0) == 2;
        for (int64_t k = 0; k < longest && rc == AM_OK; ++k) {
            while (active > 0 && offsets[order[active - 1] + 1] - offsets[order[active - 1]] <= k) --active;
            if (seg_timing && k % 64 == 0) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); HIP_TRY(hipEventRecord(ev, e->stream)); seg_ev.push_back(ev); seg_active.push_back(active); }
            rc = run_kernel(e, K_VITERBI, [&] { return launch_viterbi_fs_step(e->stream, a, (int)k, active); });
        }
        if (seg_timing && rc == AM_OK) {
            hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); HIP_TRY(hipEventRecord(ev, e->stream)); seg_ev.push_back(ev);
            HIP_TRY(hipStreamSynchronize(e->stream));
            for (size_t i = 0; i + 1 < seg_ev.size(); ++i) {
                float ms = 0.0f; (void)hipEventElapsedTime(&ms, seg_ev[i], seg_ev[i + 1]);
                const int64_t nf = std::min<int64_t>(64, longest - (int64_t)i * 64);
                fprintf(stderr, "libamfwd: frames %4zu..: %3d active at the start, %.3f ms, %.1f us per frame\n", i * 64, seg_active[i], ms, ms * 1e3 / nf);
            }
        }
        if (rc == AM_OK) rc = run_kernel(e, K_VITERBI, [&] { return launch_viterbi_fs_trace(e->stream, a, n_utts); });
        AM_TRY(rc);
    } else
    AM_TRY(run_kernel(e, K_VITERBI, [&] { return launch_viterbi(e->stream, a, n_utts); }));
    HIP_TRY(hipEventRecord(e->dec_ev[1], e->stream));
    if (env_int("AM_VITERBI_TIMING", 0) != 0)   // diagnostic: how long the host took to enqueue the launch sequence
        fprintf(stderr, "libamfwd: decode launch sequence enqueued in %.3f ms (host)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
    std::vector<int> hb(std::max<int64_t>(N, 1)), he(std::max<int64_t>(N, 1)), ht(std::max<int64_t>(N, 1));
    if (N > 0) {
        HIP_TRY(hipMemcpyAsync(hb.data(), d->seg_begin.p, sizeof(int) * N, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(he.data(), d->seg_end.p, sizeof(int) * N, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(ht.data(), d->seg_tri.p, sizeof(int) * N, hipMemcpyDeviceToHost, e->stream));
    }
    if (n_utts > 0) HIP_TRY(hipMemcpyAsync(seg_count, d->seg_count.p, sizeof(int) * n_utts, hipMemcpyDeviceToHost, e->stream));
    AM_TRY(sync_all(e));
    for (int u = 0; u < n_utts; ++u) {   // the kernel stores each utterance's segments last-to-first
        const int64_t o = offsets[u];
        const int c = seg_count[u];
        for (int j = 0; j < c; ++j) {
            seg_begin[o + j] = hb[o + c - 1 - j];
            seg_end[o + j] = he[o + c - 1 - j];
            seg_tri[o + j] = ht[o + c - 1 - j];
        }
    }
    return AM_OK;
}

extern "C" int am_decode_host(am_decoder* d, const float* lp, const int64_t* offsets, int n_utts,
                              int* seg_begin, int* seg_end, int* seg_tri, int* seg_count)
{
    if (!d || !offsets || n_utts < 0 || !seg_begin || !seg_end || !seg_tri || !seg_count || offsets[0] != 0) return fail(AM_EINVAL, "am_decode_host: bad argument");
    const int64_t N = offsets[n_utts];
    if (N > 0 && !lp) return fail(AM_EINVAL, "am_decode_host: lp is NULL");
    am_engine* e = d->e;
    HIP_TRY(hipSetDevice(e->device));
    AM_TRY(sync_all(e));
    AM_TRY(d->d_lp.ensure(std::max<size_t>(sizeof(float) * (size_t)N * d->S, 16)));
    if (N > 0) HIP_TRY(hipMemcpyAsync(d->d_lp.p, lp, sizeof(float) * (size_t)N * d->S, hipMemcpyHostToDevice, e->stream));
    return decode_device(d, d->d_lp.as<float>(), d->S, offsets, n_utts, seg_begin, seg_end, seg_tri, seg_count);
}

extern "C" int am_decode_pipeline(am_decoder* d, am_pipeline* p, const int64_t* offsets, int n_utts,
                                  int* seg_begin, int* seg_end, int* seg_tri, int* seg_count)
{
    if (!d || !p || !offsets || n_utts < 0 || !seg_begin || !seg_end || !seg_tri || !seg_count || offsets[0] != 0) return fail(AM_EINVAL, "am_decode_pipeline: bad argument");
    if (p->e != d->e) return fail(AM_EINVAL, "am_decode_pipeline: decoder and pipeline live on different engines");
    if (p->mode == 0 || !p->results_valid) return fail(AM_EINVAL, "am_decode_pipeline: the pipeline holds no posteriors (run it first)");
    if (p->members[0]->C != d->S) return fail(AM_EINVAL, "Num states in log-probs file and model image differ!");
    if (offsets[n_utts] != p->n) return fail(AM_EINVAL, "am_decode_pipeline: offsets cover %lld frames, the pipeline holds %lld", (long long)offsets[n_utts], (long long)p->n);
    HIP_TRY(hipSetDevice(d->e->device));
    AM_TRY(join_streams(d->e));
    if (p->mode == 2 && p->used_persist) { AM_TRY(sync_all(d->e)); AM_TRY(persist_check(p)); }
    if (p->mode == 1 && p->used_stack) { AM_TRY(sync_all(d->e)); AM_TRY(ff_stack_check(p)); }
    return decode_device(d, p->d_out.as<float>(), d->S, offsets, n_utts, seg_begin, seg_end, seg_tri, seg_count);
}
