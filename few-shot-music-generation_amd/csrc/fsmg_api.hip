// C-ABI of libfsmg (include/fsmg.h): model handle, HBM layout, step orchestration.
// Host-side C++ only; every kernel lives in gemm.hip / lstm_step.hip / elementwise.hip.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fsmg.h"
#include "fsmg_kernels.h"

using namespace fsmg;

namespace {

thread_local std::string g_create_error;

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

constexpr int RING_CAP = 1024;
constexpr int64_t FLAT_ALIGN = 64;   // floats (256 B)

struct ParamDesc {
    std::string name;
    int64_t rows, cols;     // reference shape (cols == 1 for vectors)
    int kind;               // 0 embedding, 1 kernel, 2 bias, 3 softmax_w, 4 softmax_b
    int layer;
    int64_t off, count;     // placement inside a flat buffer (floats)
};

struct TimerClass {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace

struct fsmg_model {
    fsmg_config cfg{};
    int V = 0, V1 = 0, T = 0, E = 0, H = 0, L = 0, Ep = 0, Hp = 0, V1p = 0, G4 = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // ---- persistent state: flat fp32 buffers P (params), G (grads + tail), M, V (Adam)
    char* state = nullptr;
    bool own_state = false;
    int64_t n_flat = 0;
    float *P = nullptr, *G = nullptr, *M = nullptr, *Vv = nullptr;
    std::vector<ParamDesc> params;
    int64_t off_emb = 0, off_w = 0, off_d = 0;
    std::vector<int64_t> off_kx, off_kh, off_b;
    std::vector<int> in_dim;            // padded input width of each layer (Ep or Hp)

    // ---- small device scalars
    long long* d_step = nullptr;
    int* d_err = nullptr;
    float* d_ring = nullptr;
    float* d_gnorm = nullptr;
    float* d_eval = nullptr;            // per-episode eval NLLs
    int eval_cap = 0;

    // ---- activations (scratch), sized for Bcap sequences
    int Bcap = 0;
    char* scratch = nullptr;
    int* d_tok = nullptr; int *X = nullptr, *Y = nullptr;
    std::vector<float*> Z, Hs, Cs;
    float2* ce_part = nullptr; float* tgt_logit = nullptr; int ce_nparts = 0;
    float *dC = nullptr, *dH = nullptr, *logits = nullptr, *dlogits = nullptr, *lse = nullptr, *ce = nullptr, *dXemb = nullptr;
    double* partials = nullptr;
    int partials_cap = 0;
    std::vector<float*> HF;             // fragment-ordered h per layer: [T+1][ceil(B/16)*16][Hp]
    float* dzF = nullptr;               // fragment-ordered dz ping-pong: [2][ceil(B/16)*16][4Hp]
    float* dzF_all = nullptr;           // persistent backward chain, all-gather form (FSMG_BWD_RS=0): fragment-ordered dz of every time step
    int64_t dzfa_floats = 0;
    float* inbox = nullptr;             // persistent backward chain, reduce-scatter form: dh partial tiles [2][row tiles][P][P][64][4]
    int64_t inbox_floats = 0;
    bool bwd_rs = true;                 // FSMG_BWD_RS=0 selects the all-gather form
    int chain_spin_limit = 1 << 18;     // FSMG_CHAIN_SPIN_LIMIT (0 forces the timeout + fallback path: tests)
    bool persist_timed_out = false;     // set when a persistent kernel gave up (the handle has switched to per-step launches)
    bool persist_cfg = true;            // what the configuration asked for; `persist` is what is in force right now
    int fallback_steps = 200;           // FSMG_FALLBACK_STEPS: train steps on per-step launches after a time-out, then the persistent path is tried again
    int fallback_left = 0;
    long long* host_counters = nullptr; // host-mapped tallies written by k_step_increment: [0] steps skipped after a time-out, [1] after a token-range error
    long long* d_counters = nullptr;    // the same memory as the device sees it
    long long seen_timeouts = 0, seen_token_errors = 0, seen_peer_failures = 0;
    bool force_fwd_rt = false;          // FSMG_FWD_RT=1: take the all-row-tiles forward kernel wherever it applies (tests)
    bool persist_fwd = true, persist_bwd = true;   // FSMG_PERSIST_FWD / FSMG_PERSIST_BWD = 0: that direction launches per step
    bool persist = true;                // FSMG_PERSISTENT=0: one launch per time step instead of one persistent launch per chain chunk
    float* khf = nullptr;               // fragment-ordered recurrent weights: per layer fwd copy, bwd copy
    float* P_saved = nullptr;           // cfg-E: theta while the handle computes at the adapted theta'
    static constexpr int MAX_TABLES = 4;
    int* table[MAX_TABLES] = {};        // device-resident packed splits [n_songs][T] (fsmg_upload_table)
    int64_t table_rows[MAX_TABLES] = {};
    int* d_idx = nullptr; int idx_cap = 0;
    int* d_gather = nullptr; int64_t gather_cap = 0;     // episode rows gathered from a table for the MAML-style step (its two passes take device token buffers)
    // XCD-local recurrence (lstm_xcd.hip; hidden size 512): per layer the forward and backward register images of K_h,
    // the h hand-off buffer, the dh-partial inboxes and the per-launch ticket counters
    bool xcd = true;                    // FSMG_XCD=0: keep the column-split persistent kernels
    int xcd_max_rows = 128;             // FSMG_XCD_MAX_ROWS: largest sequence count that takes the XCD-local kernels
    int dp_split = 0;                   // FSMG_DP_SPLIT=1 / 2: fsmg_forward_backward replays TWO graphs (forward + projection gradients | BPTT + the rest) and
                                        // records bucket 0's readiness between them, so its all-reduce runs under the second one
    int pair_mode = 2;                  // hidden size 1024 (one copy of K_h per XCD pair): 0 = column-split kernels, 1 = pair kernel forward only
                                        // (6.4 against 7.0 us per step; the backward pair kernel ties with the column-split one), 2 = both directions
    int xcd_variant = -1;               // FSMG_XCD_VARIANT: XCD_* bits for both directions (-1: lstm_xcd_default_variant)
    float* khx = nullptr;
    bool xcd_bx3 = false;               // hidden 512: the XCD-local recurrence on the bf16 matrix pipe (k_lstm_*_xcd16); one format per handle
    float* HX = nullptr; int64_t hx_floats = 0;
    float* inboxX = nullptr; int64_t inboxx_floats = 0;
    int* d_inbox_dirty = nullptr;       // device word: != 0 -> the next BPTT pass refills the inboxes first (set at creation, when the scratch moves,
                                        // and by k_step_increment after a time-out; a completed pass leaves every word reset by its reader)
    int* tickets = nullptr;             // [TICKET_LAUNCHES][8]
    static constexpr int TICKET_LAUNCHES = 64;
    int ticket_next = 0;
    int64_t n_timeouts = 0, n_persist_launches = 0, n_xcd_launches = 0, n_step_launches = 0;   // fsmg_get_stats
    bool khf_dirty = true;              // host wrote parameters since the last repack
    float* slabs = nullptr;             // split-K partial outputs of the GEMMs on the main stream
    float* arena = nullptr;             // slabs of the GEMMs whose sums are deferred into one launch (gemm(..., defer)): bump-allocated per pass
    int64_t arena_cap = 0, arena_off = 0;
    bool warned_split = false;
    // occurrence table of the input ids of a train pass (k_token_prep -> k_embed_grad): [V1] first position, [V1] count
    int* tok_first = nullptr; int* tok_count = nullptr;
    bool tok_table_open = false;        // a train-pass token_prep has been issued whose embed_grad has not (a failed call): refill before the next use
    bool last_bwd_xcd = false;          // the backward pass in flight took the XCD-local BPTT kernels (they did the conditional inbox refill)
    // eager passes: a pass on the persistent recurrent kernels is ~25 launches, which the host issues in < 0.1 ms -- replaying it
    // from a hipGraph buys nothing (measured: 534 vs 532 episodes/s at cfg-B) and costs the token staging copies, because a
    // captured token_prep cannot take the caller's pointers.  FSMG_EAGER=0: graphs wherever fsmg_config.use_graph allows.
    bool eager = true, eager_call = false;
    const int* cur_sup = nullptr; const int* cur_qry = nullptr;     // what token_prep reads: the caller's device buffers (eager) or the staging buffer
    bool fills_late = false;            // FSMG_FILLS_LATE=1: dH's slab sum + the BPTT fills behind the dW GEMM instead of in front of it (A/B)
    bool fill_early = false;            // FSMG_FILL_EARLY=1: the forward hand-off fills in front of the zx GEMM instead of behind it (A/B)
    float* colsum_slabs = nullptr;
    float* slabs2 = nullptr;            // ... and of the GEMMs on the auxiliary stream
    float* colsum_slabs2 = nullptr;
    hipStream_t probe = nullptr;        // fsmg_debug_clock_begin / _end: the shader-clock probe's own stream
    unsigned long long* d_probe = nullptr;
    hipStream_t aux = nullptr;          // low-priority stream for the projection GEMMs that overlap the recurrence
    static constexpr int NCHUNK = 16;   // max time chunks of the overlap schedule
    std::vector<int> chunk_edges;       // explicit chunk boundaries (FSMG_CHUNK_STEPS), empty = uniform
    int nchunk = 8;                     // chunks in use with one launch per step (FSMG_NCHUNK)
    int nchunk_persist = 4;             // ... and with the persistent step kernels (swept at cfg-B: 3-4 chunks x 2 blocks/CU)
    int aux_blocks_persist = 2;
    bool aux_blocks_from_env = false;
    int aux_blocks_per_cu = 2;          // occupancy cap of the overlapped GEMMs (FSMG_AUX_BLOCKS); swept: 8 x 2 is best at cfg-B
    hipEvent_t ev_chunk[NCHUNK] = {};   // main -> aux (forward) / aux -> main (backward): chunk ready
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool merge_dk = true;               // dKx and dKh of a layer as one GEMM with a two-part A (FSMG_MERGE_DK=0: two GEMMs)
    hipEvent_t ev_bucket[2] = {};       // [0] softmax gradients final, [1] backward complete
    bool overlap = true;                // FSMG_OVERLAP=0 disables the two-stream schedule
    bool overlap_forced = false;        // FSMG_OVERLAP was set: no per-call decision
    bool ov_call = false;               // the decision for the call in progress (choose_schedule)
    // XCD-partitioned schedule (FSMG_XCD_OVERLAP=1; off by default: measured 385 against 388 episodes/s at cfg-B, DESIGN.md
    // section 4): the recurrence packs its rows on the first XCDs and work-queue GEMMs on the auxiliary stream take the XCDs
    // it leaves free
    int bx3 = 1;                        // FSMG_GEMM=f32 selects the fp32-MFMA GEMM, default: bf16-split (k_gemm_bx3)
    bool xov = false, xov_call = false;
    bool bucket0_recorded = false;      // backward() recorded ev_bucket[0] itself (two-stream / XCD-partitioned order)
    int xov_dw_split = 4;               // K split of dW under this schedule: an item must be short against the chain it runs beside
    int xov_tail = 0;                   // FSMG_XOV_TAIL: time steps whose projection rows are left to a chip-wide launch behind the chain (0: none)
    int xov_pub = 6;                    // FSMG_XOV_PUB: the forward chain publishes every this many steps (a 256-row tile is 5.7 steps of 45 rows)
    int xov_strikes = 0;                // time-outs of passes in the XCD-partitioned order: the second one parks the schedule for this handle
    bool xov_last = false;              // the pass in flight took the XCD-partitioned order
    int xov_parts = 3;                  // FSMG_XOV_PARTS: 1 = forward pair only, 2 = backward pair only, 3 = both
    int* xov_prog = nullptr;            // [T] progress counters of the forward chain (LstmFwdXcdArgs::progress), the projection's gate
    // forward projection / dW: [0..1] draw counters, [2] stop flag, [3] items, [4 ..] claim words (gemm_restricted)
    static constexpr int XOV_CTL = 8192;
    int* xov_ctl = nullptr;             // [2][XOV_CTL]
    int64_t slab_cap = 0;
    // whole-phase hipGraphs, keyed by the shape of the call; dropped when scratch moves
    std::map<std::string, hipGraphExec_t> graphs;
    struct LaunchCounts { int64_t xcd = 0, persist = 0, step = 0; };
    std::map<std::string, LaunchCounts> graph_counts;   // recurrent launches one replay of a graph stands for (fsmg_get_stats)
    // decode
    float* dec = nullptr;

#ifdef FSMG_PHASE_DEBUG
    hipEvent_t ph[8] = {}; bool ph_init = false; int ph_step = 0;      // per handle (was file scope: shared by all handles)
#endif
    // gradient exchange inside the library (fsmg_comm_*): RCCL communicator, its stream, the event the compute stream waits on
    void* comm = nullptr; bool own_comm = false; int world = 1, rank = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_comm = nullptr;
    int lastB = 0;
    bool have_grads = false;
    std::string err;
    bool timing = false;
    std::string timing_only;
    std::map<std::string, TimerClass> timers;
};

namespace {

int fail(fsmg_model* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCK(h, call)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, FSMG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

// roctx ranges for rocprofv3 --marker-trace timelines (FSMG_ROCTX=1): libroctx64 is looked up at run time, so the library has no
// link-time dependency on it and the ranges cost nothing when off
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        const char* e = std::getenv("FSMG_ROCTX");
        if (!e || e[0] == '0') return;
        void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) push = nullptr;
    }
};
inline Roctx& roctx() { static Roctx r; return r; }
struct ScopedRange {
    bool on;
    explicit ScopedRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~ScopedRange() { if (on) roctx().pop(); }
};

struct ScopedTimer {
    fsmg_model* h; TimerClass* tc = nullptr; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(fsmg_model* h_, const char* cls) : h(h_) {
        if (!h->timing) return;
        if (!h->timing_only.empty() && h->timing_only != cls) return;
        tc = &h->timers[cls];
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { tc = nullptr; return; }
        hipEventRecord(a, h->stream);
    }
    ~ScopedTimer() {
        if (!tc) return;
        hipEventRecord(b, h->stream);
        tc->pending.emplace_back(a, b);
    }
};

void drain_timers(fsmg_model* h) {
    hipStreamSynchronize(h->stream);
    for (auto& kv : h->timers) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches += 1;
            }
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        kv.second.pending.clear();
    }
}

// ------------------------------------------------------------------ layout
void compute_dims(const fsmg_config& c, fsmg_model* m) {
    m->V = c.input_size; m->V1 = c.input_size + 1; m->T = c.max_len; m->E = c.embedding_size;
    m->H = c.hidden_size; m->L = c.n_layers;
    m->Ep = (int)round_up(m->E, 16);
    // Padded hidden size: a multiple of 16 (MFMA tiles) -- or of 64 when only THAT admits the persistent recurrent kernels and costs
    // at most a third more columns.  The reference's own default, hidden_size 200 (src/config/lstm_baseline.yaml:17), pads to 208,
    // which none of the persistent kernels takes (13 k-groups do not divide over 4 waves): one launch per time step, 4.75 + 6.5 us;
    // at 256 the column-split persistent kernels run it (measured: bench.py --config ref-default).  Pad units
    // are exact zeros forever (section 3), so the padding never changes a result.  FSMG_HP_ALIGN=16 / 64 forces.
    {
        const int h16 = (int)round_up(m->H, 16), h64 = (int)round_up(m->H, 64);
        static const int force = std::getenv("FSMG_HP_ALIGN") ? std::atoi(std::getenv("FSMG_HP_ALIGN")) : 0;
        int hp = h16;
        if (force == 64) hp = h64;
        else if (force != 16 && !lstm_fwd_chain_supported(45, h16) && lstm_fwd_chain_supported(45, h64) && 3 * h64 <= 4 * h16 && lstm_xcd_max_rows(h16) == 0) hp = h64;
        m->Hp = hp;
    }
    m->V1p = (int)round_up(m->V1, 4);
    m->G4 = 4 * m->Hp;
}

// Flat order: embedding | per layer: Kx (in x 4Hp), Kh (Hp x 4Hp) [contiguous = padded `kernel`], bias | softmax_w | softmax_b
int64_t build_layout(fsmg_model* m) {
    int64_t off = 0;
    auto place = [&](int64_t count) { int64_t o = off; off = round_up(off + count, FLAT_ALIGN); return o; };
    m->params.clear(); m->off_kx.clear(); m->off_kh.clear(); m->off_b.clear(); m->in_dim.clear();
    m->off_emb = place((int64_t)m->V1 * m->Ep);
    m->params.push_back({"embedding", m->V1, m->E, 0, 0, m->off_emb, (int64_t)m->V1 * m->Ep});
    for (int l = 0; l < m->L; ++l) {
        const int in_p = l == 0 ? m->Ep : m->Hp, in_r = l == 0 ? m->E : m->H;
        m->in_dim.push_back(in_p);
        const int64_t kcount = (int64_t)(in_p + m->Hp) * m->G4;
        const int64_t ko = place(kcount);
        m->off_kx.push_back(ko);
        m->off_kh.push_back(ko + (int64_t)in_p * m->G4);
        m->params.push_back({"kernel_" + std::to_string(l), in_r + m->H, 4 * m->H, 1, l, ko, kcount});
        const int64_t bo = place(m->G4);
        m->off_b.push_back(bo);
        m->params.push_back({"bias_" + std::to_string(l), 4 * m->H, 1, 2, l, bo, m->G4});
    }
    m->off_w = place((int64_t)m->Hp * m->V1p);
    m->params.push_back({"softmax_w", m->H, m->V1, 3, 0, m->off_w, (int64_t)m->Hp * m->V1p});
    m->off_d = place(m->V1p);
    m->params.push_back({"softmax_b", m->V1, 1, 4, 0, m->off_d, m->V1p});
    return off;
}

int64_t state_bytes_for(int64_t n_flat) { return (4 * n_flat + FSMG_GRAD_TAIL) * (int64_t)sizeof(float); }

const ParamDesc* find_param(fsmg_model* h, const char* name) {
    for (auto& p : h->params) if (p.name == name) return &p;
    return nullptr;
}

// packed gate column of (unit u, gate gi)
inline int64_t pcol(int u, int gi) { return 16 * (int64_t)(u >> 2) + 4 * gi + (u & 3); }

// reference-layout host tensor -> internal padded segment (zero padded), and back
void pack_param(const fsmg_model* m, const ParamDesc& p, const float* ref, float* seg) {
    std::memset(seg, 0, sizeof(float) * p.count);
    const int H = m->H, G4 = m->G4;
    switch (p.kind) {
    case 0:
        for (int64_t r = 0; r < m->V1; ++r) std::memcpy(seg + r * m->Ep, ref + r * m->E, sizeof(float) * m->E);
        break;
    case 1: {
        const int in_r = p.layer == 0 ? m->E : m->H, in_p = m->in_dim[p.layer];
        for (int64_t r = 0; r < in_r + H; ++r) {
            const int64_t ir = r < in_r ? r : in_p + (r - in_r);
            for (int gi = 0; gi < 4; ++gi)
                for (int u = 0; u < H; ++u) seg[ir * G4 + pcol(u, gi)] = ref[r * 4 * H + (int64_t)gi * H + u];
        }
        break;
    }
    case 2:
        for (int gi = 0; gi < 4; ++gi)
            for (int u = 0; u < H; ++u) seg[pcol(u, gi)] = ref[(int64_t)gi * H + u];
        break;
    case 3:
        for (int64_t r = 0; r < H; ++r) std::memcpy(seg + r * m->V1p, ref + r * m->V1, sizeof(float) * m->V1);
        break;
    case 4:
        std::memcpy(seg, ref, sizeof(float) * m->V1);
        break;
    }
}

void unpack_param(const fsmg_model* m, const ParamDesc& p, const float* seg, float* ref) {
    const int H = m->H, G4 = m->G4;
    switch (p.kind) {
    case 0:
        for (int64_t r = 0; r < m->V1; ++r) std::memcpy(ref + r * m->E, seg + r * m->Ep, sizeof(float) * m->E);
        break;
    case 1: {
        const int in_r = p.layer == 0 ? m->E : m->H, in_p = m->in_dim[p.layer];
        for (int64_t r = 0; r < in_r + H; ++r) {
            const int64_t ir = r < in_r ? r : in_p + (r - in_r);
            for (int gi = 0; gi < 4; ++gi)
                for (int u = 0; u < H; ++u) ref[r * 4 * H + (int64_t)gi * H + u] = seg[ir * G4 + pcol(u, gi)];
        }
        break;
    }
    case 2:
        for (int gi = 0; gi < 4; ++gi)
            for (int u = 0; u < H; ++u) ref[(int64_t)gi * H + u] = seg[pcol(u, gi)];
        break;
    case 3:
        for (int64_t r = 0; r < H; ++r) std::memcpy(ref + r * m->V1, seg + r * m->V1p, sizeof(float) * m->V1);
        break;
    case 4:
        std::memcpy(ref, seg, sizeof(float) * m->V1);
        break;
    }
}

int upload_tensor(fsmg_model* h, float* flat, const char* name, const float* host, int64_t count) {
    const ParamDesc* p = find_param(h, name);
    if (!p) return fail(h, FSMG_ERR_NAME, std::string("unknown parameter '") + name + "'");
    if (count != p->rows * p->cols) return fail(h, FSMG_ERR_SIZE, std::string("size mismatch for '") + name + "'");
    std::vector<float> seg(p->count);
    pack_param(h, *p, host, seg.data());
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(flat + p->off, seg.data(), sizeof(float) * p->count, hipMemcpyHostToDevice));
    if (flat == h->P) h->khf_dirty = true;
    return FSMG_OK;
}

int download_tensor(fsmg_model* h, const float* flat, const char* name, float* host, int64_t count) {
    const ParamDesc* p = find_param(h, name);
    if (!p) return fail(h, FSMG_ERR_NAME, std::string("unknown parameter '") + name + "'");
    if (count != p->rows * p->cols) return fail(h, FSMG_ERR_SIZE, std::string("size mismatch for '") + name + "'");
    std::vector<float> seg(p->count);
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(seg.data(), flat + p->off, sizeof(float) * p->count, hipMemcpyDeviceToHost));
    unpack_param(h, *p, seg.data(), host);
    return FSMG_OK;
}


// ------------------------------------------------------------------ split-K policy
// A 128x128-tile GEMM with few output tiles leaves most of the 256 CUs (2 resident blocks each)
// idle; splitting K multiplies the block count.  Cost model: MFMA time at ~100 TF/s divided by the
// slot efficiency of tiles*S blocks over the resident-block slots, plus S slabs of C written and read back.
constexpr int MAX_SPLIT = 16;
int pick_split(int64_t M, int64_t N, int64_t K, int64_t slots = 0, bool bx3 = false, int tile_mn = 0) {
    static const int max_split_env = std::getenv("FSMG_MAX_SPLIT") ? std::max(1, std::atoi(std::getenv("FSMG_MAX_SPLIT"))) : MAX_SPLIT;   // debugging knob
    if (max_split_env <= 1) return 1;
    if (slots <= 0) slots = gemm_block_slots();
    if (bx3 && tile_mn == 0) slots = slots * 3 / 4;     // k_gemm_bx3: three resident blocks per CU where k_gemm has four
    const int64_t tm = tile_mn ? tile_mn : gemm_tile_m(), tn = tile_mn ? tile_mn : 128;   // tile_mn = 256: k_gemm_bx3h, `slots` as given
    const int64_t tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
    const double t_mfma = 2.0 * M * N * K / (bx3 ? 170e12 : 100e12);
    const double t_slab = 2.0 * M * N * 4.0 / 4e12;
    int best = 1; double best_t = 1e30;
    for (int S = 1; S <= MAX_SPLIT; ++S) {
        if (S > 1 && K / S < 256) break;
        const int64_t blocks = tiles * S;
        const double eff = (double)blocks / (double)(((blocks + slots - 1) / slots) * slots);
        const double t = t_mfma / eff + (S > 1 ? S * t_slab : 0.0);
        if (t < best_t - 1e-12) { best_t = t; best = S; }
    }
    return best;
}


void drop_graphs(fsmg_model* h);

// ------------------------------------------------------------------ scratch
int ensure_scratch(fsmg_model* h, int B) {
    if (B <= h->Bcap) return FSMG_OK;
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->aux) HIPCK(h, hipStreamSynchronize(h->aux));
    drop_graphs(h);
    if (h->scratch) { HIPCK(h, hipFree(h->scratch)); h->scratch = nullptr; }
    const int64_t T = h->T, Hp = h->Hp, G4 = h->G4, rows = T * (int64_t)B;
    int64_t off = 0;
    auto place = [&](int64_t bytes) { int64_t o = off; off = round_up(off + bytes, 256); return o; };
    const int64_t o_tok = place(4 * rows), o_x = place(4 * rows), o_y = place(4 * rows);
    std::vector<int64_t> o_z(h->L), o_h(h->L), o_c(h->L);
    for (int l = 0; l < h->L; ++l) {
        o_z[l] = place(4 * rows * G4);
        o_h[l] = place(4 * (T + 1) * B * Hp);
        o_c[l] = place(4 * (T + 1) * B * Hp);
    }
    const int64_t Bp16 = (B + 15) / 16 * 16;
    std::vector<int64_t> o_hf(h->L);
    for (int l = 0; l < h->L; ++l) o_hf[l] = place(4 * (T + 1) * Bp16 * Hp);
    const int64_t o_dzf = place(4 * 2 * Bp16 * G4);
    // hand-off buffers of the persistent BPTT kernels, sized for the LARGEST row count they take at this Hp (not for
    // B: a validation batch grows the scratch far beyond that, and training steps must keep their fast path)
    int rows_rs = 0, rows_ag = 0;
    for (int r = 16; r <= (int)Bp16; r += 16) {
        if (lstm_bwd_rs_supported(r, (int)Hp)) rows_rs = r;
        if (lstm_bwd_chain_supported(r, (int)Hp)) rows_ag = r;
    }
    const bool want_inbox = h->persist && h->bwd_rs && rows_rs > 0;
    const bool want_dzfa = h->persist && !want_inbox && rows_ag > 0;
    const int64_t n_dzfa = want_dzfa ? T * (int64_t)rows_ag * G4 : 0;
    const int64_t o_dzfa = place(want_dzfa ? 4 * n_dzfa : 256);
    const int64_t n_inbox = want_inbox ? lstm_bwd_rs_inbox_floats(rows_rs, (int)Hp) : 0;
    const int64_t o_inbox = place(want_inbox ? 4 * n_inbox : 256);
    // XCD-local kernels: sized for the largest row count they take (not for B, same reason)
    const int xrows = (h->persist && h->xcd && lstm_xcd_max_rows((int)Hp) > 0 && (Hp != 1024 || h->pair_mode >= 1)) ? std::min(h->xcd_max_rows, lstm_xcd_max_rows((int)Hp)) : 0;
    const int64_t n_hx = xrows ? lstm_xcd_hx_floats(xrows, (int)T, (int)Hp, h->xcd_bx3) : 0;
    const int64_t n_inx = (xrows && (Hp != 1024 || h->pair_mode >= 2)) ? lstm_xcd_inbox_floats(xrows, (int)Hp) : 0;
    const int64_t o_hx = place(xrows ? 4 * n_hx : 256), o_inx = place(n_inx ? 4 * n_inx : 256);
    const int64_t o_dc = place(4 * (int64_t)B * Hp), o_dh = place(4 * rows * Hp);
    const int64_t o_lg = place(4 * rows * h->V1p), o_dlg = place(4 * rows * h->V1p), o_lse = place(4 * rows), o_ce = place(4 * rows);
    const int64_t o_dx = place(4 * rows * h->Ep);
    const int nparts = 2 * ((h->V1p + 127) / 128);
    const int64_t o_cep = place(8 * rows * nparts), o_tl = place(4 * rows);
    h->partials_cap = sqnorm_blocks(h->n_flat) + sqnorm_blocks(rows * h->Ep) + 8;
    const int64_t o_part = place(8 * (int64_t)h->partials_cap);
    // split-K slabs: the largest S*M*N over the backward GEMMs of this shape, over every (kernel, slot count) gemm() may pick
    // -- the 128-tile kernels on 256 .. 1024 slots, the wave-specialised one (682) and the 256 x 256-tile one (256 slots)
    int64_t slab_need = 0, arena_need = 0;
    {
        auto worst = [&](int64_t M, int64_t N, int64_t K) {
            int64_t w = 0;
            for (int64_t slots : {(int64_t)256, (int64_t)512, (int64_t)682, (int64_t)768, (int64_t)gemm_block_slots()}) {
                for (bool bx : {false, true}) {
                    const int S = pick_split(M, N, K, slots, bx);
                    if (S > 1) w = std::max(w, (int64_t)S * M * N);
                }
            }
            const int Sh = pick_split(M, N, K, 256, true, 256);
            if (Sh > 1) w = std::max(w, (int64_t)Sh * M * N);
            return w;
        };
        auto need = [&](int64_t M, int64_t N, int64_t K) { slab_need = std::max(slab_need, worst(M, N, K)); };
        // the slab sums of these are deferred (OpBatch): each needs its own slabs until the batch is flushed
        auto keep = [&](int64_t M, int64_t N, int64_t K) { arena_need += round_up(worst(M, N, K), 64) + round_up((int64_t)MAX_SPLIT * N, 64); };
        need(rows, Hp, h->V1p); need(Hp, h->V1p, rows); need(Hp, G4, rows);
        need(h->Ep, G4, rows); need(h->Ep + Hp, G4, rows); need(2 * Hp, G4, rows); need(rows, h->Ep, G4); need(rows, Hp, G4); need(rows, G4, h->Ep); need(rows, G4, Hp);
        keep(rows, Hp, h->V1p); keep(Hp, h->V1p, rows);
        for (int l = 0; l < h->L; ++l) { keep(Hp, G4, rows); keep(h->in_dim[l], G4, rows); keep(h->in_dim[l] + Hp, G4, rows); keep(rows, h->in_dim[l], G4); }
    }
    // chunked dH GEMMs of the overlap schedule have their own (smaller) shapes
    for (int nc : {h->nchunk, h->nchunk_persist})
    for (int c = 0; c < nc; ++c) {
        const int64_t m = ((int64_t)(c + 1) * T / nc - (int64_t)c * T / nc) * B;
        for (int64_t slots : {(int64_t)256, (int64_t)512, (int64_t)768, (int64_t)gemm_block_slots()}) {
            for (bool bx : {false, true}) {
                const int S = pick_split(m, Hp, h->V1p, slots, bx);
                if (S > 1) slab_need = std::max(slab_need, (int64_t)S * m * Hp);
                const int S2 = pick_split(m, h->V1p, Hp, slots, bx);
                if (S2 > 1) slab_need = std::max(slab_need, (int64_t)S2 * m * h->V1p);
            }
        }
    }
    if (h->xov) slab_need = std::max(slab_need, (int64_t)std::min(h->xov_dw_split, MAX_SPLIT) * Hp * h->V1p);   // dW in short tiles
    const int64_t o_slab = place(4 * std::max<int64_t>(slab_need, 64));
    const int64_t o_cslab = place(4 * (int64_t)MAX_SPLIT * std::max<int64_t>(h->V1p, G4));
    const int64_t o_slab2 = place(4 * std::max<int64_t>(slab_need, 64));
    const int64_t o_cslab2 = place(4 * (int64_t)MAX_SPLIT * std::max<int64_t>(h->V1p, G4));
    const int64_t o_arena = place(4 * std::max<int64_t>(arena_need, 64));
    hipError_t e = hipMalloc((void**)&h->scratch, off);
    if (e != hipSuccess) {
        h->Bcap = 0;
        return fail(h, FSMG_ERR_NOMEM, "hipMalloc of " + std::to_string(off) + " activation bytes failed: " +
                                           hipGetErrorString(e));
    }
    char* s = h->scratch;
    h->d_tok = (int*)(s + o_tok); h->X = (int*)(s + o_x); h->Y = (int*)(s + o_y);
    h->Z.assign(h->L, nullptr); h->Hs.assign(h->L, nullptr); h->Cs.assign(h->L, nullptr);
    for (int l = 0; l < h->L; ++l) {
        h->Z[l] = (float*)(s + o_z[l]); h->Hs[l] = (float*)(s + o_h[l]); h->Cs[l] = (float*)(s + o_c[l]);
    }
    h->HF.assign(h->L, nullptr);
    for (int l = 0; l < h->L; ++l) h->HF[l] = (float*)(s + o_hf[l]);
    h->dzF = (float*)(s + o_dzf);
    h->dzF_all = want_dzfa ? (float*)(s + o_dzfa) : nullptr; h->dzfa_floats = n_dzfa;
    h->inbox = want_inbox ? (float*)(s + o_inbox) : nullptr; h->inbox_floats = n_inbox;
    h->HX = xrows ? (float*)(s + o_hx) : nullptr; h->hx_floats = n_hx;
    h->inboxX = n_inx ? (float*)(s + o_inx) : nullptr; h->inboxx_floats = n_inx;
    if (h->d_inbox_dirty) { static const int one = 1; HIPCK(h, hipMemcpy(h->d_inbox_dirty, &one, sizeof(int), hipMemcpyHostToDevice)); }
    // pad rows of the fragment buffers are never written: clear once so they hold finite values
    HIPCK(h, hipMemsetAsync(s + o_hf[0], 0, (size_t)(o_dc - o_hf[0]), h->stream));
    h->dC = (float*)(s + o_dc); h->dH = (float*)(s + o_dh); h->logits = (float*)(s + o_lg);
    h->dlogits = (float*)(s + o_dlg);
    h->lse = (float*)(s + o_lse); h->ce = (float*)(s + o_ce); h->dXemb = (float*)(s + o_dx);
    h->partials = (double*)(s + o_part);
    h->ce_part = (float2*)(s + o_cep); h->tgt_logit = (float*)(s + o_tl); h->ce_nparts = nparts;
    h->slabs = (float*)(s + o_slab); h->colsum_slabs = (float*)(s + o_cslab); h->slab_cap = slab_need;
    h->slabs2 = (float*)(s + o_slab2); h->colsum_slabs2 = (float*)(s + o_cslab2);
    h->arena = (float*)(s + o_arena); h->arena_cap = arena_need; h->arena_off = 0;
    h->Bcap = B;
    return FSMG_OK;
}


void drop_graphs(fsmg_model* h) {
    for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    h->graph_counts.clear();
}

// A stream plus the split-K slab buffers its GEMMs may use.
struct Lane { hipStream_t s; float* slabs; float* colsum_slabs; int lds_pad; int slots; };
inline Lane main_lane(fsmg_model* h) { return Lane{h->stream, h->slabs, h->colsum_slabs, 0, gemm_block_slots()}; }
// forward-only passes (validation: many rows per step, patch step kernel) tolerate one more overlapped GEMM block
// per CU than training steps do (measured at cfg-B: eval 2862 vs 2690 episodes/s, train 290 vs 303)
inline Lane aux_lane(fsmg_model* h, bool forward_only = false, bool persistent_chain = false) {
    const int cap = std::min(4, (persistent_chain ? h->aux_blocks_persist : h->aux_blocks_per_cu) + (forward_only && !h->aux_blocks_from_env ? 1 : 0));
    return Lane{h->aux, h->slabs2, h->colsum_slabs2, gemm_lds_pad_for(cap), 256 * cap};
}

// C (contiguous, ldc == N) = op(A) * op(B) with the K range split over pick_split() slabs that are
// summed in a fixed order (deterministic); colsum likewise.
// Which bf16-split kernel: the wave-specialised k_gemm_bx3w (same bits as k_gemm_bx3 for the same K split; two 512-thread
// blocks per CU) pays where blocks are short-lived or few -- the projection (K = hidden size: 32 k tiles per block, +5-9 %)
// and the weight-gradient GEMMs whose M x N is only a few dozen tiles (dKh, dKx, dx: +15-20 %, a block alone on its CU
// needs 1700 cycles per k tile instead of 2470) -- and is a wash on the two large contractions over the vocabulary / the
// rows (tools/gemm_bench BX3=1 vs 2, profiles/r03_gemm_prof*.log).  FSMG_GEMM_WS=0 / 2: never / always (A/B runs).
bool use_ws_gemm(fsmg_model* h, int amode, int bmode, const GemmArgs& g, const Lane& ln) {
    static const int mode = std::getenv("FSMG_GEMM_WS") ? std::atoi(std::getenv("FSMG_GEMM_WS")) : 1;
    if (!h->bx3 || mode == 0 || ln.lds_pad != 0 || g.xcd_first != 0) return false;
    if (mode == 2) return true;
    // measured in the cfg-B step (profiles/r03b_bench_ws*.json, ms per launch without / with): projection 0.328 / 0.312,
    // dW 0.388 / 0.365, dx 0.052 / 0.048; zx 0.046 / 0.055, dH 0.336 / 0.349, dKh + dKx 0.149 / 0.151
    // At hidden size 1024 (cfg-C, profiles/r03d_cfg-C_ws*.json) the same kernel LOSES on the projection (K = 1024: 0.145 ->
    // 0.162 ms) and on dKh / dW (M = 1024), so the rule is a table of the shapes it was measured to win on, not a model.
    const int64_t tiles = ((g.M + 127) / 128) * (int64_t)((g.N + 127) / 128);
    if (amode == OP_KC && bmode == OP_XC) return g.K >= 384 && g.K <= 640 && tiles >= 512;  // projection at hidden 512
    if (amode == OP_XC && bmode == OP_XC) return tiles >= 256 && g.M <= 512;                // dW at hidden 512
    return g.K <= 4096;                                                                     // KC x KC: dx yes, dH no
}

// The 256 x 256-tile kernel k_gemm_bx3h (one 8-wave block per CU; half the loads, split work and fragment reads per MFMA; same
// bits for the same K split): where the output has enough 256-tiles x K slabs to fill the 256 CUs about once.
// FSMG_GEMM_H=0 / 2: never / wherever it can run (A/B runs).
bool use_h_gemm(fsmg_model* h, int amode, int bmode, const GemmArgs& g, const Lane& ln) {
    static const int mode = std::getenv("FSMG_GEMM_H") ? std::atoi(std::getenv("FSMG_GEMM_H")) : 1;
    if (!h->bx3 || mode == 0 || ln.lds_pad != 0 || g.xcd_first != 0) return false;
    if (amode == OP_XC && g.gather != nullptr && g.m_split == 0) return false;
    if (mode == 2) return true;
    // measured in the cfg-B step (profiles/r03p_bench_h*.json, ms per launch incl. the slab sums, without / with): dH 0.349 /
    // 0.296, dW 0.375 / 0.316, projection 0.318 / 0.310, dKh + dKx 0.154 / 0.147; zx 0.047 / 0.051, dx 0.050 / 0.057
    const int64_t tiles = ((g.M + 255) / 256) * (int64_t)((g.N + 255) / 256);
    if (amode == OP_KC && bmode == OP_KC) return g.K >= 4096 && tiles >= 32;               // dH, not dx
    if (amode == OP_XC && bmode == OP_XC) {                                                 // dW, dKh / the merged dKx + dKh
        // the merged form replaces TWO 128-tile launches: it pays from fewer rows on (cfg-E, hidden 1024, 1000-1250 rows per pass:
        // 260.6 -> 268.4 episodes/s with both layers merged, profiles/r04_merged_dk_ab.txt)
        if (g.m_split > 0 && tiles >= 64) return g.K >= 896;
        return g.K >= 2048 && tiles >= 16;
    }
    return g.K >= 384 && tiles >= 512;                                                      // the projection, not zx
}

// defer != nullptr: a split-K GEMM writes its slabs into the handle's slab ARENA (bump-allocated, reset per backward pass) and
// leaves their sum as REDUCE ops in *defer instead of launching it -- the caller flushes the batch before the first reader of C
// (the five slab sums of a cfg-B backward pass were five launches of 57 us; two now).  sq: squared-norm partials of C
// (sqnorm_blocks(M * N) doubles) as a by-product of the sum; *sq_done tells the caller whether that happened.
struct OpBatch;
int gemm(fsmg_model* h, const Lane& ln, int amode, int bmode, GemmArgs g, OpBatch* defer = nullptr, double* sq = nullptr, bool* sq_done = nullptr);
#define GEMMCK(call) do { int rc_ = (call); if (rc_ != FSMG_OK) return rc_; } while (0)

// collects the small memory passes a phase needs -- pattern fills and split-K slab sums -- and issues them as one launch (flush)
// right before the first kernel that depends on them
struct OpBatch {
    MultiOps r{};
    fsmg_model* h;
    explicit OpBatch(fsmg_model* h_) : h(h_) { r.count = 0; }
    int room(int n) { return (r.count + n > MULTI_MAX_OPS) ? flush() : FSMG_OK; }
    int add(void* p, uint32_t word, long long n_words, const int* cond = nullptr) {     // cond: fill only when *cond != 0 on the device
        if (n_words <= 0) return FSMG_OK;
        const int rc = room(1); if (rc != FSMG_OK) return rc;
        MultiOp& o = r.op[r.count++];
        o = MultiOp{}; o.kind = MULTI_FILL; o.dst = p; o.word = word; o.n = n_words; o.cond = cond;
        return FSMG_OK;
    }
    // out[i] = sum over the nslab slabs (fixed order); sq: squared-norm partials of out as a by-product
    int reduce(const float* slabs, long long stride, int nslab, float* out, long long n, double* sq = nullptr) {
        if (n <= 0) return FSMG_OK;
        const int rc = room(1); if (rc != FSMG_OK) return rc;
        MultiOp& o = r.op[r.count++];
        o = MultiOp{}; o.kind = MULTI_REDUCE; o.dst = out; o.src = slabs; o.stride = stride; o.nslab = nslab; o.n = n; o.sq = sq;
        return FSMG_OK;
    }
    int mean(const float* x, long long n, float* out) {       // *out = sum(x) / (n + 1e-12): one block of the launch
        if (n <= 0) return FSMG_OK;
        const int rc = room(1); if (rc != FSMG_OK) return rc;
        MultiOp& o = r.op[r.count++];
        o = MultiOp{}; o.kind = MULTI_MEAN; o.dst = out; o.src = x; o.n = n;
        return FSMG_OK;
    }
    int flush() {
        if (r.count == 0) return FSMG_OK;
        HIPCK(h, launch_multi_op(h->stream, r));
        r.count = 0;
        return FSMG_OK;
    }
};
typedef OpBatch FillBatch;


// Run `body` (a pure sequence of stream-ordered launches with call-invariant arguments) through a
// cached hipGraph: captured on first use for this key, replayed afterwards.  The ~300 launches of a
// step (one per time step and direction) then cost one hipGraphLaunch on the host.  Event timing
// needs eager launches, so graphs are bypassed while it is on.
template <class F>
int run_graphed(fsmg_model* h, const std::string& key, F&& body) {
    // hipGraph (ROCm 7.2) runs captured cross-stream branches one after the other, so the two-stream
    // schedule only overlaps with eager launches
    if (!h->cfg.use_graph || h->timing || h->ov_call || h->xov_call || h->eager_call) return body();
    auto it = h->graphs.find(key);
    if (it != h->graphs.end()) {           // a replay launches what the capture launched
        const auto& c = h->graph_counts[key];
        h->n_xcd_launches += c.xcd; h->n_persist_launches += c.persist; h->n_step_launches += c.step;
    }
    if (it == h->graphs.end()) {
        hipGraph_t graph = nullptr;
        const int64_t x0 = h->n_xcd_launches, p0 = h->n_persist_launches, s0 = h->n_step_launches;
        HIPCK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        const int rc = body();
        { auto& c = h->graph_counts[key]; c.xcd = h->n_xcd_launches - x0; c.persist = h->n_persist_launches - p0; c.step = h->n_step_launches - s0; }
        const hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (rc != FSMG_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || graph == nullptr)
            return fail(h, FSMG_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ei != hipSuccess) return fail(h, FSMG_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
        it = h->graphs.emplace(key, exec).first;
    }
    HIPCK(h, hipGraphLaunch(it->second, h->stream));
    return FSMG_OK;
}

// ------------------------------------------------------------------ the step pieces
// host-side parameter writes (init / set_param / restore) leave the fragment-ordered weight copies stale
int repack_recurrent_weights(fsmg_model* h, hipStream_t s);
int ensure_khf(fsmg_model* h) {
    if (!h->khf_dirty) return FSMG_OK;
    const int rc = repack_recurrent_weights(h, h->stream);
    if (rc != FSMG_OK) return rc;
    h->khf_dirty = false;
    return FSMG_OK;
}

// tokens -> the handle's fixed staging buffer (H2D or D2D), so that every later launch has
// call-invariant arguments and can live in a replayed graph
int stage_tokens(fsmg_model* h, const int32_t* support, int n_sup, const int32_t* query, int n_qry, int on_device) {
    const size_t T = h->T;
    if (on_device && h->eager_call) {       // an eager pass reads the caller's device buffers in place: no copies, nothing between two steps
        h->cur_sup = support; h->cur_qry = query;
        return FSMG_OK;
    }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (n_sup > 0) HIPCK(h, hipMemcpyAsync(h->d_tok, support, sizeof(int) * n_sup * T, kind, h->stream));
    if (n_qry > 0) HIPCK(h, hipMemcpyAsync(h->d_tok + n_sup * T, query, sizeof(int) * n_qry * T, kind, h->stream));
    h->cur_sup = h->d_tok; h->cur_qry = h->d_tok + n_sup * T;
    return FSMG_OK;
}

// (INT_MAX, 0) in every entry of the occurrence table
int reset_tok_table(fsmg_model* h) {
    if (!h->tok_first) return FSMG_OK;
    OpBatch ops(h);
    GEMMCK(ops.add(h->tok_first, 0x7FFFFFFFu, h->V1));
    GEMMCK(ops.add(h->tok_count, 0u, h->V1));
    return ops.flush();
}

// train: the pass ends in k_embed_grad, which wants the occurrence table of the input ids
int token_prep(fsmg_model* h, int n_sup, int n_qry, bool train = false) {
    const bool table = train && h->tok_first != nullptr;
    HIPCK(h, launch_token_prep(h->stream, h->cur_sup, n_sup, h->cur_qry, n_qry, h->T, h->V, h->V,
                               h->X, h->Y, h->d_err, table ? h->tok_first : nullptr, table ? h->tok_count : nullptr));
    return FSMG_OK;
}

// Two-stream schedule.  The recurrent chains are latency bound (one small kernel per time step), the
// vocabulary-projection GEMMs are throughput bound, and per time chunk they are independent:
//   forward : logits + cross entropy of chunk c need h_t only for t in chunk c
//   backward: the BPTT steps of chunk c need dH only for t in chunk c; dW needs no BPTT result at all
// so the projection work runs on a low-priority auxiliary stream, forked / joined with events (inside
// the captured graph these become parallel branches).  Event timing (eager, one class at a time)
// and FSMG_OVERLAP=0 use the single-stream order.
// time-chunk boundaries of the overlap schedule: uniform, or the explicit step counts of FSMG_CHUNK_STEPS ("12,36,34,34,12")
inline int chunk_begin(const fsmg_model* h, int c, int nch) {
    if (!h->chunk_edges.empty() && (int)h->chunk_edges.size() == nch + 1) return h->chunk_edges[c];
    return (int)((int64_t)c * h->T / nch);
}
inline bool use_overlap(const fsmg_model* h) { return h->ov_call && !h->timing && h->aux != nullptr && h->T >= std::max(h->nchunk, h->nchunk_persist); }

#ifdef FSMG_PHASE_DEBUG
// compile-time debugging aid (make EXTRA=-DFSMG_PHASE_DEBUG): GPU time of the phases of the eager overlap
// schedule, from events on the main stream; printed every 20th step
static void phase_mark(fsmg_model* h, int i) {
    if (!h->ph_init) { for (auto& e : h->ph) hipEventCreate(&e); h->ph_init = true; }
    hipEventRecord(h->ph[i], h->stream);
}
static void phase_report(fsmg_model* h) {
    if (++h->ph_step % 20) return;
    hipStreamSynchronize(h->stream);
    const char* nm[] = {"zx+memsets", "fwd chain", "fwd join+loss", "to bwd chain", "bwd chain", "dk/dx/embed + dW join", "update"};
    float tot = 0;
    for (int i = 0; i < 7; ++i) { float ms = 0; hipEventElapsedTime(&ms, h->ph[i], h->ph[i + 1]); tot += ms; fprintf(stderr, "[phase] %-24s %7.1f us\n", nm[i], ms * 1000); }
    fprintf(stderr, "[phase] total %.1f us\n", tot * 1000);
}
#define PHASE(i) phase_mark(h, i)
#else
#define PHASE(i) ((void)0)
#endif

int gemm(fsmg_model* h, const Lane& ln, int amode, int bmode, GemmArgs g, OpBatch* defer, double* sq, bool* sq_done) {
    hipStream_t s = ln.s;
    g.bx3 = h->bx3;
    if (sq_done) *sq_done = false;
    int slots = ln.slots, tile_mn = 0;
    if (use_h_gemm(h, amode, bmode, g, ln)) {
        g.bx3 = 3; slots = 256; tile_mn = 256;
    } else if (use_ws_gemm(h, amode, bmode, g, ln)) {
        g.bx3 = 2; slots = 512 * 4 / 3;             // pick_split takes 3/4 of `slots` for the bf16-split kernels: 512 here
        if (amode == OP_KC && bmode == OP_XC) g.group_m = 4;
    }
    const int S = (g.ldc == g.N) ? pick_split(g.M, g.N, g.K, slots, g.bx3 != 0, tile_mn) : 1;
    const int64_t mn = (int64_t)g.M * g.N;
    float* slabs = ln.slabs; float* cslabs = ln.colsum_slabs;
    bool deferred = false;
    if (S > 1 && defer != nullptr && ln.s == h->stream && h->arena != nullptr) {
        const int64_t need = round_up((int64_t)S * mn, 64) + (g.colsum ? round_up((int64_t)S * g.N, 64) : 0);
        if (h->arena_off + need <= h->arena_cap) {
            slabs = h->arena + h->arena_off; cslabs = slabs + round_up((int64_t)S * mn, 64);
            h->arena_off += need;
            deferred = true;
        }
    }
    if (S <= 1 || (!deferred && (int64_t)S * mn > h->slab_cap)) {
        if (S > 1 && !h->warned_split) {           // a shape-dependent cliff: say so once
            h->warned_split = true;
            fprintf(stderr, "[fsmg] split-K of a %d x %d x %d GEMM dropped: %d slabs do not fit the slab buffer (%lld floats)\n", g.M, g.N, g.K, S, (long long)h->slab_cap);
        }
        g.ksplit = 1;
        HIPCK(h, launch_gemm(s, amode, bmode, g, ln.lds_pad));
        return FSMG_OK;
    }
    float* C = g.C; float* colsum = g.colsum;
    g.C = slabs; g.c_slab = mn; g.ksplit = S;
    if (colsum) { g.colsum = cslabs; g.colsum_slab = g.N; }
    HIPCK(h, launch_gemm(s, amode, bmode, g, ln.lds_pad));
    if (deferred) {
        GEMMCK(defer->room(colsum ? 2 : 1));
        GEMMCK(defer->reduce(slabs, mn, S, C, mn, sq));
        if (colsum) GEMMCK(defer->reduce(cslabs, g.N, S, colsum, g.N));
        if (sq_done) *sq_done = sq != nullptr;
        return FSMG_OK;
    }
    HIPCK(h, launch_reduce_slabs2(s, slabs, mn, S, C, mn, cslabs, g.N, colsum, colsum ? g.N : 0));
    return FSMG_OK;
}

// the XCD-local kernels take this row count at this hidden size (and their buffers exist)
inline bool use_xcd(const fsmg_model* h, int B, bool backward = false) {
    if (h->Hp == 1024 && h->pair_mode < (backward ? 2 : 1)) return false;
    return h->persist && h->xcd && h->khx != nullptr && h->HX != nullptr && B <= h->xcd_max_rows && lstm_xcd_supported(B, h->Hp) &&
           lstm_xcd_hx_floats(B, h->T, h->Hp, h->xcd_bx3) <= h->hx_floats &&
           ((h->Hp == 1024 && !backward) || lstm_xcd_inbox_floats(B, h->Hp) <= h->inboxx_floats);
}
// Two-stream (eager) or single-stream (hipGraph replay) order for a pass over B sequences.  The XCD-local recurrent kernels
// put a high-priority wave on every SIMD of the chip and spend half of their time in hand-offs; GEMM waves beside them
// stretch both (measured at cfg-B: 374-379 episodes/s two-stream with 1-4 chunks against 383-385 single-stream), so a pass
// that takes them runs single-stream; the per-step kernels of big validation batches keep the overlap.
inline void choose_schedule(fsmg_model* h, int B, bool train = false) {
    h->ov_call = h->overlap && (h->overlap_forced || !use_xcd(h, B));
    h->xov_call = false;
    // a pass whose recurrence is one persistent launch per direction is short enough to issue eagerly; per-step kernels (big
    // validation batches, the fallback after a time-out) keep the graph
    h->eager_call = h->eager && !h->ov_call && h->persist && h->persist_fwd && h->persist_bwd &&
                    (use_xcd(h, B) || lstm_fwd_chain_supported(B, h->Hp) || lstm_fwd_chain_rt_supported(B, h->Hp));
    // XCD-partitioned schedule (round 4 form): the bf16-split chains packed on ceil(B / 16) XCDs, the 256-tile work-queue GEMMs of
    // the projection / its weight gradient on the others
    if (train && h->xov && h->Hp == 512 && h->xcd_bx3 && h->bx3 && h->L == 1 && !h->ov_call && h->aux != nullptr && use_xcd(h, B) && h->persist_fwd && h->persist_bwd &&
        (!h->timing || h->timing_only == "lstm_fwd" || h->timing_only == "lstm_bwd")) {
        const int rpx = lstm_xcd16_packed_rows(B);
        h->xov_call = rpx > 0 && (B + rpx - 1) / rpx <= 5;          // at least three XCDs for the GEMMs
        if (h->xov_call) h->eager_call = true;
    }
}
// first XCD the packed recurrence leaves free
inline int xov_first_free(int B) { const int rpx = lstm_xcd16_packed_rows(B); return rpx > 0 ? (B + rpx - 1) / rpx : 8; }

// every XCD-local launch of a pass gets its own 8 zeroed ticket counters
inline int* next_tickets(fsmg_model* h) {
    int* t = h->tickets + 8 * (h->ticket_next % fsmg_model::TICKET_LAUNCHES);
    ++h->ticket_next;
    return t;
}

#ifdef FSMG_EXPERIMENTS
inline int xov_debug() { static const int dbg = std::getenv("FSMG_XOV_DEBUG") ? std::atoi(std::getenv("FSMG_XOV_DEBUG")) : 0; return dbg; }
#endif
// Work-queue GEMM in two launches of k_gemm_bx3h<..., QUEUE> (GemmArgs::xcd_first): the restricted one lets the XCDs >= first
// draw items (all of them: the two launches drain one queue); the clean-up one, ordered behind the kernel that owned the other
// XCDs, lets the whole chip take what is left.  work / claim words are zeroed on the main stream before the fork.
inline int gemm_items(const GemmArgs& g) { return ((g.M + 255) / 256) * ((g.N + 255) / 256) * std::max(1, g.ksplit); }
inline bool xov_fits(const GemmArgs& g) { return 4 + gemm_items(g) <= fsmg_model::XOV_CTL; }
inline void xov_gate(fsmg_model* h, GemmArgs& g, int B) {     // the projection's A rows arrive time step by time step
    const int rpx = lstm_xcd16_packed_rows(B);
    g.gate = h->xov_prog; g.gate_expect = lstm_xcd_active_blocks(B, rpx); g.gate_rows = B; g.gate_last = h->T - 1;       // (blocks below xcd_first join when the CHAIN is over)
    // a tile waits for its rows for a fraction of the chain's 0.4 ms; 0.2 s of ~1 us polls without them (a host that was descheduled
    // between the two launches is back long before that) means the launches are not running side by side -- a profiler or debugger
    // that serialises dispatches: give up like any timed-out hand-off
    g.gate_err = h->d_err; g.gate_spin = h->chain_spin_limit > 0 ? 200000 : 0; g.gate_every = h->xov_pub;
}
int gemm_restricted(fsmg_model* h, hipStream_t s, int amode, int bmode, GemmArgs g, int first, int* ctl) {
    g.bx3 = 3; g.xcd_first = first; g.work = ctl; g.stop = ctl + 2; g.claim = ctl + 4; g.work_limit = gemm_items(g);
#ifdef FSMG_EXPERIMENTS         // FSMG_XOV_DEBUG (make experiments): the A/B runs of DESIGN.md 9.2
    const int dbg = xov_debug();
    if (dbg & 1) g.work_limit = 0;                 // nothing for the restricted launch: the serial order on the packed kernels
    if ((dbg & 1) && (dbg & (128 | 256))) g.gate = nullptr;
    if (dbg & 16) g.dbg |= 32;                     // agent-scope loads of the gated operand
    if (dbg & 8) g.dbg |= 128;                     // agent-scope acquire behind the gate
    if (dbg & 32) g.dbg |= 64;                     // blocks below xcd_first never join
#endif
    HIPCK(h, launch_gemm(s, amode, bmode, g, 0));
    return FSMG_OK;
}
int gemm_cleanup(fsmg_model* h, hipStream_t s, int amode, int bmode, GemmArgs g, int* ctl) {
    g.bx3 = 3; g.xcd_first = -1; g.work = ctl; g.claim = ctl + 4;
#ifdef FSMG_EXPERIMENTS
    if ((xov_debug() & 1) && (xov_debug() & (128 | 256))) g.gate = nullptr;
#endif
    HIPCK(h, launch_gemm(s, amode, bmode, g, 0));
    return FSMG_OK;
}

// logits of the rows of time steps [t0, t1) = top-layer outputs * W + d
GemmArgs logits_args(fsmg_model* h, int B, int t0, int t1) {
    const int Hp = h->Hp;
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    GemmArgs g{};
    g.A = h->Hs[h->L - 1] + (size_t)B * Hp + (size_t)r0 * Hp; g.lda = Hp;
    g.B = h->P + h->off_w; g.ldb = h->V1p;
    g.C = h->logits + (size_t)r0 * h->V1p; g.ldc = h->V1p; g.M = (int)m; g.N = h->V1p; g.K = Hp;
    g.bias = h->P + h->off_d; g.ksplit = 1; g.nt_store = 1;      // streaming stores: read back by the cross entropy much later (A/B: profiles/r03t_ntp_*)
    return g;
}
int ce_rows(fsmg_model* h, hipStream_t s, int B, int t0, int t1, int64_t rows_total) {
    ScopedTimer tm(h, "ce");
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    HIPCK(h, launch_ce_rows(s, h->logits + (size_t)r0 * h->V1p, h->V1p, (int)m, h->V1, h->Y + r0, h->lse + r0,
                            h->ce + r0, h->dlogits + (size_t)r0 * h->V1p, (float)(1.0 / ((double)rows_total + 1e-12))));
    return FSMG_OK;
}

int logits_and_ce(fsmg_model* h, const Lane& ln, int B, int t0, int t1, int64_t rows_total, bool want_dlogits) {
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    GemmArgs g = logits_args(h, B, t0, t1);
    if (!want_dlogits) {
        // validation: no backward pass will read the logits, so they are never written; the GEMM epilogue emits
        // per-row softmax partials and a small kernel finishes the cross entropy
        ScopedTimer tm(h, "gemm_logits");
        g.ce_part = h->ce_part + (size_t)r0 * h->ce_nparts; g.ce_tgt = h->Y + r0; g.ce_tgt_logit = h->tgt_logit + r0;
        g.ce_nvocab = h->V1; g.bx3 = h->bx3;
        if (use_h_gemm(h, OP_KC, OP_XC, g, ln)) g.bx3 = 3;                  // (the softmax partials are per 64-column half of a 128-column tile in every kernel)
        else if (use_ws_gemm(h, OP_KC, OP_XC, g, ln)) { g.bx3 = 2; g.group_m = 4; }
        HIPCK(h, launch_gemm(ln.s, OP_KC, OP_XC, g, ln.lds_pad));          // K = Hp: never split
        HIPCK(h, launch_ce_combine(ln.s, h->ce_part + (size_t)r0 * h->ce_nparts, h->ce_nparts,
                                   h->tgt_logit + r0, (int)m, h->ce + r0));
        return FSMG_OK;
    }
    {
        ScopedTimer tm(h, "gemm_logits");
        GEMMCK(gemm(h, ln, OP_KC, OP_XC, g));
    }
    return ce_rows(h, ln.s, B, t0, t1, rows_total);
}

// FSMG_FILL_EARLY=1 (A/B): the hand-off fills of a layer's forward chain in front of its x-part GEMM instead of right in front of
// the chain, so that the chain does not start on an L2 full of fill lines
int chain_fills_early(fsmg_model* h, OpBatch& fills, int l, int B, bool chain, bool xcd, int xov_words) {
    const int T = h->T, Hp = h->Hp;
    const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
    if (chain) GEMMCK(fills.add(h->HF[l] + Bp16 * Hp, 0xFFFFFFFFu, (long long)T * Bp16 * Hp));
    if (xcd) {
        GEMMCK(fills.add(h->tickets, 0u, (long long)8 * fsmg_model::TICKET_LAUNCHES));
        h->ticket_next = 0;
        const long long step_f = lstm_xcd_hx_floats(B, 0, Hp, h->xcd_bx3);
        GEMMCK(fills.add(h->HX, 0u, step_f));
        GEMMCK(fills.add(h->HX + step_f, 0xFFFFFFFFu, step_f * T));
        if (xov_words > 0) GEMMCK(fills.add(h->xov_ctl, 0u, xov_words));
    }
    return fills.flush();
}

int forward(fsmg_model* h, int B, int rows_per_group, int ngroups, float* loss_out, bool want_dlogits) {
    ScopedRange rng_(want_dlogits ? "fsmg.forward(train)" : "fsmg.forward(eval)");
    const int T = h->T, Hp = h->Hp, G4 = h->G4;
    const int64_t rows = (int64_t)T * B;
    const Lane mainl = main_lane(h);
    hipStream_t s = h->stream;
    const bool ov = use_overlap(h);
    const bool xcd = use_xcd(h, B) && h->persist_fwd;
    const bool chain1 = !xcd && h->persist && h->persist_fwd && !h->force_fwd_rt && lstm_fwd_chain_supported(B, Hp);
    const bool chain_rt = !xcd && h->persist && h->persist_fwd && !chain1 && lstm_fwd_chain_rt_supported(B, Hp);   // all row tiles per block
    const bool chain = chain1 || chain_rt;
    const int nch_ov = ov ? ((chain || xcd) ? h->nchunk_persist : h->nchunk) : 1;
    // XCD-partitioned schedule: the chain packed on the first XCDs publishes the time steps it has finished, the projection's
    // row tiles are drawn by the other XCDs as their rows arrive (and by the whole chip once the chain is over)
    // The queue takes the rows of the time steps [0, t_cut); the last few steps' rows (complete only when the chain is) go to a
    // chip-wide launch of the 128-tile kernel behind it: a 256 x 256 tile is 85 us of latency with 1/6 of the CUs busy, the same
    // rows as 128 x 128 tiles are one under-full round of ~40 us (same bits: the kernels share k order and term order, K = H is never split)
    // (t_cut is a multiple of the publishing period: the queue's last row tile then waits for a step that IS published)
    const int t_cut = (T >= 4 * h->xov_tail && h->xov_tail > 0) ? (T - h->xov_tail) / h->xov_pub * h->xov_pub : T;
    GemmArgs ghead = logits_args(h, B, 0, t_cut);
    const bool xov = h->xov_call && (h->xov_parts & 1) && xcd && want_dlogits && !ov && xov_fits(ghead);
    const int xfree = xov_first_free(B);
    const int rpx = xov ? lstm_xcd16_packed_rows(B) : 0;
    if (xov) xov_gate(h, ghead, B);
    PHASE(0);
    for (int l = 0; l < h->L; ++l) {
        const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
        const bool top = l == h->L - 1;
        FillBatch fills(h);                 // zero states + hand-off patterns of this layer: ONE launch, issued ahead of the chain
        GEMMCK(fills.add(h->Hs[l], 0u, (long long)B * Hp));
        GEMMCK(fills.add(h->Cs[l], 0u, (long long)B * Hp));
        if (!xcd) GEMMCK(fills.add(h->HF[l], 0u, (long long)Bp16 * Hp));
        if (h->fill_early && !xov) GEMMCK(chain_fills_early(h, fills, l, B, chain, xcd, 0));
        {
            ScopedTimer tm(h, "gemm_zx");
            GemmArgs g{};
            if (l == 0) { g.A = h->P + h->off_emb; g.lda = h->Ep; g.gather = h->X; g.K = h->Ep; }
            else { g.A = h->Hs[l - 1] + (size_t)B * Hp; g.lda = Hp; g.K = Hp; }
            g.B = h->P + h->off_kx[l]; g.ldb = G4;
            g.C = h->Z[l]; g.ldc = G4; g.M = (int)rows; g.N = G4;
            g.bias = h->P + h->off_b[l]; g.ksplit = 1;
            GEMMCK(gemm(h, mainl, OP_KC, OP_XC, g));
        }
        PHASE(1);
        auto chain_fills = [&]() -> int {
            if (chain)       // "not written yet" fill pattern of the h fragments of time indices 1..T (index 0 is the zero state)
                GEMMCK(fills.add(h->HF[l] + Bp16 * Hp, 0xFFFFFFFFu, (long long)T * Bp16 * Hp));
            if (xcd) {       // the same for the XCD-local hand-off buffer, and fresh ticket counters for this layer's launches
                GEMMCK(fills.add(h->tickets, 0u, (long long)8 * fsmg_model::TICKET_LAUNCHES));
                h->ticket_next = 0;
                const long long step_f = lstm_xcd_hx_floats(B, 0, Hp, h->xcd_bx3, xov && top ? rpx : 0);
                GEMMCK(fills.add(h->HX, 0u, step_f));
                GEMMCK(fills.add(h->HX + step_f, 0xFFFFFFFFu, step_f * T));
                if (xov && top) {
                    GEMMCK(fills.add(h->xov_ctl, 0u, 4 + gemm_items(ghead)));
                    GEMMCK(fills.add(h->xov_prog, 0u, T));
                }
            }
            return fills.flush();
        };
        if (!h->fill_early || xov) GEMMCK(chain_fills());
        if (xov && top) {         // the projection's queue launch on the auxiliary stream, confined to the XCDs the chain leaves free
            HIPCK(h, hipEventRecord(h->ev_fork, s));
            HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
            GEMMCK(gemm_restricted(h, h->aux, OP_KC, OP_XC, ghead, xfree, h->xov_ctl));
            HIPCK(h, hipEventRecord(h->ev_join, h->aux));
        }
        const int nch = nch_ov;
        for (int c = 0; c < nch; ++c) {
            const int t0 = chunk_begin(h, c, nch);
            const int t1 = chunk_begin(h, c + 1, nch);
            if (xcd) {
                ScopedTimer tm(h, "lstm_fwd");
                LstmFwdXcdArgs a{};
                a.rpx = (xov && top) ? rpx : 0; a.progress = (xov && top) ? h->xov_prog : nullptr; a.Hp = Hp;
#ifdef FSMG_EXPERIMENTS
                { const int dbg = xov_debug(); a.progress_lag = ((dbg & 2) ? 2 : 0) | ((dbg & 256) ? 256 : 0); if ((dbg & 128) && (dbg & 1)) a.progress = nullptr; }
#endif
                a.progress_every = h->xov_pub; a.bx3 = h->xcd_bx3 ? 1 : 0;
                a.variant = h->xcd_variant >= 0 ? h->xcd_variant : lstm_xcd_default_variant(B, true, Hp, a.rpx);
                a.KhX = h->khx + (size_t)(2 * l) * lstm_xcd_weight_floats((int)Hp, h->xcd_bx3); a.HX = h->HX; a.Z = h->Z[l]; a.Cs = h->Cs[l]; a.Hs = h->Hs[l];
                a.tickets = next_tickets(h); a.err_flag = h->d_err; a.B = B; a.T = T; a.t0 = t0; a.t1 = t1; a.spin_limit = h->chain_spin_limit;
                HIPCK(h, launch_lstm_fwd_xcd(s, a));
                ++h->n_xcd_launches;
            } else if (chain) {
                ScopedTimer tm(h, "lstm_fwd");
                LstmFwdChainArgs a{};
                a.KhF = h->khf + (size_t)(2 * l) * Hp * G4; a.HF = h->HF[l]; a.Z = h->Z[l]; a.Cs = h->Cs[l]; a.Hs = h->Hs[l];
                a.err_flag = h->d_err; a.B = B; a.Hp = Hp; a.T = T; a.t0 = t0; a.t1 = t1; a.spin_limit = h->chain_spin_limit;
                HIPCK(h, chain_rt ? launch_lstm_fwd_chain_rt(s, a) : launch_lstm_fwd_chain(s, a));
                ++h->n_persist_launches;
            } else {
                ScopedTimer tm(h, "lstm_fwd");
                for (int t = t0; t < t1; ++t) {
                    LstmFwdArgs a{};
                    a.KhF = h->khf + (size_t)(2 * l) * Hp * G4;
                    a.hF_prev = h->HF[l] + (size_t)t * Bp16 * Hp;
                    a.hF_next = h->HF[l] + (size_t)(t + 1) * Bp16 * Hp;
                    a.z = h->Z[l] + (size_t)t * B * G4;
                    a.c_prev = h->Cs[l] + (size_t)t * B * Hp;
                    a.c_next = h->Cs[l] + (size_t)(t + 1) * B * Hp;
                    a.h_next = h->Hs[l] + (size_t)(t + 1) * B * Hp;
                    a.B = B; a.Hp = Hp;
                    HIPCK(h, launch_lstm_fwd_step(s, a));
                }
                h->n_step_launches += t1 - t0;
            }
            if (top && ov) {      // projection + CE of this chunk on the auxiliary stream
                HIPCK(h, hipEventRecord(h->ev_chunk[c], s));
                HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_chunk[c], 0));
                GEMMCK(logits_and_ce(h, aux_lane(h, !want_dlogits, chain || xcd), B, t0, t1, rows, want_dlogits));
            }
        }
    }
    PHASE(2);
    if (ov) {
        HIPCK(h, hipEventRecord(h->ev_join, h->aux));
        HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));
    } else if (xov) {
        {
            ScopedTimer tm(h, "gemm_logits");      // (what is left of the queue when the chain is over, on the whole chip)
            if (t_cut < T) {       // the last steps' rows as 128 x 128 tiles (three blocks per CU): they fit the XCDs the chain has just left,
                GemmArgs gt = logits_args(h, B, t_cut, T);      // beside the queue's tiles still in flight on the others
                gt.bx3 = 1; gt.ksplit = 1;
                HIPCK(h, launch_gemm(s, OP_KC, OP_XC, gt, 0));
            }
            GEMMCK(gemm_cleanup(h, s, OP_KC, OP_XC, ghead, h->xov_ctl));
            HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));    // the restricted launch and its tiles in flight
        }
        GEMMCK(ce_rows(h, s, B, 0, T, rows));
    } else {
        GEMMCK(logits_and_ce(h, mainl, B, 0, T, rows, want_dlogits));
    }
    if (!want_dlogits) {         // (a train pass reduces its loss in backward(): k_sum_partials, no launch of its own)
        ScopedTimer tm(h, "ce");
        HIPCK(h, launch_loss_reduce(s, h->ce, T, B, rows_per_group, ngroups, loss_out));
    }
    h->lastB = B;
    return FSMG_OK;
}

int dhout_chunk(fsmg_model* h, const Lane& ln, int B, int t0, int t1, OpBatch* defer = nullptr) {
    ScopedTimer tm(h, "gemm_dhout");     // dH = dlogits * W^T for the rows of time steps [t0, t1)
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    GemmArgs g{};
    g.A = h->dlogits + (size_t)r0 * h->V1p; g.lda = h->V1p; g.B = h->P + h->off_w; g.ldb = h->V1p;
    g.C = h->dH + (size_t)r0 * h->Hp; g.ldc = h->Hp; g.M = (int)m; g.N = h->Hp; g.K = h->V1p; g.ksplit = 1;
    return gemm(h, ln, OP_KC, OP_KC, g, defer);
}

GemmArgs dw_args(fsmg_model* h, int B) {      // dW = Hout^T * dlogits, dd = colsum(dlogits)
    GemmArgs g{};
    g.A = h->Hs[h->L - 1] + (size_t)B * h->Hp; g.lda = h->Hp; g.B = h->dlogits; g.ldb = h->V1p;
    g.C = h->G + h->off_w; g.ldc = h->V1p; g.M = h->Hp; g.N = h->V1p; g.K = (int)((int64_t)h->T * B);
    g.colsum = h->G + h->off_d; g.ksplit = 1;
    return g;
}
int dw_gemm(fsmg_model* h, const Lane& ln, int B, OpBatch* defer = nullptr) {
    ScopedTimer tm(h, "gemm_dw");
    return gemm(h, ln, OP_XC, OP_XC, dw_args(h, B), defer);
}

// what a BPTT chain wants filled before it starts (dC zero, hand-off patterns, ticket counters); rpx: rows packed per XCD
int bptt_fills(fsmg_model* h, OpBatch& fills, int B, bool xcd, bool rs, bool chain, int rpx) {
    const int T = h->T, Hp = h->Hp, G4 = h->G4;
    GEMMCK(fills.add(h->dC, 0u, (long long)B * Hp));
    if (xcd) {
        // the dh-partial inboxes are refilled only when the device flag says so: every word a pass writes is read and reset
        // by its consumer, so a completed pass leaves them all-"not written" (33 MB per pass at hidden 512, 100 MB per
        // layer at hidden 1024 otherwise)
        GEMMCK(fills.add(h->inboxX, 0xFFFFFFFFu, lstm_xcd_inbox_floats(B, Hp, rpx), h->d_inbox_dirty));    // same launch as the zero fills
        GEMMCK(fills.add(h->tickets, 0u, (long long)8 * fsmg_model::TICKET_LAUNCHES));
        h->ticket_next = 0;
    } else if (rs) {        // "not written yet" fill pattern of the dh partial inboxes
        GEMMCK(fills.add(h->inbox, 0xFFFFFFFFu, lstm_bwd_rs_inbox_floats(B, Hp)));
    } else if (chain) {     // ... or of the dz fragments of every time step
        const long long Bp16 = (B + 15) / 16 * 16;
        GEMMCK(fills.add(h->dzF_all, 0xFFFFFFFFu, (long long)T * Bp16 * G4));
    }
    return FSMG_OK;
}

// part 0: the whole pass; part 1: up to and including the projection gradients (dH, dW, dd: bucket 0 of the gradient exchange
// is final behind it); part 2: the rest.  Only the single-stream order can be cut there: the two-stream and the XCD-partitioned
// orders run everything in part 1 (they record bucket 0 themselves) and nothing in part 2.
int backward(fsmg_model* h, int B, int part = 0) {
    ScopedRange rng_(part == 2 ? "fsmg.backward(2)" : "fsmg.backward");
    const int T = h->T, Hp = h->Hp, G4 = h->G4;
    const int64_t rows = (int64_t)T * B;
    const Lane mainl = main_lane(h);
    hipStream_t s = h->stream;
    const bool ov = use_overlap(h);
    const bool xcd = use_xcd(h, B, true) && h->persist_bwd;
    const bool rs = !xcd && h->persist && h->persist_bwd && h->inbox != nullptr && lstm_bwd_rs_supported(B, Hp) && lstm_bwd_rs_inbox_floats(B, Hp) <= h->inbox_floats;
    const bool chain = xcd || rs || (h->persist && h->persist_bwd && h->dzF_all != nullptr && lstm_bwd_chain_supported(B, Hp) &&
                              (int64_t)T * ((B + 15) / 16 * 16) * G4 <= h->dzfa_floats);
    const int nch = ov ? (chain ? h->nchunk_persist : h->nchunk) : 1;
    const Lane auxl = aux_lane(h, false, chain);
    // XCD-partitioned schedule: dW's tiles are claimed by the free XCDs while the top layer's chain runs, the rest after it
    GemmArgs gdw = dw_args(h, B);
    int dw_split = 1;
    if (h->xov_call && (h->xov_parts & 2) && xcd && !ov) dw_split = std::max(1, std::min(std::min(h->xov_dw_split, MAX_SPLIT), gdw.K / 256));
    while (dw_split > 1 && (int64_t)dw_split * gdw.M * gdw.N > h->slab_cap) --dw_split;
    if (dw_split > 1) {
        gdw.C = mainl.slabs; gdw.c_slab = (int64_t)gdw.M * gdw.N; gdw.ksplit = dw_split;
        gdw.colsum = mainl.colsum_slabs; gdw.colsum_slab = gdw.N;
    }
    const bool xov = h->xov_call && (h->xov_parts & 2) && xcd && !ov && dw_split > 1 && xov_fits(gdw);
    const int rpx = xov ? lstm_xcd16_packed_rows(B) : 0;
    const bool cut = !ov && !xov;           // the order that can be cut behind the projection gradients
    if (part == 2 && !cut) return FSMG_OK;
    // dp_split == 2: the cut sits behind the LAST recurrent chain instead -- an XCD-local chain needs every CU of the chip, so
    // a collective kernel started in front of it only delays it; behind it the exchange of bucket 0 runs beside the
    // weight- / input-gradient GEMMs of the bottom layer, the embedding gradient and the norm
    const bool cut_late = cut && h->dp_split == 2 && part != 0;
    bool dx_sq_done = false, top_fills_done = false;
    PHASE(3);
    // Slab sums of the split-K GEMMs ride in two launches per pass instead of one each: `fills` (issued right in front of a
    // recurrent chain: what the chain reads -- dH -- plus the fills) and `late` (in front of the embedding gradient: every
    // weight gradient + dx with its squared-norm partials).  Only the order that runs start to end on one stream in one call
    // defers; the cut (episode-parallel) and overlapped orders keep a sum behind each GEMM.
    const bool defer_ok = part == 0 && (cut || xov);
    if (part != 2) h->arena_off = 0;
    h->last_bwd_xcd = xcd;
    FillBatch fills(h);                     // embedding-gradient zero + the top layer's BPTT buffers: one launch
    OpBatch late(h);
    OpBatch* const d_now = defer_ok ? &fills : nullptr;
    OpBatch* const d_late = defer_ok ? &late : nullptr;
    if (part != 2) GEMMCK(fills.add(h->G + h->off_emb, 0u, (long long)h->V1 * h->Ep));
    if (ov) GEMMCK(fills.flush());          // (two-stream order: the auxiliary stream forks right below)
    if (part == 2) {
    } else if (ov) {
        // aux: dH chunks in the order BPTT consumes them (last chunk first), then dW
        HIPCK(h, hipEventRecord(h->ev_fork, s));
        HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
        for (int c = nch - 1; c >= 0; --c) {
            const int t0 = chunk_begin(h, c, nch), t1 = chunk_begin(h, c + 1, nch);
            GEMMCK(dhout_chunk(h, auxl, B, t0, t1));
            HIPCK(h, hipEventRecord(h->ev_chunk[c], h->aux));
        }
        GEMMCK(dw_gemm(h, auxl, B));
        HIPCK(h, hipEventRecord(h->ev_join, h->aux));
        HIPCK(h, hipEventRecord(h->ev_bucket[0], h->aux));
        h->bucket0_recorded = true;
    } else if (xov) {
        GEMMCK(fills.add(h->xov_ctl + fsmg_model::XOV_CTL, 0u, 4 + gemm_items(gdw)));
        GEMMCK(fills.flush());                // (the queue words must be zero before the auxiliary stream forks)
        GEMMCK(dhout_chunk(h, mainl, B, 0, T, d_now));
        HIPCK(h, hipEventRecord(h->ev_fork, s));
        HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
        GEMMCK(gemm_restricted(h, h->aux, OP_XC, OP_XC, gdw, xov_first_free(B), h->xov_ctl + fsmg_model::XOV_CTL));
        HIPCK(h, hipEventRecord(h->ev_join, h->aux));
    } else {
        GEMMCK(dhout_chunk(h, mainl, B, 0, T, d_now));
        if (defer_ok && !h->fills_late) {     // dH's slab sum + the top chain's fills go out in front of dW: the chain starts right behind a GEMM
            GEMMCK(bptt_fills(h, fills, B, xcd, rs, chain, 0));
            GEMMCK(fills.flush());
            top_fills_done = true;
        }
        GEMMCK(dw_gemm(h, mainl, B, d_late));
    }
    if (part == 1 && cut && !cut_late) return fills.flush();
    for (int l = h->L - 1; l >= 0; --l) {
        const bool top = l == h->L - 1;
        if (part == 2 && cut_late && l > 0) continue;                       // done in part 1
        const bool skip_chain = part == 2 && cut_late;                      // layer 0: its chain ran in part 1
        if (!skip_chain) {
        if (top && ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_chunk[nch - 1], 0));
        PHASE(4);
        if (!(top && top_fills_done)) {
            GEMMCK(bptt_fills(h, fills, B, xcd, rs, chain, xov && top ? rpx : 0));
            GEMMCK(fills.flush());
        }
        for (int c = nch - 1; c >= 0; --c) {
            const int t0 = chunk_begin(h, c, nch), t1 = chunk_begin(h, c + 1, nch);
            if (top && ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_chunk[c], 0));
            ScopedTimer tm(h, "lstm_bwd");
            if (xcd) {
                LstmBwdXcdArgs a{};
                a.rpx = (xov && top) ? rpx : 0; a.Hp = Hp; a.bx3 = h->xcd_bx3 ? 1 : 0; a.variant = h->xcd_variant >= 0 ? h->xcd_variant : lstm_xcd_default_variant(B, false, Hp);
                a.KhXb = h->khx + (size_t)(2 * l + 1) * lstm_xcd_weight_floats((int)Hp, h->xcd_bx3); a.inbox = h->inboxX; a.Z = h->Z[l]; a.Cs = h->Cs[l];
                a.dc = h->dC; a.dH = h->dH; a.tickets = next_tickets(h); a.err_flag = h->d_err; a.B = B; a.T = T; a.t0 = t0; a.t1 = t1; a.spin_limit = h->chain_spin_limit;
                HIPCK(h, launch_lstm_bwd_xcd(s, a));
                ++h->n_xcd_launches;
                continue;
            }
            if (rs) {
                LstmBwdRsArgs a{};
                a.KhF = h->khf + (size_t)(2 * l + 1) * Hp * G4; a.inbox = h->inbox; a.Z = h->Z[l]; a.Cs = h->Cs[l];
                a.dc = h->dC; a.dH = h->dH; a.err_flag = h->d_err; a.B = B; a.Hp = Hp; a.T = T; a.t0 = t0; a.t1 = t1; a.spin_limit = h->chain_spin_limit;
                HIPCK(h, launch_lstm_bwd_rs(s, a));
                ++h->n_persist_launches;
                continue;
            }
            if (chain) {
                LstmBwdChainArgs a{};
                a.KhF = h->khf + (size_t)(2 * l + 1) * Hp * G4; a.dzF_all = h->dzF_all; a.Z = h->Z[l]; a.Cs = h->Cs[l];
                a.dc = h->dC; a.dH = h->dH; a.err_flag = h->d_err; a.B = B; a.Hp = Hp; a.T = T; a.t0 = t0; a.t1 = t1; a.spin_limit = h->chain_spin_limit;
                HIPCK(h, launch_lstm_bwd_chain(s, a));
                ++h->n_persist_launches;
                continue;
            }
            for (int t = t1 - 1; t >= t0; --t) {
                LstmBwdArgs a{};
                const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
                a.KhF = h->khf + (size_t)(2 * l + 1) * Hp * G4;
                a.dzF_next = (t + 1 < T) ? h->dzF + (size_t)((t + 1) & 1) * Bp16 * G4 : nullptr;
                a.dzF_cur = h->dzF + (size_t)(t & 1) * Bp16 * G4;
                a.gates = h->Z[l] + (size_t)t * B * G4;
                a.c_t = h->Cs[l] + (size_t)(t + 1) * B * Hp;
                a.c_prev = h->Cs[l] + (size_t)t * B * Hp;
                a.dc = h->dC;
                a.dh_top = h->dH + (size_t)t * B * Hp;
                a.B = B; a.Hp = Hp;
                HIPCK(h, launch_lstm_bwd_step(s, a));
            }
            h->n_step_launches += t1 - t0;
        }
        if (xov && top) {                     // the rest of dW chip-wide, then the fixed-order slab sum
            ScopedTimer tm(h, "gemm_dw");
            GEMMCK(gemm_cleanup(h, s, OP_XC, OP_XC, gdw, h->xov_ctl + fsmg_model::XOV_CTL));
            HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));
            if (dw_split > 1) {
                const int64_t mn = (int64_t)gdw.M * gdw.N;
                if (d_late) {                     // the fixed-order slab sums ride with the other weight gradients'
                    GEMMCK(late.reduce(mainl.slabs, mn, dw_split, h->G + h->off_w, mn));
                    GEMMCK(late.reduce(mainl.colsum_slabs, gdw.N, dw_split, h->G + h->off_d, gdw.N));
                } else {
                    HIPCK(h, launch_reduce_slabs(s, mainl.slabs, mn, dw_split, h->G + h->off_w, mn));
                    HIPCK(h, launch_reduce_slabs(s, mainl.colsum_slabs, gdw.N, dw_split, h->G + h->off_d, gdw.N));
                    HIPCK(h, hipEventRecord(h->ev_bucket[0], s));
                    h->bucket0_recorded = true;
                }
            }
        }
        }   // !skip_chain
        if (part == 1 && cut_late && l == 0) return FSMG_OK;                // bucket 0 travels beside what follows
        const int in_p = h->in_dim[l];
        PHASE(5);
        {
            ScopedTimer tm(h, "gemm_dk");
            // dKx and dKh are one matrix of the flat gradient (the [in | h_prev] rows of `kernel_l`) and contract the same dZ over the
            // same rows: where the 256 x 256-tile kernel takes the shape they are ONE GEMM with a two-part A (GemmArgs::m_split) --
            // one K split and one set of slabs instead of two (cfg-B: 63 MB of slabs instead of 100), no 128-tile launch for dKx
            GemmArgs m{};
            if (l == 0) { m.A = h->P + h->off_emb; m.lda = h->Ep; m.gather = h->X; }
            else { m.A = h->Hs[l - 1] + (size_t)B * Hp; m.lda = Hp; }
            m.A2 = h->Hs[l]; m.lda2 = Hp; m.m_split = in_p;
            m.B = h->Z[l]; m.ldb = G4; m.C = h->G + h->off_kx[l]; m.ldc = G4; m.M = in_p + Hp; m.N = G4; m.K = (int)rows;
            m.colsum = h->G + h->off_b[l]; m.ksplit = 1;
            const bool merged = h->merge_dk && in_p % 256 == 0 && Hp % 4 == 0 && h->off_kh[l] == h->off_kx[l] + (int64_t)in_p * G4 &&
                                (l > 0 || 4LL * h->V1 * h->Ep < 0xfffff000LL) && 4LL * Hp * rows < 0xfffff000LL && 4LL * G4 * rows < 0xfffff000LL &&
                                (((uintptr_t)m.A | (uintptr_t)m.A2 | (uintptr_t)m.B) & 15) == 0 && gemm_dma_enabled() && use_h_gemm(h, OP_XC, OP_XC, m, mainl);
            if (merged) {
                GEMMCK(gemm(h, mainl, OP_XC, OP_XC, m, d_late));
            } else {
            GemmArgs g{};                     // dKh = Hprev^T * dZ, db = colsum(dZ)
            g.A = h->Hs[l]; g.lda = Hp; g.B = h->Z[l]; g.ldb = G4;
            g.C = h->G + h->off_kh[l]; g.ldc = G4; g.M = Hp; g.N = G4; g.K = (int)rows;
            g.colsum = h->G + h->off_b[l]; g.ksplit = 1;
            GEMMCK(gemm(h, mainl, OP_XC, OP_XC, g, d_late));
            GemmArgs k{};                     // dKx = in^T * dZ
            if (l == 0) { k.A = h->P + h->off_emb; k.lda = h->Ep; k.gather = h->X; }
            else { k.A = h->Hs[l - 1] + (size_t)B * Hp; k.lda = Hp; }
            k.B = h->Z[l]; k.ldb = G4; k.C = h->G + h->off_kx[l]; k.ldc = G4;
            k.M = in_p; k.N = G4; k.K = (int)rows; k.ksplit = 1;
            GEMMCK(gemm(h, mainl, OP_XC, OP_XC, k, d_late));
            }
        }
        {
            ScopedTimer tm(h, "gemm_dx");     // d_in = dZ * Kx^T
            GemmArgs g{};
            g.A = h->Z[l]; g.lda = G4; g.B = h->P + h->off_kx[l]; g.ldb = G4;
            g.C = (l == 0) ? h->dXemb : h->dH; g.ldc = in_p;
            g.M = (int)rows; g.N = in_p; g.K = G4; g.ksplit = 1;
            // layer 0: the sum rides with the weight gradients' and leaves the squared-norm partials of dXemb behind;
            // above: the layer below reads dH next, the sum goes out with that layer's fills
            if (l == 0) GEMMCK(gemm(h, mainl, OP_KC, OP_KC, g, d_late, h->partials, &dx_sq_done));
            else GEMMCK(gemm(h, mainl, OP_KC, OP_KC, g, d_now));
        }
    }
    {
        ScopedTimer tm(h, "embed_grad");
        // tail[1] = the mean loss of the pass: nobody reads it before the step's last kernels, so it rides with the slab sums (one
        // block of a launch that keeps the rest of the chip busy) instead of costing a launch behind the cross entropy
        const bool loss_in_batch = late.r.count > 0;
        if (loss_in_batch) GEMMCK(late.mean(h->ce, rows, h->G + h->n_flat + 1));
        GEMMCK(late.flush());
        HIPCK(h, launch_embed_grad(s, h->X, (int)rows, h->dXemb, h->Ep, h->G + h->off_emb, h->tok_first, h->tok_count));
        h->tok_table_open = false;
        const int nb = sqnorm_blocks(rows * h->Ep);
        if (!dx_sq_done) HIPCK(h, launch_sqnorm_partials(s, h->dXemb, rows * h->Ep, h->partials));
        // tail[0] = squared norm of the embedding-slice gradients, tail[1] = mean loss of the pass, tail[2] / tail[3] = time-out /
        // token-range indicators
        HIPCK(h, launch_sum_partials(s, h->partials, nb, h->G + h->n_flat + 0, h->d_err, loss_in_batch ? nullptr : h->ce, (int)rows, h->G + h->n_flat + 1));
    }
    if (ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));     // dW / dd landed
    PHASE(6);
    h->have_grads = true;
    return FSMG_OK;
}

// K_h of every layer into the layouts the recurrent kernels read: one launch (up to REPACK_MAX_LAYERS layers).  inc != nullptr:
// the launch also closes the train step (k_step_increment's work on one thread of it); *inc_done says whether it did.
int repack_recurrent_weights(fsmg_model* h, hipStream_t s, const StepIncArgs* inc, bool* inc_done) {
    if (inc_done) *inc_done = false;
    const bool x_ok = h->khx == nullptr || h->Hp == 512 || h->Hp == 1024 || h->Hp == 256;
    if (h->L <= REPACK_MAX_LAYERS && x_ok) {
        RepackAllArgs a{};
        a.n = h->L; a.Hp = h->Hp; a.bx3 = h->xcd_bx3 ? 1 : 0;
        for (int l = 0; l < h->L; ++l) {
            a.Kh[l] = h->P + h->off_kh[l];
            a.cf[l] = h->khf + (size_t)(2 * l) * h->Hp * h->G4; a.cb[l] = h->khf + (size_t)(2 * l + 1) * h->Hp * h->G4;
            a.xf[l] = h->khx ? h->khx + (size_t)(2 * l) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3) : nullptr;
            a.xb[l] = h->khx ? h->khx + (size_t)(2 * l + 1) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3) : nullptr;
        }
        HIPCK(h, launch_repack_kh_all(s, a, inc));
        if (inc_done) *inc_done = inc != nullptr;
        return FSMG_OK;
    }
    for (int l = 0; l < h->L; ++l) {
        HIPCK(h, launch_repack_kh(s, h->P + h->off_kh[l], h->khf + (size_t)(2 * l) * h->Hp * h->G4,
                                  h->khf + (size_t)(2 * l + 1) * h->Hp * h->G4, h->Hp));
        if (h->khx) HIPCK(h, launch_repack_kh_xcd(s, h->P + h->off_kh[l], h->khx + (size_t)(2 * l) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3),
                                                  h->khx + (size_t)(2 * l + 1) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3), h->Hp, h->xcd_bx3));
    }
    return FSMG_OK;
}
int repack_recurrent_weights(fsmg_model* h, hipStream_t s) { return repack_recurrent_weights(h, s, nullptr, nullptr); }

int apply_update(fsmg_model* h, float grad_scale) {
    ScopedRange rng_("fsmg.clip+adam");
    hipStream_t s = h->stream;
    ScopedTimer tm(h, "update");
    const bool slices = h->cfg.clip_norm_mode == FSMG_CLIP_TF1_SLICES;
    const int64_t skip = slices ? round_up((int64_t)h->V1 * h->Ep, FLAT_ALIGN) : 0;   // embedding is the first segment
    const int64_t n = h->n_flat - skip;
    const int nb = sqnorm_blocks(n);
    HIPCK(h, launch_sqnorm_partials(s, h->G + skip, n, h->partials));
    UpdateArgs a{};
    a.p = h->P; a.m = h->M; a.v = h->Vv; a.g = h->G; a.n = h->n_flat;
    a.partials = h->partials; a.n_partials = nb; a.tail = h->G + h->n_flat; a.use_slices = slices ? 1 : 0;
    a.grad_scale = grad_scale; a.lr = h->cfg.lr; a.n_decay = h->cfg.n_decay; a.clip = h->cfg.max_grad_norm;
    a.step = h->d_step; a.gnorm_out = h->d_gnorm; a.err_flag = h->d_err;
    HIPCK(h, launch_adam_update(s, a));
    // refresh the fragment-ordered recurrent weights and close the step (ring[step] = loss, ++step, or the skip tallies) -- one launch
    StepIncArgs inc{};
    inc.step = h->d_step; inc.loss_src = h->G + h->n_flat + 1; inc.loss_scale = grad_scale; inc.ring = h->d_ring; inc.ring_cap = RING_CAP;
    inc.err_flag = h->d_err; inc.counters = h->d_counters; inc.handoff_dirty = h->d_inbox_dirty; inc.clear_ok = h->last_bwd_xcd ? 1 : 0;
    bool inc_done = false;
    GEMMCK(repack_recurrent_weights(h, s, &inc, &inc_done));
    if (!inc_done) HIPCK(h, launch_step_increment(s, inc));
    PHASE(7);
#ifdef FSMG_PHASE_DEBUG
    phase_report(h);
#endif
    h->have_grads = false;
    return FSMG_OK;
}

// cfg-E inner loop: theta' <- theta' - lr * clip_by_global_norm(grads of the last backward); Adam state and step untouched
int sgd_update(fsmg_model* h, float lr) {
    hipStream_t s = h->stream;
    ScopedTimer tm(h, "update");
    const bool slices = h->cfg.clip_norm_mode == FSMG_CLIP_TF1_SLICES;
    const int64_t skip = slices ? round_up((int64_t)h->V1 * h->Ep, FLAT_ALIGN) : 0;
    const int64_t n = h->n_flat - skip;
    HIPCK(h, launch_sqnorm_partials(s, h->G + skip, n, h->partials));
    UpdateArgs a{};
    a.p = h->P; a.g = h->G; a.n = h->n_flat;
    a.partials = h->partials; a.n_partials = sqnorm_blocks(n); a.tail = h->G + h->n_flat; a.use_slices = slices ? 1 : 0;
    a.lr = lr; a.clip = h->cfg.max_grad_norm; a.gnorm_out = h->d_gnorm; a.err_flag = h->d_err;
    HIPCK(h, launch_sgd_update(s, a));
    h->khf_dirty = true;
    h->have_grads = false;
    return ensure_khf(h);
}

int save_theta(fsmg_model* h) {
    if (!h->P_saved) {
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (hipMalloc((void**)&h->P_saved, sizeof(float) * (size_t)h->n_flat) != hipSuccess)
            return fail(h, FSMG_ERR_NOMEM, "hipMalloc(saved parameters) failed");
    }
    HIPCK(h, hipMemcpyAsync(h->P_saved, h->P, sizeof(float) * (size_t)h->n_flat, hipMemcpyDeviceToDevice, h->stream));
    return FSMG_OK;
}
int restore_theta(fsmg_model* h) {
    HIPCK(h, hipMemcpyAsync(h->P, h->P_saved, sizeof(float) * (size_t)h->n_flat, hipMemcpyDeviceToDevice, h->stream));
    h->khf_dirty = true;
    return ensure_khf(h);
}

// a persistent recurrent kernel gave up waiting for its peers (its blocks were not co-resident): one launch per time step
// for the next `fallback_steps` train steps, then the persistent path is tried again
void on_timeout(fsmg_model* h) {
    ++h->n_timeouts;
    h->persist_timed_out = true;
    if (h->persist) {
        // reached from the asynchronous path too (after_update with loss == NULL polls the host-mapped tallies): later replays
        // of the same execs may still be queued or running, so drain both streams before the execs are destroyed
        h->persist = false;
        hipStreamSynchronize(h->stream);
        if (h->aux) hipStreamSynchronize(h->aux);
        drop_graphs(h);
    }
    h->fallback_left = h->fallback_steps;
    // two launches that must run side by side are one more way to time out (something serialises the dispatches: a counter-collecting
    // profiler, a debugger): a handle that has seen it twice keeps the serial order
    if (h->xov_last && ++h->xov_strikes >= 2 && h->xov) {
        h->xov = false;
        fprintf(stderr, "[fsmg] the XCD-partitioned order timed out twice on this handle (its two launches are not running side by side?): serial order from here on\n");
    }
    // the aborted pass may have left dh partials in the BPTT inboxes and nothing on the device is going to say so on the paths that
    // end without k_step_increment (fsmg_maml_eval's adaptation, a forward-only pass): raise the refill flag from here.  Safe in
    // stream order: the flag is only read by fills of LATER calls.
    if (h->d_inbox_dirty) {
        static const int one = 1;
        hipStreamSynchronize(h->stream);
        hipMemcpy(h->d_inbox_dirty, &one, sizeof(int), hipMemcpyHostToDevice);
    }
    h->tok_table_open = true;           // and the occurrence table may hold entries of a pass whose embed_grad was cut short
}

// Compares the host-mapped tallies of k_step_increment with what this handle has already seen (no synchronisation: the
// caller decides whether the stream has been drained).  0 = nothing new, 2 = a train step was skipped after a time-out,
// 1 = after a token-range error.
int poll_skipped(fsmg_model* h) {
    if (!h->host_counters) return 0;
    const long long to = h->host_counters[0], tk = h->host_counters[1];
    int what = 0;
    const long long pf = h->host_counters[2];
    if (pf != h->seen_peer_failures) { h->seen_peer_failures = pf; what = 3; }
    if (tk != h->seen_token_errors) { h->seen_token_errors = tk; what = 1; }
    if (to != h->seen_timeouts) { h->seen_timeouts = to; on_timeout(h); what = 2; }
    return what;
}

int report(fsmg_model* h, int what) {
    if (what == 2)
        return fail(h, FSMG_ERR_HIP, "persistent recurrent kernel timed out waiting for a peer block (blocks not co-resident); "
                                     "this handle now uses one launch per time step");
    if (what == 1) return fail(h, FSMG_ERR_TOKEN_RANGE, "token id outside [0, input_size)");
    if (what == 3) return fail(h, FSMG_ERR_STATE, "a rank of the episode-parallel job failed before the gradient exchange: the step was skipped on every rank");
    return FSMG_OK;
}

int check_tokens_and_read(fsmg_model* h, const float* d_src, float scale, float* host_out, int n, bool train_tail = false) {
    // one synchronising readback: the loss value(s), then what went wrong.  After a train step k_step_increment has
    // already tallied a skipped step (own or a peer rank's time-out, token-range error) in host-mapped memory and cleared
    // the device flag; a forward-only pass leaves the flag for this function to read and clear.
    std::vector<float> tmp(n);
    int err = 0;
    HIPCK(h, hipMemcpyAsync(tmp.data(), d_src, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    if (!train_tail) HIPCK(h, hipMemcpyAsync(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (train_tail) err = poll_skipped(h);
    else if (err) {
        HIPCK(h, hipMemsetAsync(h->d_err, 0, sizeof(int), h->stream));
        if (err == 2) on_timeout(h);
    }
    if (err) return report(h, err);
    for (int i = 0; i < n; ++i) host_out[i] = tmp[i] * scale;
    return FSMG_OK;
}

int validate_shape(fsmg_model* h, int N, int K, int Q) {
    if (N <= 0 || K < 0 || Q < 0 || (int64_t)N * (K + Q) <= 0 || (int64_t)N * (K + Q) > (1 << 20))
        return fail(h, FSMG_ERR_INVALID, "bad episode shape N/K/Q");
    return FSMG_OK;
}

// host RNG for Glorot init: value depends on (seed, tensor index, logical element index) only
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// forward + backward of one episode whose tokens `stage` puts into the handle's staging buffer ([n_sup + n_qry][T], support rows first)
int apply_update(fsmg_model* h, float grad_scale);
// with_update: the clip + Adam update (grad_scale 1) rides in the same captured graph -- the single-GPU train step; a
// gradient exchange between the two halves (episode-parallel training) needs them as separate calls
template <class Stage>
int forward_backward_core(fsmg_model* h, int32_t N, int32_t K, int32_t Q, Stage&& stage, bool with_update = false) {
    hipSetDevice(h->device);
    int rc = validate_shape(h, N, K, Q);
    if (rc != FSMG_OK) return rc;
    const int B = N * (K + Q);
    if (h->fallback_left > 0 && --h->fallback_left == 0 && h->persist != h->persist_cfg) {   // try the persistent path again
        h->persist = h->persist_cfg;
        drop_graphs(h);
    }
    if ((rc = ensure_scratch(h, B)) != FSMG_OK) return rc;
    choose_schedule(h, B, true);
    h->xov_last = h->xov_call;
    if ((rc = ensure_khf(h)) != FSMG_OK) return rc;
    if (h->tok_table_open && (rc = reset_tok_table(h)) != FSMG_OK) return rc;     // a pass that never reached its embed_grad
    h->tok_table_open = true;
    if ((rc = stage()) != FSMG_OK) return rc;
    const int n_sup = N * K, n_qry = N * Q;
    h->bucket0_recorded = false;
    const std::string shape_key = std::to_string(n_sup) + ":" + std::to_string(n_qry);
    if (h->dp_split && !with_update) {
        // episode-parallel order: bucket 0 (softmax gradients, 56 % of the bytes at cfg-B) is final when the first graph ends and
        // travels while the second one (BPTT, weight / input gradients, embedding gradient) runs
        rc = run_graphed(h, "fb1:" + shape_key, [&]() -> int {
            int r = token_prep(h, n_sup, n_qry, true);
            if (r == FSMG_OK) r = forward(h, B, B, 1, h->G + h->n_flat + 1, true);
            if (r == FSMG_OK) r = backward(h, B, 1);
            return r;
        });
        if (rc != FSMG_OK) return rc;
        if (!h->bucket0_recorded) { HIPCK(h, hipEventRecord(h->ev_bucket[0], h->stream)); h->bucket0_recorded = true; }
        rc = run_graphed(h, "fb2:" + shape_key, [&]() -> int { return backward(h, B, 2); });
    } else {
        rc = run_graphed(h, (with_update ? "fbu:" : "fb:") + shape_key, [&]() -> int {
            int r = token_prep(h, n_sup, n_qry, true);
            if (r == FSMG_OK) r = forward(h, B, B, 1, h->G + h->n_flat + 1, true);
            if (r == FSMG_OK) r = backward(h, B);
            if (r == FSMG_OK && with_update) r = apply_update(h, 1.0f);
            return r;
        });
    }
    if (rc != FSMG_OK) return rc;
    h->tok_table_open = false;          // (a replayed graph ran its embed_grad too)
    // bucket readiness for an overlapped gradient exchange: with the two-stream (eager) schedule bucket 0 was
    // recorded right behind the dW GEMM on the aux stream; a replayed graph finishes as a whole
    // (the fused single-GPU step has applied its update already: nobody waits for a bucket, and two event records between
    // consecutive steps are ~10 us of queue time)
    if (!with_update) {
        if (!h->bucket0_recorded) HIPCK(h, hipEventRecord(h->ev_bucket[0], h->stream));
        HIPCK(h, hipEventRecord(h->ev_bucket[1], h->stream));
    }
    h->lastB = B;
    h->have_grads = !with_update;
    return FSMG_OK;
}

int after_update(fsmg_model* h, float grad_scale, float* loss) {
    h->have_grads = false;
    if (loss) return check_tokens_and_read(h, h->G + h->n_flat + 1, grad_scale, loss, 1, true);
    // no read-back: skipped steps of EARLIER calls that have retired by now are noticed here (a time-out switches the
    // handle to per-step launches; the skipped episodes stay skipped -- fsmg_get_stats counts them)
    const int what = poll_skipped(h);
    if (what == 1 || what == 3) return report(h, what);
    return FSMG_OK;
}

// ---- RCCL, looked up at run time (the library has no link-time dependency on it; a process that already loaded torch's
// librccl.so gets that one)
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
    bool ok = false;
    Rccl() {
        static const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* lib = nullptr;
        for (const char* name : names) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy already in the process first
        for (const char* name : names) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { why = "librccl.so not found (dlopen)"; return; }
        GetUniqueId = (int (*)(UniqueId*))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, UniqueId, int))dlsym(lib, "ncclCommInitRank");
        CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
        Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclBroadcast");
        GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && Broadcast && GroupStart && GroupEnd && GetErrorString;
        if (!ok) why = "librccl.so lacks an expected symbol";
    }
};
inline Rccl& rccl() { static Rccl r; return r; }
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0, NCCL_CHAR = 0;
#define NCCLCK(h, call)                                                                                   \
    do {                                                                                                  \
        const int e_ = (call);                                                                            \
        if (e_ != 0) return fail(h, FSMG_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(e_)); \
    } while (0)

// sum of the gradient buffer over the ranks: three buckets on the communication stream, each behind its readiness event
// (bucket 0 = softmax gradients: final behind the projection-gradient GEMMs when the backward pass is cut there); the compute
// stream (not the host) then waits for the communication stream
int exchange_gradients(fsmg_model* h) {
    Rccl& r = rccl();
    HIPCK(h, hipStreamWaitEvent(h->comm_stream, h->ev_bucket[0], 0));
    NCCLCK(h, r.AllReduce(h->G + h->off_w, h->G + h->off_w, (size_t)(h->n_flat - h->off_w), NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    HIPCK(h, hipStreamWaitEvent(h->comm_stream, h->ev_bucket[1], 0));
    NCCLCK(h, r.GroupStart());
    NCCLCK(h, r.AllReduce(h->G, h->G, (size_t)h->off_w, NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    NCCLCK(h, r.AllReduce(h->G + h->n_flat, h->G + h->n_flat, (size_t)FSMG_GRAD_TAIL, NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    NCCLCK(h, r.GroupEnd());
    HIPCK(h, hipEventRecord(h->ev_comm, h->comm_stream));
    HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
    return FSMG_OK;
}

int after_update(fsmg_model* h, float grad_scale, float* loss);
// the episode-parallel step with the exchange inside the library: forward + backward, all-reduce, clip + Adam (1 / world)
template <class FB>
int dp_train_step(fsmg_model* h, float* loss, FB&& forward_backward) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc = forward_backward();
        int local_rc = FSMG_OK; std::string local_msg;
        if (rc != FSMG_OK) {
            // A failure on THIS rank's host (an allocation, a launch) must not leave the peers blocked in ncclAllReduce: join the
            // collectives with the "this rank's gradients are garbage" indicator raised (tail[4]; summed like the time-out and
            // token-range indicators), so that every rank skips the update and every rank's step ends -- then report the failure.
            local_rc = rc; local_msg = h->err;
            if (launch_fill32(h->stream, h->G + h->n_flat + 4, 0x3f800000u, 1) != hipSuccess) return local_rc;     // 1.0f
            if (hipEventRecord(h->ev_bucket[0], h->stream) != hipSuccess || hipEventRecord(h->ev_bucket[1], h->stream) != hipSuccess) return local_rc;
            h->have_grads = true;
        }
        rc = exchange_gradients(h);
        const float scale = 1.0f / (float)h->world;
        if (rc == FSMG_OK) {
            uint32_t bits; std::memcpy(&bits, &scale, 4);
            rc = run_graphed(h, "up:" + std::to_string(bits) + (h->last_bwd_xcd ? "x" : "s"), [&]() -> int { return apply_update(h, scale); });
        }
        if (rc == FSMG_OK) rc = after_update(h, scale, loss);
        if (local_rc != FSMG_OK) { h->err = local_msg; return local_rc; }
        // a time-out on ANY rank travelled in the reduced tail: every rank skipped the update, reports it here and repeats the
        // step on per-step launches, in lock-step
        if (rc == FSMG_ERR_HIP && h->persist_timed_out && attempt == 0) { h->persist_timed_out = false; continue; }
        return rc;
    }
    return FSMG_OK;
}

// forward + backward + update as ONE captured graph (17 us between two graph launches at cfg-B otherwise)
template <class Stage>
int fused_train_step(fsmg_model* h, int32_t N, int32_t K, int32_t Q, float* loss, Stage&& stage) {
    if (h->comm != nullptr) return dp_train_step(h, loss, [&]() { return forward_backward_core(h, N, K, Q, stage, false); });
    int rc = forward_backward_core(h, N, K, Q, stage, true);
    if (rc == FSMG_OK) rc = after_update(h, 1.0f, loss);
    if (rc == FSMG_ERR_HIP && h->persist_timed_out) {
        // a persistent step kernel could not get all of its blocks resident (another workload holds the CUs): the
        // update kernels saw the flag and left parameters, Adam state and step counter alone, and the handle has
        // fallen back to one launch per time step -- repeat the step that way
        h->persist_timed_out = false;
        rc = forward_backward_core(h, N, K, Q, stage, true);
        if (rc == FSMG_OK) rc = after_update(h, 1.0f, loss);
    }
    return rc;
}

}  // namespace

// =========================================================================== C ABI
extern "C" {

int fsmg_version(void) { return FSMG_VERSION; }

const char* fsmg_last_error(fsmg_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

uint64_t fsmg_state_bytes(const fsmg_config* cfg) {
    if (!cfg) return 0;
    fsmg_model m;
    compute_dims(*cfg, &m);
    return (uint64_t)state_bytes_for(build_layout(&m));
}

int fsmg_create(const fsmg_config* cfg, fsmg_handle* out) {
    if (!cfg || !out) return fail(nullptr, FSMG_ERR_INVALID, "null config/out");
    *out = nullptr;
    if (cfg->config_version != FSMG_CONFIG_VERSION)
        return fail(nullptr, FSMG_ERR_INVALID, "fsmg_config.config_version is " + std::to_string(cfg->config_version) + ", this library expects " +
                                                   std::to_string(FSMG_CONFIG_VERSION) + " (caller built against another include/fsmg.h)");
    if (cfg->gemm < 0 || cfg->gemm > FSMG_GEMM_F32 || cfg->schedule < 0 || cfg->schedule > FSMG_SCHEDULE_XCD_PARTITIONED ||
        cfg->recurrence < 0 || cfg->recurrence > FSMG_RECURRENCE_XCD_LOCAL)
        return fail(nullptr, FSMG_ERR_INVALID, "fsmg_config.gemm / schedule / recurrence out of range");
    if (cfg->input_size <= 0 || cfg->max_len <= 0 || cfg->embedding_size <= 0 || cfg->hidden_size <= 0 ||
        cfg->n_layers <= 0 || cfg->n_layers > 16 || cfg->embedding_size > 1024 || !(cfg->n_decay > 0.f) ||
        !(cfg->max_grad_norm > 0.f))
        return fail(nullptr, FSMG_ERR_INVALID, "config out of range (sizes must be > 0, embedding_size <= 1024, n_layers <= 16)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FSMG_ERR_NO_DEVICE, "no HIP device visible: libfsmg has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, FSMG_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipSetDevice(cfg->device) != hipSuccess || hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, FSMG_ERR_NO_DEVICE, "cannot select HIP device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FSMG_ERR_NO_DEVICE, std::string("libfsmg is built for gfx950 only, device is ") + prop.gcnArchName);

    fsmg_model* h = new (std::nothrow) fsmg_model();
    if (!h) return fail(nullptr, FSMG_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->device = cfg->device;
    compute_dims(*cfg, h);
    h->n_flat = build_layout(h);
    auto bail = [&](int code, const std::string& msg) { g_create_error = msg; fsmg_destroy(h); return code; };

    if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(FSMG_ERR_HIP, "hipStreamCreate failed");
        h->own_stream = true;
    }
    {
        if (const char* eg = std::getenv("FSMG_GRAPH")) h->cfg.use_graph = (eg[0] != '0');   // debugging override
        // two-stream schedule: pays when the vocabulary projection dominates the recurrence (measured: +13 % at
        // cfg-B/D where V1/(4H*L) = 4.9; -7 % at cfg-C where it is 0.6), so by default it is chosen from the
        // shapes; FSMG_OVERLAP=0/1 forces the single-stream (hipGraph-replayed) / two-stream (eager) order
        // the configuration first, the environment (debugging overrides) on top of it
        h->overlap = (int64_t)h->V1 >= 8LL * h->H * h->L;
        if (cfg->schedule == FSMG_SCHEDULE_SINGLE_STREAM) { h->overlap = false; h->overlap_forced = true; }
        if (cfg->schedule == FSMG_SCHEDULE_TWO_STREAM) { h->overlap = true; h->overlap_forced = true; }
        if (cfg->schedule == FSMG_SCHEDULE_XCD_PARTITIONED) h->xov = true;
        if (cfg->gemm == FSMG_GEMM_F32) h->bx3 = 0;
        if (cfg->recurrence == FSMG_RECURRENCE_PER_STEP) h->persist = false;
        if (cfg->recurrence == FSMG_RECURRENCE_COLUMN_SPLIT) h->xcd = false;
        if (cfg->recurrence == FSMG_RECURRENCE_XCD_LOCAL) h->pair_mode = 2;
        if (cfg->dp_split_backward) h->dp_split = cfg->dp_split_backward == 2 ? 2 : 1;
        const char* env = std::getenv("FSMG_OVERLAP");
        if (env) { h->overlap = env[0] != '0'; h->overlap_forced = true; }
        if (const char* e = std::getenv("FSMG_GEMM")) h->bx3 = std::strcmp(e, "f32") != 0;
        if (const char* e = std::getenv("FSMG_XCD_OVERLAP")) h->xov = std::atoi(e) != 0;
        if (const char* e = std::getenv("FSMG_XOV_PARTS")) h->xov_parts = std::max(1, std::min(3, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_EAGER")) h->eager = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_MERGE_DK")) h->merge_dk = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_PERSISTENT")) h->persist = (e[0] != '0');
        h->persist_cfg = h->persist;
        if (const char* e = std::getenv("FSMG_FALLBACK_STEPS")) h->fallback_steps = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XCD")) h->xcd = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_XCD_PAIR")) h->pair_mode = std::max(0, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_FWD_RT")) h->force_fwd_rt = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_CHAIN_SPIN_LIMIT")) h->chain_spin_limit = std::max(0, std::atoi(e));
#ifdef FSMG_EXPERIMENTS         // settled A/Bs: tuning values and rejected alternatives, experiment builds only
        if (const char* e = std::getenv("FSMG_XOV_DW_SPLIT")) h->xov_dw_split = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XOV_PUB")) h->xov_pub = std::max(1, std::min(64, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_FILL_EARLY")) h->fill_early = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_FILLS_LATE")) h->fills_late = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_XOV_TAIL")) h->xov_tail = std::max(0, std::min(64, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_BWD_RS")) h->bwd_rs = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_DP_SPLIT")) h->dp_split = std::max(0, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_XCD_VARIANT")) h->xcd_variant = std::atoi(e);
        if (const char* e = std::getenv("FSMG_XCD_MAX_ROWS")) h->xcd_max_rows = std::max(1, std::min(128, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_PERSIST_FWD")) h->persist_fwd = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_PERSIST_BWD")) h->persist_bwd = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_NCHUNK")) h->nchunk = h->nchunk_persist = std::max(1, std::min((int)fsmg_model::NCHUNK, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_CHUNK_STEPS")) {      // e.g. "12,36,34,34,12": must add up to max_len
            std::vector<int> edges{0};
            for (const char* p = e; *p;) { edges.push_back(edges.back() + std::max(1, std::atoi(p))); while (*p && *p != ',') ++p; if (*p) ++p; }
            if (edges.back() == h->T && (int)edges.size() - 1 <= (int)fsmg_model::NCHUNK) {
                h->chunk_edges = edges;
                h->nchunk = h->nchunk_persist = (int)edges.size() - 1;
            }
        }
        if (const char* e = std::getenv("FSMG_AUX_BLOCKS")) { h->aux_blocks_per_cu = h->aux_blocks_persist = std::max(1, std::min(4, std::atoi(e))); h->aux_blocks_from_env = true; }
#endif
        int least = 0, greatest = 0;
        hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, least) != hipSuccess) return bail(FSMG_ERR_HIP, "aux stream create failed");
        for (int c = 0; c < fsmg_model::NCHUNK; ++c)
            if (hipEventCreateWithFlags(&h->ev_chunk[c], hipEventDisableTiming) != hipSuccess) return bail(FSMG_ERR_HIP, "event create failed");
        if (hipEventCreateWithFlags(&h->ev_bucket[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_bucket[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) return bail(FSMG_ERR_HIP, "event create failed");
    }
    const int64_t sb = state_bytes_for(h->n_flat);
    if (cfg->state_arena) {
        if (cfg->state_arena_bytes < (uint64_t)sb || ((uintptr_t)cfg->state_arena & 255u))
            return bail(FSMG_ERR_INVALID, "state_arena too small or not 256-byte aligned");
        h->state = (char*)cfg->state_arena; h->own_state = false;
    } else {
        if (hipMalloc((void**)&h->state, sb) != hipSuccess) return bail(FSMG_ERR_NOMEM, "hipMalloc(state) failed");
        h->own_state = true;
    }
    h->P = (float*)h->state; h->G = h->P + h->n_flat;          // G has n_flat + FSMG_GRAD_TAIL floats
    h->M = h->G + h->n_flat + FSMG_GRAD_TAIL; h->Vv = h->M + h->n_flat;
    if (hipMemsetAsync(h->state, 0, sb, h->stream) != hipSuccess) return bail(FSMG_ERR_HIP, "memset(state) failed");

    char* small = nullptr;
    const size_t tok_words = (size_t)round_up(h->V1, 64);
    const size_t prog_words = (size_t)round_up(h->T + 8, 64);
    const size_t small_bytes = 256 * 4 + sizeof(float) * RING_CAP + sizeof(int) * 8 * fsmg_model::TICKET_LAUNCHES + sizeof(int) * 2 * fsmg_model::XOV_CTL +
                               sizeof(int) * 2 * tok_words + sizeof(int) * prog_words;
    if (hipMalloc((void**)&small, small_bytes) != hipSuccess) return bail(FSMG_ERR_NOMEM, "hipMalloc(scalars) failed");
    hipMemsetAsync(small, 0, small_bytes, h->stream);
    h->d_step = (long long*)small; h->d_err = (int*)(small + 256); h->d_gnorm = (float*)(small + 512);
    h->d_inbox_dirty = (int*)(small + 768);
    { static const int one = 1; hipStreamSynchronize(h->stream); hipMemcpy(h->d_inbox_dirty, &one, sizeof(int), hipMemcpyHostToDevice); }
    h->d_ring = (float*)(small + 1024);
    if (hipHostMalloc((void**)&h->host_counters, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->d_counters, h->host_counters, 0) != hipSuccess)
        return bail(FSMG_ERR_NOMEM, "hipHostMalloc(mapped step counters) failed");
    std::memset(h->host_counters, 0, 64);
    h->tickets = (int*)(small + 1024 + sizeof(float) * RING_CAP);
    h->xov_ctl = h->tickets + 8 * fsmg_model::TICKET_LAUNCHES;
    h->tok_first = h->xov_ctl + 2 * fsmg_model::XOV_CTL; h->tok_count = h->tok_first + tok_words;
    h->xov_prog = h->tok_count + tok_words;
    if (reset_tok_table(h) != FSMG_OK) return bail(FSMG_ERR_HIP, "fill of the token occurrence table failed");

    // decode scratch: per layer h ping/pong + c, plus x and argmax block scratch
    {
        const size_t nblk = (h->V1 + 255) / 256;
        const size_t fl = (size_t)h->L * 3 * h->Hp + 2 * nblk + 64;
        if (hipMalloc((void**)&h->dec, sizeof(float) * fl + 256) != hipSuccess) return bail(FSMG_ERR_NOMEM, "hipMalloc(decode) failed");
    }
    if (hipMalloc((void**)&h->khf, sizeof(float) * (size_t)h->L * 2 * h->Hp * h->G4) != hipSuccess)
        return bail(FSMG_ERR_NOMEM, "hipMalloc(fragment weights) failed");
    if (h->persist && h->xcd && lstm_xcd_supported(1, h->Hp)) {
        // bf16-split XCD-local kernels where the episode the handle is created for has the rows that make them the faster ones
        // (cfg-D: 100 sequences); FSMG_XCD_BX3=0/1 forces.  One format per handle: weight images and hand-off buffer follow it.
        h->xcd_bx3 = lstm_xcd_bx3_pays(cfg->max_sequences > 0 ? cfg->max_sequences : 45, h->Hp) && (cfg->max_sequences <= h->xcd_max_rows);
        // AUTO schedule: the XCD-partitioned order where it was measured to pay (cfg-B: +8 % against the serial order, same bits as
        // the serial order on the same kernels) -- one 512-unit layer, the episode's rows on at most five XCDs (16 per XCD), and a
        // projection with enough 256 x 256 tiles to keep the other XCDs busy for the length of a chain
        {
            const int b0 = cfg->max_sequences > 0 ? cfg->max_sequences : 45;
            const int rpx = lstm_xcd16_packed_rows(b0);
            const long long items = (((long long)h->T * b0 + 255) / 256) * ((h->V1p + 255) / 256);
            const bool eligible = h->bx3 && h->Hp == 512 && h->L == 1 && rpx > 0 && (b0 + rpx - 1) / rpx <= 5 && b0 >= 16 && h->T >= 32 && items >= 320 &&
                                  4 + items <= fsmg_model::XOV_CTL;
            if (cfg->schedule == FSMG_SCHEDULE_AUTO && std::getenv("FSMG_XCD_OVERLAP") == nullptr && !h->overlap_forced) h->xov = eligible;
        }
        // the XCD-partitioned schedule packs the rows on ceil(B / 16) XCDs: only the bf16-split kernels take 16 rows per XCD at one
        // MFMA phase's cost
        if (h->xov && h->bx3 && h->Hp == 512) h->xcd_bx3 = true;
        if (const char* e = std::getenv("FSMG_XCD_BX3")) h->xcd_bx3 = std::atoi(e) != 0 && h->Hp == 512;
        if (hipMalloc((void**)&h->khx, sizeof(float) * (size_t)h->L * 2 * lstm_xcd_weight_floats(h->Hp, h->xcd_bx3)) != hipSuccess)
            return bail(FSMG_ERR_NOMEM, "hipMalloc(XCD-local weight images) failed");
    }
    const int b0 = cfg->max_sequences > 0 ? cfg->max_sequences : 45;
    if (ensure_scratch(h, b0) != FSMG_OK) { std::string e = h->err; return bail(FSMG_ERR_NOMEM, e); }
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(FSMG_ERR_HIP, "stream sync failed");
    *out = h;
    return FSMG_OK;
}

int fsmg_destroy(fsmg_handle h) {
    if (!h) return FSMG_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->aux) hipStreamSynchronize(h->aux);
    drain_timers(h);
    drop_graphs(h);
    if (h->scratch) hipFree(h->scratch);
    if (h->d_step) hipFree(h->d_step);
    if (h->dec) hipFree(h->dec);
    if (h->khf) hipFree(h->khf);
    if (h->khx) hipFree(h->khx);
    if (h->P_saved) hipFree(h->P_saved);
    for (int* t : h->table) if (t) hipFree(t);
    if (h->d_idx) hipFree(h->d_idx);
    if (h->d_gather) hipFree(h->d_gather);
    if (h->d_eval) hipFree(h->d_eval);
    if (h->host_counters) hipHostFree(h->host_counters);
    if (h->own_state && h->state) hipFree(h->state);
    for (int c = 0; c < fsmg_model::NCHUNK; ++c) if (h->ev_chunk[c]) hipEventDestroy(h->ev_chunk[c]);
    if (h->ev_bucket[0]) hipEventDestroy(h->ev_bucket[0]);
    if (h->ev_bucket[1]) hipEventDestroy(h->ev_bucket[1]);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->comm && h->own_comm && rccl().ok) rccl().CommDestroy(h->comm);
    if (h->ev_comm) hipEventDestroy(h->ev_comm);
    if (h->comm_stream) hipStreamDestroy(h->comm_stream);
    if (h->probe) { hipStreamSynchronize(h->probe); hipStreamDestroy(h->probe); }
    if (h->d_probe) hipFree(h->d_probe);
    if (h->aux) hipStreamDestroy(h->aux);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return FSMG_OK;
}

int fsmg_synchronize(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    HIPCK(h, hipStreamSynchronize(h->stream));
    return FSMG_OK;
}

int fsmg_init_params(fsmg_handle h, uint64_t seed) {
    if (!h) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    int idx = 0;
    for (auto& p : h->params) {
        const int64_t n = p.rows * p.cols;
        std::vector<float> ref(n, 0.0f);
        if (p.kind != 2) {                   // LSTM biases start at zero
            const double fan_in = (double)p.rows, fan_out = p.cols == 1 ? (double)p.rows : (double)p.cols;
            const float limit = (float)std::sqrt(6.0 / (fan_in + fan_out));
            const uint64_t base = splitmix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(idx + 1)));
            for (int64_t i = 0; i < n; ++i) {
                const uint64_t r = splitmix64(base + (uint64_t)i);
                const float u = (float)(((r >> 40) + 0.5) * (1.0 / 16777216.0));
                ref[i] = (2.0f * u - 1.0f) * limit;
            }
        }
        int rc = upload_tensor(h, h->P, p.name.c_str(), ref.data(), n);
        if (rc != FSMG_OK) return rc;
        ++idx;
    }
    HIPCK(h, hipMemsetAsync(h->M, 0, sizeof(float) * (size_t)h->n_flat, h->stream));
    HIPCK(h, hipMemsetAsync(h->Vv, 0, sizeof(float) * (size_t)h->n_flat, h->stream));
    HIPCK(h, hipMemsetAsync(h->d_step, 0, sizeof(long long), h->stream));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return FSMG_OK;
}

int fsmg_num_params(fsmg_handle h) { return h ? (int)h->params.size() : FSMG_ERR_INVALID; }

int fsmg_param_info(fsmg_handle h, int idx, char* name, int name_cap, int64_t* rows, int64_t* cols) {
    if (!h || idx < 0 || idx >= (int)h->params.size()) return FSMG_ERR_INVALID;
    const ParamDesc& p = h->params[idx];
    if (name && name_cap > 0) { std::strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (rows) *rows = p.rows;
    if (cols) *cols = p.cols;
    return FSMG_OK;
}

int fsmg_set_param(fsmg_handle h, const char* name, const float* host, int64_t count) {
    if (!h || !name || !host) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    return upload_tensor(h, h->P, name, host, count);
}
int fsmg_get_param(fsmg_handle h, const char* name, float* host, int64_t count) {
    if (!h || !name || !host) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    return download_tensor(h, h->P, name, host, count);
}
int fsmg_set_opt_state(fsmg_handle h, const char* name, const float* m, const float* v, int64_t count) {
    if (!h || !name || !m || !v) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    int rc = upload_tensor(h, h->M, name, m, count);
    return rc != FSMG_OK ? rc : upload_tensor(h, h->Vv, name, v, count);
}
int fsmg_get_opt_state(fsmg_handle h, const char* name, float* m, float* v, int64_t count) {
    if (!h || !name || !m || !v) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    int rc = download_tensor(h, h->M, name, m, count);
    return rc != FSMG_OK ? rc : download_tensor(h, h->Vv, name, v, count);
}
int fsmg_set_step(fsmg_handle h, int64_t global_step) {
    if (!h || global_step < 0) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    long long v = global_step;
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(h->d_step, &v, sizeof(v), hipMemcpyHostToDevice));
    return FSMG_OK;
}
int fsmg_get_step(fsmg_handle h, int64_t* global_step) {
    if (!h || !global_step) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    long long v = 0;
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(&v, h->d_step, sizeof(v), hipMemcpyDeviceToHost));
    poll_skipped(h);
    *global_step = v;
    return FSMG_OK;
}
int fsmg_get_grad(fsmg_handle h, const char* name, float* host, int64_t count) {
    if (!h || !name || !host) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    return download_tensor(h, h->G, name, host, count);
}

int fsmg_forward_backward(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K,
                          int32_t Q, int32_t tokens_on_device) {
    if (!h || !support || !query) return FSMG_ERR_INVALID;
    return forward_backward_core(h, N, K, Q, [&]() { return stage_tokens(h, support, N * K, query, N * Q, tokens_on_device); });
}

// ---- device-resident episode table: a split's packed [n_songs][T] token table lives in HBM and an
// episode is an index gather on the GPU (reference src/data/episode.py:62-74, src/data/dataset.py:187-199 fill the same
// rows from the host cache): a step uploads N*(K+Q) indices (180 B at cfg-B) instead of 23 KB of tokens.
int fsmg_upload_table(fsmg_handle h, int32_t table_id, const int32_t* host_table, int64_t n_songs) {
    if (!h || !host_table || table_id < 0 || table_id >= fsmg_model::MAX_TABLES || n_songs <= 0 || n_songs > (1LL << 30) / std::max(1, h->T))
        return h ? fail(h, FSMG_ERR_INVALID, "bad table id / size") : FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->table[table_id]) { hipFree(h->table[table_id]); h->table[table_id] = nullptr; h->table_rows[table_id] = 0; }
    const size_t bytes = sizeof(int) * (size_t)n_songs * h->T;
    if (hipMalloc((void**)&h->table[table_id], bytes) != hipSuccess) return fail(h, FSMG_ERR_NOMEM, "hipMalloc(token table) failed");
    HIPCK(h, hipMemcpy(h->table[table_id], host_table, bytes, hipMemcpyHostToDevice));
    h->table_rows[table_id] = n_songs;
    return FSMG_OK;
}

static int stage_indexed(fsmg_handle h, int32_t table_id, const int32_t* sup_idx, int n_sup, const int32_t* qry_idx, int n_qry) {
    if (table_id < 0 || table_id >= fsmg_model::MAX_TABLES || !h->table[table_id]) return fail(h, FSMG_ERR_STATE, "no token table uploaded under this id");
    const int n = n_sup + n_qry;
    if (h->idx_cap < n) {
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (h->d_idx) hipFree(h->d_idx);
        h->idx_cap = std::max(n, 4096);
        if (hipMalloc((void**)&h->d_idx, sizeof(int) * h->idx_cap) != hipSuccess) { h->idx_cap = 0; h->d_idx = nullptr; return fail(h, FSMG_ERR_NOMEM, "hipMalloc(indices) failed"); }
    }
    if (n_sup > 0) HIPCK(h, hipMemcpyAsync(h->d_idx, sup_idx, sizeof(int) * n_sup, hipMemcpyHostToDevice, h->stream));
    if (n_qry > 0) HIPCK(h, hipMemcpyAsync(h->d_idx + n_sup, qry_idx, sizeof(int) * n_qry, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, launch_gather_rows(h->stream, h->table[table_id], h->d_idx, n, h->T, (int)h->table_rows[table_id], h->d_tok, h->d_err));
    h->cur_sup = h->d_tok; h->cur_qry = h->d_tok + (size_t)n_sup * h->T;
    return FSMG_OK;
}

int fsmg_forward_backward_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                                  int32_t N, int32_t K, int32_t Q) {
    if (!h || !support_idx || !query_idx) return FSMG_ERR_INVALID;
    return forward_backward_core(h, N, K, Q, [&]() { return stage_indexed(h, table_id, support_idx, N * K, query_idx, N * Q); });
}

int fsmg_train_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                            int32_t N, int32_t K, int32_t Q, float* loss) {
    if (!h || !support_idx || !query_idx) return FSMG_ERR_INVALID;
    return fused_train_step(h, N, K, Q, loss, [&]() { return stage_indexed(h, table_id, support_idx, N * K, query_idx, N * Q); });
}

int fsmg_grad_buffer(fsmg_handle h, void** device_ptr, int64_t* count) {
    if (!h || !device_ptr || !count) return FSMG_ERR_INVALID;
    *device_ptr = h->G;
    *count = h->n_flat + FSMG_GRAD_TAIL;
    return FSMG_OK;
}

int fsmg_grad_bucket(fsmg_handle h, int32_t bucket, void** device_ptr, int64_t* count) {
    if (!h || !device_ptr || !count || bucket < 0 || bucket >= FSMG_NUM_BUCKETS) return FSMG_ERR_INVALID;
    if (bucket == 0) { *device_ptr = h->G + h->off_w; *count = h->n_flat - h->off_w; }
    else if (bucket == 1) { *device_ptr = h->G; *count = h->off_w; }
    else { *device_ptr = h->G + h->n_flat; *count = FSMG_GRAD_TAIL; }
    return FSMG_OK;
}

int fsmg_stream_wait_bucket(fsmg_handle h, void* stream, int32_t bucket) {
    if (!h || bucket < 0 || bucket >= FSMG_NUM_BUCKETS) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    if (!h->have_grads) return fail(h, FSMG_ERR_STATE, "no backward pass is pending");
    HIPCK(h, hipStreamWaitEvent((hipStream_t)stream, h->ev_bucket[bucket == 0 ? 0 : 1], 0));
    return FSMG_OK;
}

int fsmg_apply_update(fsmg_handle h, float grad_scale, float* loss) {
    if (!h) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    if (!h->have_grads) return fail(h, FSMG_ERR_STATE, "fsmg_apply_update without a preceding fsmg_forward_backward");
    if (!(grad_scale > 0.f)) return fail(h, FSMG_ERR_INVALID, "grad_scale must be > 0");
    uint32_t bits; std::memcpy(&bits, &grad_scale, 4);
    int rc = run_graphed(h, "up:" + std::to_string(bits) + (h->last_bwd_xcd ? "x" : "s"), [&]() -> int { return apply_update(h, grad_scale); });
    if (rc != FSMG_OK) return rc;
    return after_update(h, grad_scale, loss);
}

// ---- the gradient exchange inside the library
static int comm_prepare(fsmg_handle h) {
    if (!rccl().ok) return fail(h, FSMG_ERR_STATE, "RCCL is not available: " + rccl().why);
    hipSetDevice(h->device);
    if (!h->comm_stream) HIPCK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (!h->ev_comm) HIPCK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
    return FSMG_OK;
}

int fsmg_comm_unique_id(char id[FSMG_COMM_ID_BYTES]) {
    if (!id) return FSMG_ERR_INVALID;
    if (!rccl().ok) return fail(nullptr, FSMG_ERR_STATE, "RCCL is not available: " + rccl().why);
    Rccl::UniqueId u;
    const int e = rccl().GetUniqueId(&u);
    if (e != 0) return fail(nullptr, FSMG_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(e));
    std::memcpy(id, u.internal, FSMG_COMM_ID_BYTES);
    return FSMG_OK;
}

int fsmg_comm_init(fsmg_handle h, const char id[FSMG_COMM_ID_BYTES], int32_t world_size, int32_t rank) {
    if (!h || !id || world_size < 1 || rank < 0 || rank >= world_size) return FSMG_ERR_INVALID;
    if (h->comm) return fail(h, FSMG_ERR_STATE, "a communicator is already attached");
    int rc = comm_prepare(h);
    if (rc != FSMG_OK) return rc;
    Rccl::UniqueId u;
    std::memcpy(u.internal, id, FSMG_COMM_ID_BYTES);
    void* c = nullptr;
    NCCLCK(h, rccl().CommInitRank(&c, world_size, u, rank));
    h->comm = c; h->own_comm = true; h->world = world_size; h->rank = rank;
    drop_graphs(h);
    return FSMG_OK;
}

int fsmg_comm_attach(fsmg_handle h, void* nccl_comm, int32_t world_size, int32_t rank) {
    if (!h || !nccl_comm || world_size < 1 || rank < 0 || rank >= world_size) return FSMG_ERR_INVALID;
    if (h->comm) return fail(h, FSMG_ERR_STATE, "a communicator is already attached");
    int rc = comm_prepare(h);
    if (rc != FSMG_OK) return rc;
    h->comm = nccl_comm; h->own_comm = false; h->world = world_size; h->rank = rank;
    drop_graphs(h);
    return FSMG_OK;
}

int fsmg_comm_broadcast_state(fsmg_handle h, int32_t root) {
    if (!h) return FSMG_ERR_INVALID;
    if (!h->comm || root < 0 || root >= h->world) return fail(h, FSMG_ERR_STATE, "no communicator attached / bad root");
    hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    NCCLCK(h, rccl().GroupStart());
    NCCLCK(h, rccl().Broadcast(h->P, h->P, (size_t)h->n_flat, NCCL_FLOAT, root, h->comm, h->comm_stream));
    NCCLCK(h, rccl().Broadcast(h->M, h->M, (size_t)2 * h->n_flat, NCCL_FLOAT, root, h->comm, h->comm_stream));     // m and v are adjacent
    NCCLCK(h, rccl().Broadcast(h->d_step, h->d_step, sizeof(long long), NCCL_CHAR, root, h->comm, h->comm_stream));
    NCCLCK(h, rccl().GroupEnd());
    HIPCK(h, hipStreamSynchronize(h->comm_stream));
    h->khf_dirty = true;
    return FSMG_OK;
}

int fsmg_comm_release(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCK(h, hipStreamSynchronize(h->comm_stream));
    if (h->comm && h->own_comm && rccl().ok) rccl().CommDestroy(h->comm);
    h->comm = nullptr; h->own_comm = false; h->world = 1; h->rank = 0;
    drop_graphs(h);
    return FSMG_OK;
}

int fsmg_train_step(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                    int32_t tokens_on_device, float* loss) {
    if (!h || !support || !query) return FSMG_ERR_INVALID;
    return fused_train_step(h, N, K, Q, loss, [&]() { return stage_tokens(h, support, N * K, query, N * Q, tokens_on_device); });
}

// ---- cfg-E (BASELINE.json configs[4]): MAML-style inner / outer loop, first order.  DESIGN.md "cfg-E".
static int maml_adapt(fsmg_handle h, const int32_t* support, int32_t N, int32_t K, int32_t inner_steps, float inner_lr, int32_t on_device) {
    int rc = save_theta(h);
    for (int i = 0; rc == FSMG_OK && i < inner_steps; ++i) {
        rc = fsmg_forward_backward(h, support, support, N, K, 0, on_device);       // support rows only
        if (rc == FSMG_OK) rc = sgd_update(h, inner_lr);
    }
    return rc;
}

int fsmg_maml_forward_backward(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                               int32_t inner_steps, float inner_lr, int32_t tokens_on_device) {
    if (!h || !support || !query) return FSMG_ERR_INVALID;
    if (inner_steps < 0 || inner_steps > 64 || !(inner_lr >= 0.f) || K <= 0 || Q <= 0) return fail(h, FSMG_ERR_INVALID, "bad inner_steps / inner_lr / K / Q");
    hipSetDevice(h->device);
    int rc = maml_adapt(h, support, N, K, inner_steps, inner_lr, tokens_on_device);
    if (rc == FSMG_OK) rc = fsmg_forward_backward(h, query, query, N, Q, 0, tokens_on_device);     // query rows at theta'
    const int rc2 = h->P_saved ? restore_theta(h) : FSMG_OK;                                         // theta comes back whatever happened
    return rc != FSMG_OK ? rc : rc2;
}

int fsmg_maml_step(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* loss) {
    if (h && h->comm != nullptr)           // per-rank inner loop (no communication), query gradients exchanged like a plain step's
        return dp_train_step(h, loss, [&]() { return fsmg_maml_forward_backward(h, support, query, N, K, Q, inner_steps, inner_lr, tokens_on_device); });
    int rc = fsmg_maml_forward_backward(h, support, query, N, K, Q, inner_steps, inner_lr, tokens_on_device);
    if (rc != FSMG_OK) return rc;
    rc = fsmg_apply_update(h, 1.0f, loss);
    if (rc == FSMG_ERR_HIP && h->persist_timed_out) {        // same recovery as fsmg_train_step: nothing was updated, repeat per step
        h->persist_timed_out = false;
        rc = fsmg_maml_forward_backward(h, support, query, N, K, Q, inner_steps, inner_lr, tokens_on_device);
        if (rc == FSMG_OK) rc = fsmg_apply_update(h, 1.0f, loss);
    }
    return rc;
}

// cfg-E on the device-resident split table: the episode's rows are gathered ONCE into a buffer of the handle's own and the
// inner / outer passes read them there (an index outside the table raises the token-range flag like any bad token)
static int gather_episode(fsmg_handle h, int32_t table_id, const int32_t* sup_idx, int n_sup, const int32_t* qry_idx, int n_qry) {
    if (table_id < 0 || table_id >= fsmg_model::MAX_TABLES || !h->table[table_id]) return fail(h, FSMG_ERR_STATE, "no token table uploaded under this id");
    const int n = n_sup + n_qry;
    if (h->idx_cap < n || h->gather_cap < (int64_t)n * h->T) {
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (h->d_idx) hipFree(h->d_idx);
        if (h->d_gather) hipFree(h->d_gather);
        h->d_idx = nullptr; h->d_gather = nullptr; h->idx_cap = 0; h->gather_cap = 0;
        const int cap = std::max(n, 4096);
        if (hipMalloc((void**)&h->d_idx, sizeof(int) * cap) != hipSuccess || hipMalloc((void**)&h->d_gather, sizeof(int) * (size_t)cap * h->T) != hipSuccess)
            return fail(h, FSMG_ERR_NOMEM, "hipMalloc(episode gather buffers) failed");
        h->idx_cap = cap; h->gather_cap = (int64_t)cap * h->T;
    }
    if (n_sup > 0) HIPCK(h, hipMemcpyAsync(h->d_idx, sup_idx, sizeof(int) * n_sup, hipMemcpyHostToDevice, h->stream));
    if (n_qry > 0) HIPCK(h, hipMemcpyAsync(h->d_idx + n_sup, qry_idx, sizeof(int) * n_qry, hipMemcpyHostToDevice, h->stream));
    HIPCK(h, launch_gather_rows(h->stream, h->table[table_id], h->d_idx, n, h->T, (int)h->table_rows[table_id], h->d_gather, h->d_err));
    return FSMG_OK;
}

int fsmg_maml_forward_backward_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                                       int32_t N, int32_t K, int32_t Q, int32_t inner_steps, float inner_lr) {
    if (!h || !support_idx || !query_idx || N <= 0 || K <= 0 || Q <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    const int rc = gather_episode(h, table_id, support_idx, N * K, query_idx, N * Q);
    if (rc != FSMG_OK) return rc;
    return fsmg_maml_forward_backward(h, h->d_gather, h->d_gather + (size_t)N * K * h->T, N, K, Q, inner_steps, inner_lr, 1);
}

int fsmg_maml_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                           int32_t N, int32_t K, int32_t Q, int32_t inner_steps, float inner_lr, float* loss) {
    if (!h || !support_idx || !query_idx || N <= 0 || K <= 0 || Q <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    const int rc = gather_episode(h, table_id, support_idx, N * K, query_idx, N * Q);
    if (rc != FSMG_OK) return rc;
    return fsmg_maml_step(h, h->d_gather, h->d_gather + (size_t)N * K * h->T, N, K, Q, inner_steps, inner_lr, 1, loss);
}

int fsmg_maml_eval(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* nll) {
    if (!h || !support || !query || !nll) return FSMG_ERR_INVALID;
    if (inner_steps < 0 || inner_steps > 64 || !(inner_lr >= 0.f) || K <= 0 || Q <= 0) return fail(h, FSMG_ERR_INVALID, "bad inner_steps / inner_lr / K / Q");
    hipSetDevice(h->device);
    int rc = FSMG_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = maml_adapt(h, support, N, K, inner_steps, inner_lr, tokens_on_device);
        // the flag of a time-out / token error inside the adaptation is still set: fsmg_eval_step reads and reports it
        if (rc == FSMG_OK) rc = fsmg_eval_step(h, query, N, Q, tokens_on_device, nll);
        const int rc2 = h->P_saved ? restore_theta(h) : FSMG_OK;
        h->have_grads = false;
        if (rc == FSMG_OK) rc = rc2;
        if (!(rc == FSMG_ERR_HIP && h->persist_timed_out)) break;
        h->persist_timed_out = false;                        // adapted with garbage (skipped) steps: repeat on per-step launches
    }
    return rc;
}

int fsmg_eval_batch(fsmg_handle h, const int32_t* queries, int32_t n_episodes, int32_t N, int32_t Q,
                    int32_t tokens_on_device, float* nll) {
    if (!h || !queries || !nll || n_episodes <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    int rc = validate_shape(h, N, 0, Q);
    if (rc != FSMG_OK) return rc;
    const int per = N * Q;
    if (per <= 0) return fail(h, FSMG_ERR_INVALID, "empty query set");
    // validation episodes are independent: batch up to EVAL_EPISODES of them per pass so that the recurrent
    // chain (one launch per time step regardless of the row count) is amortised over many rows
    constexpr int EVAL_EPISODES = 16;
    if ((rc = ensure_scratch(h, per * std::min<int>(n_episodes, EVAL_EPISODES))) != FSMG_OK) return rc;
    if ((rc = ensure_khf(h)) != FSMG_OK) return rc;
    const int chunk_eps = std::min<int>(h->Bcap / per, 64);
    if (h->eval_cap < chunk_eps) {
        HIPCK(h, hipStreamSynchronize(h->stream));
        if (h->d_eval) hipFree(h->d_eval);
        HIPCK(h, hipMalloc((void**)&h->d_eval, sizeof(float) * chunk_eps));
        h->eval_cap = chunk_eps;
    }
    bool retried = false;
    for (int e0 = 0; e0 < n_episodes; e0 += chunk_eps) {
        const int ne = std::min(chunk_eps, n_episodes - e0);
        const int B = ne * per;
        choose_schedule(h, B);
        const int32_t* q = queries + (size_t)e0 * per * h->T;
        if ((rc = stage_tokens(h, q, 0, q, B, tokens_on_device)) != FSMG_OK) return rc;
        rc = run_graphed(h, "ev:" + std::to_string(per) + ":" + std::to_string(ne), [&]() -> int {
            int r = token_prep(h, 0, B);
            if (r == FSMG_OK) r = forward(h, B, per, ne, h->d_eval, false);
            return r;
        });
        if (rc != FSMG_OK) return rc;
        h->lastB = B;
        rc = check_tokens_and_read(h, h->d_eval, 1.0f, nll + e0, ne);
        if (rc == FSMG_ERR_HIP && h->persist_timed_out && !retried) {
            // a persistent kernel could not get its blocks resident: the handle has switched to one launch per time
            // step; repeat this chunk that way (validation must not abort a training run, nor leave peer ranks hanging)
            h->persist_timed_out = false;
            retried = true;
            e0 -= chunk_eps;
            continue;
        }
        if (rc != FSMG_OK) return rc;
        retried = false;
    }
    return FSMG_OK;
}

int fsmg_eval_step(fsmg_handle h, const int32_t* query, int32_t N, int32_t Q, int32_t tokens_on_device, float* nll) {
    return fsmg_eval_batch(h, query, 1, N, Q, tokens_on_device, nll);
}

int fsmg_sample(fsmg_handle h, int32_t num, int32_t* out_tokens) {
    if (!h || num < 0 || (num > 0 && !out_tokens)) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    hipStream_t s = h->stream;
    const int Hp = h->Hp, L = h->L;
    float* hb = h->dec;                       // [L][2][Hp]
    float* cb = h->dec + (size_t)L * 2 * Hp;  // [L][Hp]
    float* arg_scratch = cb + (size_t)L * Hp;
    HIPCK(h, launch_fill32(s, h->dec, 0u, (long long)((sizeof(float) * (size_t)L * 3 * Hp) / 4)));
    std::vector<int> toks(num);
    int word = h->V;                          // start word
    int* d_hist = nullptr;
    HIPCK(h, hipMalloc((void**)&d_hist, sizeof(int) * (size_t)(num + 1)));
    HIPCK(h, hipMemcpyAsync(d_hist, &word, sizeof(int), hipMemcpyHostToDevice, s));
    // greedy decode is a host loop in the reference too (one sess.run per token, lstm_baseline.py:142-154):
    // the argmax token is read back each step because it selects the next embedding row.
    for (int i = 0; i < num; ++i) {
        const float* x = h->P + h->off_emb + (size_t)word * h->Ep;
        const int pin = i & 1, pout = pin ^ 1;
        for (int l = 0; l < L; ++l) {
            float* h_in = hb + ((size_t)l * 2 + pin) * Hp;
            float* h_out = hb + ((size_t)l * 2 + pout) * Hp;
            hipError_t e = launch_decode_cell(s, h->P + h->off_kx[l], h->in_dim[l], h->P + h->off_kh[l],
                                              h->P + h->off_b[l], x, h_in, h_out, cb + (size_t)l * Hp, Hp);
            if (e != hipSuccess) { hipFree(d_hist); return fail(h, FSMG_ERR_HIP, hipGetErrorString(e)); }
            x = h_out;
        }
        hipError_t e = launch_decode_argmax(s, h->P + h->off_w, h->V1p, h->P + h->off_d, x, Hp, h->V1,
                                            d_hist + i + 1, arg_scratch);
        if (e == hipSuccess) e = hipMemcpyAsync(&word, d_hist + i + 1, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { hipFree(d_hist); return fail(h, FSMG_ERR_HIP, hipGetErrorString(e)); }
        toks[i] = word;
    }
    hipFree(d_hist);
    for (int i = 0; i < num; ++i) out_tokens[i] = toks[i];
    return FSMG_OK;
}

int fsmg_read_losses(fsmg_handle h, float* out, int32_t n) {
    if (!h || !out || n <= 0 || n > RING_CAP) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    std::vector<float> ring(RING_CAP);
    long long step = 0;
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(ring.data(), h->d_ring, sizeof(float) * RING_CAP, hipMemcpyDeviceToHost));
    HIPCK(h, hipMemcpy(&step, h->d_step, sizeof(step), hipMemcpyDeviceToHost));
    poll_skipped(h);
    if (step < n) return fail(h, FSMG_ERR_INVALID, "fewer train steps than requested losses");
    for (int i = 0; i < n; ++i) out[i] = ring[(step - n + i) % RING_CAP];
    return FSMG_OK;
}

int fsmg_get_stats(fsmg_handle h, fsmg_stats* out) {
    if (!h || !out) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    poll_skipped(h);
    std::memset(out, 0, sizeof(*out));
    out->timeouts = h->n_timeouts;
    out->steps_skipped_timeout = h->host_counters ? h->host_counters[0] : 0;
    out->steps_skipped_token_range = h->host_counters ? h->host_counters[1] : 0;
    out->steps_skipped_peer_failure = h->host_counters ? h->host_counters[2] : 0;
    out->xcd_launches = h->n_xcd_launches;
    out->persistent_launches = h->n_persist_launches;
    out->step_launches = h->n_step_launches;
    out->persistent_path = h->persist ? 1 : 0;
    out->fallback_steps_left = h->fallback_left;
    return FSMG_OK;
}

int fsmg_debug_set(fsmg_handle h, const char* what, int64_t value) {
    if (!h || !what) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->aux) HIPCK(h, hipStreamSynchronize(h->aux));
    if (!std::strcmp(what, "chain_spin_limit")) h->chain_spin_limit = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (!std::strcmp(what, "fallback_steps")) h->fallback_steps = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30));
    else if (!std::strcmp(what, "eager")) h->eager = value != 0;
    else if (!std::strcmp(what, "persistent")) { h->persist = h->persist_cfg = value != 0; h->fallback_left = 0; }
    else return fail(h, FSMG_ERR_NAME, std::string("unknown knob '") + what + "'");
    drop_graphs(h);
    return FSMG_OK;
}

// ---- unigram baseline
struct fsmg_unigram {
    int V = 0, device = 0;
    hipStream_t stream = nullptr;
    unsigned* counts = nullptr;
    int* words = nullptr; int64_t words_cap = 0;
    float* out = nullptr;           // [0] nll, [1] sum of counts; then an int: argmax; then an int: error flag
    std::string err;
};
namespace {
int ufail(fsmg_unigram* u, int code, const std::string& msg) { if (u) u->err = msg; else g_create_error = msg; return code; }
#define UCK(u, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return ufail(u, FSMG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
int unigram_stage(fsmg_unigram* u, const int32_t* words, int64_t n, int on_device, const int** dev) {
    if (on_device) { *dev = words; return FSMG_OK; }
    if (u->words_cap < n) {
        UCK(u, hipStreamSynchronize(u->stream));
        if (u->words) hipFree(u->words);
        u->words = nullptr; u->words_cap = 0;
        const int64_t cap = std::max<int64_t>(n, 1 << 16);
        if (hipMalloc((void**)&u->words, sizeof(int) * (size_t)cap) != hipSuccess) return ufail(u, FSMG_ERR_NOMEM, "hipMalloc(words) failed");
        u->words_cap = cap;
    }
    UCK(u, hipMemcpyAsync(u->words, words, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, u->stream));
    *dev = u->words;
    return FSMG_OK;
}
int unigram_read(fsmg_unigram* u, float* nll) {
    float host[4] = {0.f, 0.f, 0.f, 0.f};
    UCK(u, hipMemcpyAsync(host, u->out, sizeof(host), hipMemcpyDeviceToHost, u->stream));
    UCK(u, hipStreamSynchronize(u->stream));
    int flag; std::memcpy(&flag, &host[3], 4);
    if (flag != 0) {
        UCK(u, hipMemsetAsync(u->out + 3, 0, 4, u->stream));
        return ufail(u, FSMG_ERR_TOKEN_RANGE, "word id outside [0, input_size)");
    }
    if (nll) *nll = host[0];
    return FSMG_OK;
}
}  // namespace

int fsmg_unigram_create(int32_t input_size, int32_t device, fsmg_unigram_handle* out) {
    if (!out || input_size <= 0) return ufail(nullptr, FSMG_ERR_INVALID, "bad input_size / out");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ufail(nullptr, FSMG_ERR_NO_DEVICE, "no HIP device visible: libfsmg has no CPU fallback");
    if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return ufail(nullptr, FSMG_ERR_NO_DEVICE, "device ordinal out of range");
    fsmg_unigram* u = new (std::nothrow) fsmg_unigram();
    if (!u) return ufail(nullptr, FSMG_ERR_NOMEM, "host allocation failed");
    u->V = input_size; u->device = device;
    if (hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&u->counts, sizeof(unsigned) * (size_t)input_size) != hipSuccess ||
        hipMalloc((void**)&u->out, 256) != hipSuccess) { fsmg_unigram_destroy(u); return ufail(nullptr, FSMG_ERR_NOMEM, "device allocation failed"); }
    hipMemsetAsync(u->out, 0, 256, u->stream);
    if (launch_fill32(u->stream, u->counts, 1u, input_size) != hipSuccess || hipStreamSynchronize(u->stream) != hipSuccess) {     // alpha = 1
        fsmg_unigram_destroy(u); return ufail(nullptr, FSMG_ERR_HIP, "count initialisation failed");
    }
    *out = u;
    return FSMG_OK;
}
int fsmg_unigram_destroy(fsmg_unigram_handle u) {
    if (!u) return FSMG_OK;
    hipSetDevice(u->device);
    if (u->stream) hipStreamSynchronize(u->stream);
    if (u->counts) hipFree(u->counts);
    if (u->words) hipFree(u->words);
    if (u->out) hipFree(u->out);
    if (u->stream) hipStreamDestroy(u->stream);
    delete u;
    return FSMG_OK;
}
const char* fsmg_unigram_last_error(fsmg_unigram_handle u) { return u ? u->err.c_str() : g_create_error.c_str(); }
int fsmg_unigram_nll(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* nll) {
    if (!u || !words || n <= 0 || !nll) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    const int* dev = nullptr;
    int rc = unigram_stage(u, words, n, on_device, &dev);
    if (rc != FSMG_OK) return rc;
    UCK(u, launch_unigram_nll(u->stream, dev, n, u->counts, u->V, u->out, (int*)(u->out + 3)));
    return unigram_read(u, nll);
}
int fsmg_unigram_train(fsmg_unigram_handle u, const int32_t* words, int64_t n, int32_t on_device, float* loss) {
    if (!u || !words || n <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    const int* dev = nullptr;
    int rc = unigram_stage(u, words, n, on_device, &dev);
    if (rc != FSMG_OK) return rc;
    // the loss with the counts BEFORE the update, like LSTMBaseline.train's pre-update loss; a batch with an id out of range
    // is rejected as a whole (the NLL kernel has seen every word before the update runs)
    UCK(u, launch_unigram_nll(u->stream, dev, n, u->counts, u->V, u->out, (int*)(u->out + 3)));
    float l = 0.f;
    rc = unigram_read(u, &l);
    if (rc != FSMG_OK) return rc;
    UCK(u, launch_unigram_update(u->stream, dev, n, u->counts, u->V, (int*)(u->out + 3)));
    if (!on_device) UCK(u, hipStreamSynchronize(u->stream));      // the staging buffer may be reused by the next call
    if (loss) *loss = l;
    return FSMG_OK;
}
int fsmg_unigram_get_counts(fsmg_unigram_handle u, float* host, int64_t count) {
    if (!u || !host || count != u->V) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    std::vector<unsigned> tmp((size_t)count);
    UCK(u, hipStreamSynchronize(u->stream));
    UCK(u, hipMemcpy(tmp.data(), u->counts, sizeof(unsigned) * (size_t)count, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < count; ++i) host[i] = (float)tmp[(size_t)i];
    return FSMG_OK;
}
int fsmg_unigram_set_counts(fsmg_unigram_handle u, const float* host, int64_t count) {
    if (!u || !host || count != u->V) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    std::vector<unsigned> tmp((size_t)count);
    for (int64_t i = 0; i < count; ++i) {
        if (!(host[i] >= 0.f) || host[i] > 4.0e9f) return ufail(u, FSMG_ERR_INVALID, "counts must be finite and >= 0");
        tmp[(size_t)i] = (unsigned)std::llround((double)host[i]);
    }
    UCK(u, hipStreamSynchronize(u->stream));
    UCK(u, hipMemcpy(u->counts, tmp.data(), sizeof(unsigned) * (size_t)count, hipMemcpyHostToDevice));
    return FSMG_OK;
}
int fsmg_unigram_argmax(fsmg_unigram_handle u, int32_t* word) {
    if (!u || !word) return FSMG_ERR_INVALID;
    hipSetDevice(u->device);
    UCK(u, launch_unigram_argmax(u->stream, u->counts, u->V, (int*)(u->out + 2)));
    int w = 0;
    UCK(u, hipMemcpyAsync(&w, u->out + 2, sizeof(int), hipMemcpyDeviceToHost, u->stream));
    UCK(u, hipStreamSynchronize(u->stream));
    *word = w;
    return FSMG_OK;
}

int fsmg_debug_clock_begin(fsmg_handle h, int32_t microseconds) {
    if (!h || microseconds <= 0 || microseconds > 1000000) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    if (!h->probe) HIPCK(h, hipStreamCreateWithFlags(&h->probe, hipStreamNonBlocking));
    if (!h->d_probe) HIPCK(h, hipMalloc((void**)&h->d_probe, 64));
    HIPCK(h, hipMemsetAsync(h->d_probe, 0, 64, h->probe));
    HIPCK(h, launch_clock_probe(h->probe, (long long)microseconds * 100, h->d_probe));
    return FSMG_OK;
}
int fsmg_debug_clock_end(fsmg_handle h, float* ghz) {
    if (!h || !ghz) return FSMG_ERR_INVALID;
    if (!h->probe || !h->d_probe) return fail(h, FSMG_ERR_STATE, "fsmg_debug_clock_end without fsmg_debug_clock_begin");
    hipSetDevice(h->device);
    unsigned long long v[2] = {0, 0};
    HIPCK(h, hipStreamSynchronize(h->probe));
    HIPCK(h, hipMemcpy(v, h->d_probe, sizeof(v), hipMemcpyDeviceToHost));
    *ghz = v[1] ? (float)((double)v[0] / (double)v[1] * 0.1) : 0.0f;
    return FSMG_OK;
}

int fsmg_debug_dims(fsmg_handle h, int32_t dims[5]) {
    if (!h || !dims) return FSMG_ERR_INVALID;
    dims[0] = h->Ep; dims[1] = h->Hp; dims[2] = h->V1p; dims[3] = h->lastB; dims[4] = h->T;
    return FSMG_OK;
}

int fsmg_debug_read(fsmg_handle h, const char* what, float* host, int64_t count) {
    if (!h || !what || !host || count <= 0) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    const int64_t B = h->lastB, T = h->T, Hp = h->Hp, G4 = h->G4, rows = T * B;
    const float* src = nullptr; int64_t cap = 0;
    auto layer_of = [&](const char* prefix) -> int {
        const size_t n = std::strlen(prefix);
        if (std::strncmp(what, prefix, n) != 0) return -1;
        const int l = std::atoi(what + n);
        return (l >= 0 && l < h->L && std::strlen(what) > n) ? l : -1;
    };
    int l;
    if (!std::strcmp(what, "xcd_bx3")) { host[0] = h->xcd_bx3 ? 1.0f : 0.0f; return FSMG_OK; }      // a host-side fact: which XCD-local kernel family this handle runs
    if (!std::strcmp(what, "xcd_partitioned")) {      // ... and whether its train passes take the XCD-partitioned order: [0] yes / no, [1] XCDs the chains occupy, [2] the last pass
        host[0] = h->xov ? 1.0f : 0.0f;
        if (count > 1) { const int b = h->lastB > 0 ? h->lastB : 45, rpx = lstm_xcd16_packed_rows(b); host[1] = (h->xov && rpx > 0) ? (float)((b + rpx - 1) / rpx) : 8.0f; }
        if (count > 2) host[2] = h->xov_last ? 1.0f : 0.0f;       // [2] whether the LAST train pass took it (its row count decides per call)
        return FSMG_OK;
    }
    if (!std::strcmp(what, "logits")) { src = h->logits; cap = rows * h->V1p; }
    else if (!std::strcmp(what, "dlogits")) { src = h->dlogits; cap = rows * h->V1p; }
    else if (!std::strcmp(what, "lse")) { src = h->lse; cap = rows; }
    else if (!std::strcmp(what, "ce")) { src = h->ce; cap = rows; }
    else if (!std::strcmp(what, "dx")) { src = h->dXemb; cap = rows * h->Ep; }
    else if (!std::strcmp(what, "dh")) { src = h->dH; cap = rows * Hp; }
    else if (!std::strcmp(what, "gnorm")) { src = h->d_gnorm; cap = 1; }
    else if (!std::strcmp(what, "tail")) { src = h->G + h->n_flat; cap = FSMG_GRAD_TAIL; }
    else if ((l = layer_of("gates")) >= 0) { src = h->Z[l]; cap = rows * G4; }
    else if ((l = layer_of("h")) >= 0) { src = h->Hs[l]; cap = (T + 1) * B * Hp; }
    else if ((l = layer_of("c")) >= 0) { src = h->Cs[l]; cap = (T + 1) * B * Hp; }
    else return fail(h, FSMG_ERR_NAME, std::string("unknown debug buffer '") + what + "'");
    if (count > cap) return fail(h, FSMG_ERR_SIZE, "debug read larger than the buffer");
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(host, src, sizeof(float) * count, hipMemcpyDeviceToHost));
    return FSMG_OK;
}

int fsmg_debug_step_profile(fsmg_handle h, int32_t which, uint64_t* stamps, int64_t cap, int32_t* n_blocks,
                            int32_t* n_waves) {
    if (!h || !stamps || !n_blocks || !n_waves || h->lastB <= 0 || h->T < 3) return FSMG_ERR_INVALID;
    hipSetDevice(h->device);
    const int B = h->lastB, Hp = h->Hp, G4 = h->G4, l = h->L - 1, t = h->T / 2;
    const int nb = which == 0 ? (G4 / 16) * ((B + 15) / 16) : (Hp / 16) * ((B + 15) / 16);
    const int nw = which == 0 ? 4 : 8;
    const int64_t n = (int64_t)nb * nw * 8;
    if (cap < n) return fail(h, FSMG_ERR_SIZE, "stamp buffer too small");
    unsigned long long* d = nullptr;
    HIPCK(h, hipMalloc((void**)&d, sizeof(unsigned long long) * n));
    HIPCK(h, hipMemsetAsync(d, 0, sizeof(unsigned long long) * n, h->stream));
    for (int rep = 0; rep < 3; ++rep) {        // last repetition is the one read back (warm instruction cache)
        if (which == 0) {
            LstmFwdArgs a{};
            const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
            a.KhF = h->khf + (size_t)(2 * l) * Hp * G4; a.hF_prev = h->HF[l] + (size_t)t * Bp16 * Hp;
            a.hF_next = h->HF[l] + (size_t)(t + 1) * Bp16 * Hp; a.z = h->Z[l] + (size_t)t * B * G4;
            a.c_prev = h->Cs[l] + (size_t)t * B * Hp; a.c_next = h->Cs[l] + (size_t)(t + 1) * B * Hp;
            a.h_next = h->Hs[l] + (size_t)(t + 1) * B * Hp; a.B = B; a.Hp = Hp;
            HIPCK(h, launch_lstm_fwd_step(h->stream, a, d));
        } else {
            LstmBwdArgs a{};
            const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
            a.KhF = h->khf + (size_t)(2 * l + 1) * Hp * G4; a.dzF_next = h->dzF + (size_t)((t + 1) & 1) * Bp16 * G4;
            a.dzF_cur = h->dzF + (size_t)(t & 1) * Bp16 * G4; a.gates = h->Z[l] + (size_t)t * B * G4;
            a.c_t = h->Cs[l] + (size_t)(t + 1) * B * Hp; a.c_prev = h->Cs[l] + (size_t)t * B * Hp; a.dc = h->dC;
            a.dh_top = h->dH + (size_t)t * B * Hp; a.B = B; a.Hp = Hp;
            HIPCK(h, launch_lstm_bwd_step(h->stream, a, d));
        }
    }
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipMemcpy(stamps, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    hipFree(d);
    *n_blocks = nb; *n_waves = nw;
    return FSMG_OK;
}

int fsmg_timing_enable(fsmg_handle h, int32_t on) {
    if (!h) return FSMG_ERR_INVALID;
    drain_timers(h);
    h->timing = on != 0;
    return FSMG_OK;
}
int fsmg_timing_select(fsmg_handle h, const char* kernel_class) {
    if (!h) return FSMG_ERR_INVALID;
    drain_timers(h);
    h->timing_only = kernel_class ? kernel_class : "";
    return FSMG_OK;
}
int fsmg_timing_read(fsmg_handle h, const char* kernel_class, double* total_ms, int64_t* launches) {
    if (!h || !kernel_class) return FSMG_ERR_INVALID;
    drain_timers(h);
    auto it = h->timers.find(kernel_class);
    if (total_ms) *total_ms = it == h->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == h->timers.end() ? 0 : it->second.launches;
    return FSMG_OK;
}
int fsmg_timing_reset(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    drain_timers(h);
    h->timers.clear();
    return FSMG_OK;
}

}  // extern "C"
