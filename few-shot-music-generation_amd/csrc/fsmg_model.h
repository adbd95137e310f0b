// Host-side internals of libfsmg shared by the api_*.hip translation units: the model handle, its HBM layout, the helpers every
// C entry point uses.  Nothing here is exported; include/fsmg.h is the public surface, fsmg_kernels.h the launcher interface.
//   api_handle.hip    fsmg_create / fsmg_destroy, knobs, statistics, greedy decode
//   api_layout.hip    padded parameter layout, host <-> device tensor transfers, parameter / optimizer-state entry points
//   api_scratch.hip   activation scratch sizing, split-K policy
//   api_schedule.hip  which kernel and which order a pass takes (choose_schedule, the XCD-partitioned gate / queue), gemm()
//   api_forward.hip   forward pass builder          api_backward.hip  backward pass builder
//   api_update.hip    clip + Adam, inner-loop SGD, time-out bookkeeping, loss read-back
//   api_step.hip      train / eval / MAML-style entry points          api_comm.hip  RCCL glue (fsmg_comm_*)
//   api_unigram.hip   unigram baseline                                 api_debug.hip debug reads, timers, clock probe
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fsmg.h"
#include "fsmg_kernels.h"

namespace fsmg_host {
using namespace fsmg;
struct OpBatch;

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

constexpr int RING_CAP = 1024;
constexpr int64_t FLAT_ALIGN = 64;   // floats (256 B)
constexpr int MAX_SPLIT = 16;        // K-split cap of the GEMMs (pick_split)

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

}  // namespace fsmg_host

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
    std::vector<fsmg_host::ParamDesc> params;
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
    float *dC = nullptr, *dH = nullptr, *logits = nullptr, *dlogits = nullptr, *lse = nullptr, *ce = nullptr, *dXemb = nullptr, *dXpart = nullptr;
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
    bool retry_armed = false;           // a train step was skipped on the device and the handle has changed its schedule for the repeat: a persistent kernel
                                        // gave up (per-step launches now) or a row left the fused softmax's range (cross-entropy pass now)
    bool persist_cfg = true;            // what the configuration asked for; `persist` is what is in force right now
    int fallback_steps = 200;           // FSMG_FALLBACK_STEPS: train steps on per-step launches after a time-out, then the persistent path is tried again
    int fallback_left = 0;
    long long* host_counters = nullptr; // host-mapped tallies written by k_step_increment: [0] steps skipped after a time-out, [1] after a token-range error
    long long* d_counters = nullptr;    // the same memory as the device sees it
    long long seen_timeouts = 0, seen_token_errors = 0, seen_peer_failures = 0, seen_range_skips = 0;
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
    int xcd_variant_bwd = -1;           // FSMG_XCD_VARIANT_BWD: the backward kernels' bits alone (-1: xcd_variant / the default)
    float* khx = nullptr;
    bool xcd_bx3 = false;               // the XCD-local recurrence on the bf16 matrix pipe (k_lstm_*_xcd16 at hidden 512, k_lstm_*_pair16 at hidden 1024); one format
                                        // per handle at a time (weight images, hand-off buffer)
    bool xcd_bx3_forced = false;        // FSMG_XCD_BX3 set: the format never follows the row count of the train passes (hidden 1024: select_xcd_format)
    float* HX = nullptr; int64_t hx_floats = 0;
    float* inboxX = nullptr; int64_t inboxx_floats = 0;
    int* d_inbox_dirty = nullptr;       // device word: != 0 -> the next BPTT pass refills the inboxes first (set at creation, when the scratch moves,
                                        // and by k_step_increment after a time-out; a completed pass leaves every word reset by its reader)
    int* tickets = nullptr;             // [TICKET_LAUNCHES][8]
    static constexpr int TICKET_LAUNCHES = 64;
    int ticket_next = 0;
    int64_t n_timeouts = 0, n_persist_launches = 0, n_xcd_launches = 0, n_step_launches = 0;   // fsmg_get_stats
    bool khf_dirty = true;              // host wrote parameters since the last repack
    // the column-split fragment copies (khf) are read by the column-split / per-step kernels only: a handle whose passes take the
    // XCD-local kernels refreshes just the register images behind an update and leaves khf STALE until a pass is about to read it
    // (a big validation batch, the fallback after a time-out: ensure_cs(), api_update.hip) -- 45 -> ~17 us per update at hidden 1024
    bool lazy_cs = true;                // FSMG_LAZY_CS=0: every repack refreshes both layouts
    bool cs_stale = false;
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
    hipStream_t aux = nullptr;          // second stream (DEFAULT priority since round 5: api_handle.hip pick_concurrent_aux) for what runs beside the main one
    // whether `aux` really runs beside `stream`: the runtime maps a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues per
    // priority, and two streams of one handle on the same queue run one after the other.  fsmg_create probes it (launch_queue_probe)
    // and draws another stream until the two overlap; -1: never found one (the orders that need two streams side by side are
    // switched off for the handle), else the number of streams it drew
    int aux_tries = 0;
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
    bool xov_eligible = false;          // what `xov` was decided to be at creation before the second-stream probe had its say (fsmg_debug_set("reprobe_aux"))
    bool bucket0_recorded = false;      // backward() recorded ev_bucket[0] itself (two-stream / XCD-partitioned order)
    int xov_dw_split = 4;               // K split of dW under this schedule: an item must be short against the chain it runs beside
    int xov_tail = 0;                   // FSMG_XOV_TAIL: time steps whose projection rows are left to a chip-wide launch behind the chain (0: none)
    int xov_pub = 6;                    // FSMG_XOV_PUB: the forward chain publishes every this many steps (a 256-row tile is 5.7 steps of 45 rows)
    int xov_strikes = 0;                // time-outs of passes in the XCD-partitioned order: the second one parks the schedule for this handle
    bool xov_last = false;              // the pass in flight took the XCD-partitioned order
    int xov_parts = 3;                  // FSMG_XOV_PARTS: 1 = forward pair, 2 = backward pair (dW beside the top chain), 4 = stacked layers: dK of layer l + 1
                                        // beside the BPTT chain of layer l (hidden 1024); default 3 at hidden 512, 2 at hidden 1024 (where the order is opt-in and
                                        // every part was measured to lose: api_handle.hip, profiles/r06_cfgC_xov_ab.txt)
    int* xov_prog = nullptr;            // [T] progress counters of the forward chain (LstmFwdXcdArgs::progress), the projection's gate
    // forward projection / dW: [0..1] draw counters, [2] stop flag, [3] items, [4 ..] claim words (gemm_restricted)
    static constexpr int XOV_CTL = 8192;
    int* xov_ctl = nullptr;             // [2][XOV_CTL]
    int64_t slab_cap = 0;
    // whole-phase hipGraphs, keyed by the shape of the call; dropped when scratch moves
    std::map<std::string, hipGraphExec_t> graphs;
    struct LaunchCounts { int64_t xcd = 0, persist = 0, step = 0; bool bwd_xcd = false; };   // bwd_xcd: what last_bwd_xcd was when the capture ended
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
    // clip + Adam in two launches (eager passes): everything in front of softmax_w on the main stream, [softmax_w, softmax_b] on the
    // auxiliary stream beside the NEXT step's input phase (token_prep, the x-part GEMM and the fills read none of it).  The first
    // launch publishes (go, clip scale, alpha) in d_decided, the second consumes them -- it runs while k_step_increment moves the
    // step counter and clears the flags.  upd_pending: the second launch may still be in flight; settle_pending() orders the main
    // stream behind it (fsmg_host::begin_call does that for every entry point, forward() right before the first recurrent chain).
    // MEASURED AND REJECTED as a default (round 5, profiles/r05_ab_step_variants.txt, r05_step_timeline_upd_split.txt): the second
    // launch hides its 27 us, but beside it token_prep takes 9.9 instead of 4.9 us, the x-part GEMM 56.9 instead of 45.6 us, and the
    // wait in front of the first chain adds ~5 us: cfg-B -1.1 %, cfg-C -0.5 %, cfg-E -0.8 %, cfg-D +0.4 %, ref-default +0.4 %.
    bool upd_split = false;             // FSMG_UPD_SPLIT=1 / fsmg_debug_set("upd_split", 1): two launches
    bool upd_pending = false;
    hipEvent_t ev_upd_fork = nullptr, ev_upd = nullptr;
    float* d_decided = nullptr;         // [4]: go (1 / 0), clip scale, alpha, unused
    // the split-K slabs of the main lane hold a GEMM's partial sums whose REDUCE op is still waiting in this batch (the XCD-partitioned
    // order's dW): gemm() flushes it before anything else writes those slabs
    fsmg_host::OpBatch* slabs_owner = nullptr;
    // self-check of the gated projection (XCD-partitioned order): the first passes of a handle compute the logits a second time on the
    // serial path and compare the words; a difference skips the step like a time-out and parks the order for this handle
    int xov_selfcheck_left = 2;         // FSMG_XOV_SELFCHECK=n
    int xov_selfcheck_every = 1000;     // FSMG_XOV_SELFCHECK_EVERY / fsmg_debug_set("xov_selfcheck_every"): one more checked pass every this many passes
                                        // under the XCD-partitioned order, for the handle's whole life (0: the first passes only)
    long long xov_passes = 0, xov_selfcheck_runs = 0;      // passes that took the order / passes that were checked (fsmg_debug_read("xov_selfcheck"))
    bool xov_selfcheck_fault = false;   // fsmg_debug_set("xov_selfcheck_fault"): compare against a buffer that is NOT the recomputed logits (tests)
    long long seen_selfcheck_mismatch = 0;
    // cross entropy writes dlogits over the logits it has just read (one 230 MB buffer instead of two at cfg-B: the pair no longer
    // exceeds the 256 MB memory-side cache); fsmg_debug_read("logits") of a train pass then returns dlogits -- FSMG_INPLACE_DLOGITS=0
    // or fsmg_debug_set("inplace_dlogits", 0) keeps both
    bool inplace_dlogits = true;
    // Fused softmax of a train pass (round 5): the projection's epilogue stores E = exp(logit) and per-slice softmax partials, a
    // per-row kernel (k_ce_finish) derives lse / ce / c_r = 1 / (n S_r), patches E[r][y_r] -= S_r and writes c_r * h_r; then
    // dH = diag(c) (E' W^T) (the scale rides in dH's slab sum), dW = (diag(c) Hout)^T E', dd = sum_r c_r E'[r] (weighted column sums in
    // the dW kernel): (softmax - onehot) / n is never materialised and the 460 MB cross-entropy pass is gone.  Taken where the
    // projection's weight gradient runs on the 256 x 256-tile kernel (use_h_gemm; the projection itself: any bf16-split kernel) and
    // dlogits are in place; a row whose sum of exp(logit) leaves [e^-60, 1e30] or whose exp(target logit) is below 1e-30 (CE_SUM_MIN /
    // CE_SUM_MAX / CE_TGT_MIN) makes the step fall back to the shifted softmax (launch_ce_rows) for good on this handle.
    // FSMG_FUSED_SOFTMAX=0 / fsmg_debug_set("fused_softmax", 0): the cross-entropy pass of rounds 1-4.
    bool fused_softmax = true;
    bool fs_call = false;               // the pass in flight takes it (forward() decides, backward() follows)
    float* Hsc = nullptr;               // [T * Bcap][Hp]: c_r * top-layer h_r (dW's A operand)
    float* crow = nullptr;              // [T * Bcap + 32]: c_r (dH's row scale, dW's column-sum weights)
    long long seen_softmax_range = 0;
    // the bandwidth-bound tail of a backward pass -- the deferred slab sums (dW / dd, the upper layers' weight gradients, dx), the mean
    // loss, k_embed_grad, the embedding-slice norm -- on the auxiliary stream BESIDE the bottom layer's weight-gradient GEMM, which is
    // issued behind dx instead of in front of it (the two do not depend on each other); the main stream waits for it right behind that
    // GEMM.  Same kernels on the same operands: same bits.  FSMG_TAIL_ASIDE=0 / fsmg_debug_set("tail_aside", 0): everything in line.
#ifdef FSMG_EXPERIMENTS
    // MEASURED AND REJECTED (round 5, experiment builds only: profiles/r05_ce_under_tail_*): the cross entropy UNDER the forward pair's
    // tail.  Every work-queue tile of the projection is stored write-through and counts itself in xov_done[row tile]; the cross entropy
    // is a persistent grid on a third stream that starts when the chain is over and takes each row behind its row tile's counter.
    // Bit-identical -- and 7-25 % SLOWER: the persistent blocks hold CUs the last tiles need (136 VGPRs: no co-residency with a GEMM
    // block), the tail grows from 105 to 225 us and the cross entropy itself takes 330 us instead of 92; the idle CU-time of the tail
    // (~12 k CU-us) is half of what the pass needs in the first place.  FSMG_CE_TAIL=1, FSMG_CE_TAIL_BLOCKS.
    bool ce_tail = false;
    int ce_tail_blocks = 256;
    hipStream_t aux2 = nullptr;
    hipEvent_t ev_ce_fork = nullptr, ev_ce = nullptr;
#endif
    int* xov_done = nullptr;            // [XOV_DONE] completion counters of the projection's row tiles (GemmArgs::done; experiment builds)
    static constexpr int XOV_DONE = 256;
    bool tail_aside = true;
    bool side_pending = false;          // the auxiliary stream may still be reading the main lane's slabs (gemm() waits before it reuses them)
    hipEvent_t ev_side_fork = nullptr, ev_side = nullptr;
    // take_turn (api_handle.hip): recorded on this handle's stream / auxiliary stream by the NEXT handle that takes the device, which then
    // waits for them
    hipEvent_t ev_turn = nullptr, ev_turn_aux = nullptr;
    std::atomic<bool> capturing{false};      // run_graphed is between BeginCapture and EndCapture on `stream`: nobody else records on it
    std::string err;
    bool timing = false;
    std::string timing_only;
    std::map<std::string, fsmg_host::TimerClass> timers;
};

namespace fsmg_host {

extern thread_local std::string g_create_error;

inline int fail(fsmg_model* h, int code, const std::string& msg) {
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

inline void drain_timers(fsmg_model* h) {
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

// ------------------------------------------------------------------ one prologue for every C entry point that takes a handle
// begin_call: selects the handle's device and settles what an earlier call left in flight on another stream (settle_pending: the
// softmax half of the last clip + Adam update runs on the auxiliary stream beside the NEXT step's input phase -- any other reader
// or writer of the parameters, the optimizer state or the gradient buffer must be ordered behind it first).  The train-step entry
// points pass keep_pending: their forward pass orders itself behind the update where it first reads the softmax parameters.
int begin_call(fsmg_model* h, bool keep_pending = false);
int settle_pending(fsmg_model* h);
// the turnstile's lock (api_handle.hip take_turn).  run_graphed holds it from `capturing = true` to `capturing = false`: take_turn
// reads the flag and records an event on the last handle's stream under the same lock, so it can never record into a capture that
// began between its check and its record (check-then-act race between threads)
std::mutex& turn_mutex();
// One handle's passes at a time per device (begin_call -> take_turn, api_handle.hip).  The persistent kernels -- recurrence chains with
// cross-CU hand-offs, gated work-queue GEMMs -- spin on CUs they hold and assume that their peers are resident; two handles of a process
// whose calls overlap on the GPU (calls return before the work is done) time-share those CUs at best and run into the hand-off time-out
// at worst (a skipped and repeated step: correct, slow).  So the first call of handle B after a call of handle A on the same device
// records what A has issued so far (both of A's streams) and makes B's streams wait for it.  A process with one handle pays a mutex
// and a pointer compare per call.  Calls on different handles from different threads AT THE SAME TIME are ordered behind the part of
// the other call that had been issued, no more.  FSMG_TURNSTILE=0: off.
#define BEGIN_CALL(h, ...)                                                           \
    do {                                                                             \
        const int rc_begin_ = fsmg_host::begin_call(h, ##__VA_ARGS__);               \
        if (rc_begin_ != FSMG_OK) return rc_begin_;                                  \
    } while (0)

// ------------------------------------------------------------------ layout (api_layout.hip)
void compute_dims(const fsmg_config& c, fsmg_model* m);
int64_t build_layout(fsmg_model* m);
int64_t state_bytes_for(int64_t n_flat);
const ParamDesc* find_param(fsmg_model* h, const char* name);
int upload_tensor(fsmg_model* h, float* flat, const char* name, const float* host, int64_t count);
int download_tensor(fsmg_model* h, const float* flat, const char* name, float* host, int64_t count);

// ------------------------------------------------------------------ scratch + split-K policy (api_scratch.hip)
int pick_split(int64_t M, int64_t N, int64_t K, int64_t slots = 0, bool bx3 = false, int tile_mn = 0);
int ensure_scratch(fsmg_model* h, int B);
void drop_graphs(fsmg_model* h);

// A stream plus the split-K slab buffers its GEMMs may use.
struct Lane { hipStream_t s; float* slabs; float* colsum_slabs; int lds_pad; int slots; };
inline Lane main_lane(fsmg_model* h) { return Lane{h->stream, h->slabs, h->colsum_slabs, 0, gemm_block_slots()}; }
// forward-only passes (validation: many rows per step, patch step kernel) tolerate one more overlapped GEMM block
// per CU than training steps do (measured at cfg-B: eval 2862 vs 2690 episodes/s, train 290 vs 303)
inline Lane aux_lane(fsmg_model* h, bool forward_only = false, bool persistent_chain = false) {
    const int cap = std::min(4, (persistent_chain ? h->aux_blocks_persist : h->aux_blocks_per_cu) + (forward_only && !h->aux_blocks_from_env ? 1 : 0));
    return Lane{h->aux, h->slabs2, h->colsum_slabs2, gemm_lds_pad_for(cap), 256 * cap};
}

// ------------------------------------------------------------------ kernel / order selection (api_schedule.hip)
bool use_ws_gemm(fsmg_model* h, int amode, int bmode, const GemmArgs& g, const Lane& ln);
bool use_h_gemm(fsmg_model* h, int amode, int bmode, const GemmArgs& g, const Lane& ln);

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
    hipStream_t s;                      // where flush() launches (the main stream unless the caller moves the batch aside)
    explicit OpBatch(fsmg_model* h_) : h(h_), s(h_->stream) { r.count = 0; }
    ~OpBatch() { if (h->slabs_owner == this) h->slabs_owner = nullptr; }
    OpBatch(const OpBatch&) = delete;
    OpBatch& operator=(const OpBatch&) = delete;
    int room(int n) { return (r.count + n > MULTI_MAX_OPS) ? flush() : FSMG_OK; }
    int add(void* p, uint32_t word, long long n_words, const int* cond = nullptr) {     // cond: fill only when *cond != 0 on the device
        if (n_words <= 0) return FSMG_OK;
        const int rc = room(1); if (rc != FSMG_OK) return rc;
        MultiOp& o = r.op[r.count++];
        o = MultiOp{}; o.kind = MULTI_FILL; o.dst = p; o.word = word; o.n = n_words; o.cond = cond;
        return FSMG_OK;
    }
    // out[i] = sum over the nslab slabs (fixed order); sq: squared-norm partials of out as a by-product
    // row_scale (without sq only): out[i] = row_scale[i / row_len] * sum
    int reduce(const float* slabs, long long stride, int nslab, float* out, long long n, double* sq = nullptr, const float* row_scale = nullptr, int row_len = 0) {
        if (n <= 0) return FSMG_OK;
        // the row scale is applied by k_multi_op's float4 branch only (elementwise.hip): refuse what would take another branch and come
        // out UNSCALED instead of queueing it (today Hp % 16 == 0 keeps every caller on that branch)
        if (row_scale != nullptr && (sq != nullptr || row_len <= 0 || ((n | stride | (long long)row_len) & 3) != 0))
            return fail(h, FSMG_ERR_STATE, "internal: a row-scaled slab sum needs n, stride and row_len multiples of 4 and no squared-norm partials");
        const int rc = room(1); if (rc != FSMG_OK) return rc;
        MultiOp& o = r.op[r.count++];
        o = MultiOp{}; o.kind = MULTI_REDUCE; o.dst = out; o.src = slabs; o.stride = stride; o.nslab = nslab; o.n = n; o.sq = sq;
        o.row_scale = row_scale; o.row_len = row_len;
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
        if (h->slabs_owner == this) h->slabs_owner = nullptr;      // the sums that were waiting for the main lane's slabs go out now
        if (r.count == 0) return FSMG_OK;
        HIPCK(h, launch_multi_op(s, r));
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
        // a replayed pass leaves the host-side facts its capture left: which BPTT family ran decides whether the update that follows
        // may clear the inbox-refill flag (the "up:...x / s" key and StepIncArgs::clear_ok)
        if (key[0] == 'f') h->last_bwd_xcd = c.bwd_xcd;
    }
    if (it == h->graphs.end()) {
        hipGraph_t graph = nullptr;
        const int64_t x0 = h->n_xcd_launches, p0 = h->n_persist_launches, s0 = h->n_step_launches;
        int rc; hipError_t e;
        {
            std::lock_guard<std::mutex> turn_lk(turn_mutex());        // nobody records on this handle's streams while they capture (take_turn)
            h->capturing.store(true);
            if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { h->capturing.store(false); return fail(h, FSMG_ERR_HIP, "hipStreamBeginCapture failed"); }
            rc = body();
            { auto& c = h->graph_counts[key]; c.xcd = h->n_xcd_launches - x0; c.persist = h->n_persist_launches - p0; c.step = h->n_step_launches - s0; c.bwd_xcd = h->last_bwd_xcd; }
            e = hipStreamEndCapture(h->stream, &graph);
            h->capturing.store(false);
        }
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
inline void phase_mark(fsmg_model* h, int i) {
    if (!h->ph_init) { for (auto& e : h->ph) hipEventCreate(&e); h->ph_init = true; }
    hipEventRecord(h->ph[i], h->stream);
}
inline void phase_report(fsmg_model* h) {
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

// the XCD-local kernels take this row count at this hidden size (and their buffers exist)
inline bool use_xcd(const fsmg_model* h, int B, bool backward = false) {
    if (h->Hp == 1024 && h->pair_mode < (backward ? 2 : 1)) return false;
    return h->persist && h->xcd && h->khx != nullptr && h->HX != nullptr && B <= h->xcd_max_rows && lstm_xcd_supported(B, h->Hp) &&
           lstm_xcd_hx_floats(B, h->T, h->Hp, h->xcd_bx3) <= h->hx_floats &&
           ((h->Hp == 1024 && !backward) || lstm_xcd_inbox_floats(B, h->Hp) <= h->inboxx_floats);
}
// where a train pass's cross entropy leaves dlogits (and the projection-gradient GEMMs read it)
inline float* dlogits_buf(const fsmg_model* h) { return h->inplace_dlogits ? h->logits : h->dlogits; }
// first XCD the packed recurrence leaves free
inline int xov_first_free(int B, int Hp = 512) {      // (hidden 1024: a weight copy takes an XCD pair)
    const int rpx = lstm_xcd16_packed_rows(B, Hp);
    return rpx > 0 ? (Hp == 1024 ? 2 : 1) * ((B + rpx - 1) / rpx) : 8;
}

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
void choose_schedule(fsmg_model* h, int B, bool train = false);
void xov_gate(fsmg_model* h, GemmArgs& g, int B);
int gemm_prepare_queue(fsmg_model* h, GemmArgs& g, int split, OpBatch* defer, bool* ok);
int select_xcd_format(fsmg_model* h, int B);
int gemm_restricted(fsmg_model* h, hipStream_t s, int amode, int bmode, GemmArgs g, int first, int* ctl);
int gemm_cleanup(fsmg_model* h, hipStream_t s, int amode, int bmode, GemmArgs g, int* ctl);

// ------------------------------------------------------------------ the step pieces
// api_forward.hip
int stage_tokens(fsmg_model* h, const int32_t* support, int n_sup, const int32_t* query, int n_qry, int on_device);
int reset_tok_table(fsmg_model* h);
int token_prep(fsmg_model* h, int n_sup, int n_qry, bool train = false);
int forward(fsmg_model* h, int B, int rows_per_group, int ngroups, float* loss_out, bool want_dlogits);
GemmArgs dw_args(fsmg_model* h, int B);        // api_backward.hip: dW = Hout^T dlogits, dd = colsum(dlogits) (forward() asks which kernel it will take)
// api_backward.hip
// part 0: the whole pass; part 1: up to and including the projection gradients; part 2: the rest (see backward())
int backward(fsmg_model* h, int B, int part = 0);
// api_update.hip
int repack_recurrent_weights(fsmg_model* h, hipStream_t s, const StepIncArgs* inc, bool* inc_done);
int repack_recurrent_weights(fsmg_model* h, hipStream_t s);
int ensure_khf(fsmg_model* h);
int ensure_cs(fsmg_model* h);           // the column-split copies of K_h are current (call OUTSIDE a graph capture, before a pass that reads them)
// a pass over B sequences reads the column-split copies: its recurrence does not take the XCD-local kernels in some direction
inline bool pass_reads_cs(const fsmg_model* h, int B, bool train) {
    return !(use_xcd(h, B) && h->persist_fwd) || (train && !(use_xcd(h, B, true) && h->persist_bwd));
}
int apply_update(fsmg_model* h, float grad_scale);
int sgd_update(fsmg_model* h, float lr);
int save_theta(fsmg_model* h);
int restore_theta(fsmg_model* h);
void on_timeout(fsmg_model* h);
void on_softmax_range(fsmg_model* h);
inline bool is_retry(int rc) { return rc == FSMG_ERR_TIMEOUT || rc == FSMG_ERR_SOFTMAX_RANGE; }
int poll_skipped(fsmg_model* h);
int report(fsmg_model* h, int what);
int check_tokens_and_read(fsmg_model* h, const float* d_src, float scale, float* host_out, int n, bool train_tail = false);
int after_update(fsmg_model* h, float grad_scale, float* loss);
// api_comm.hip
int exchange_gradients(fsmg_model* h);
void comm_destroy(fsmg_model* h);       // fsmg_destroy's share: communicator, its stream and event

}  // namespace fsmg_host
