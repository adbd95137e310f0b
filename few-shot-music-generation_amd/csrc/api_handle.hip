// Handle life cycle (fsmg_create / fsmg_destroy), knobs, statistics, greedy decode.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include <mutex>
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

thread_local std::string g_create_error;

// The auxiliary stream has the DEFAULT priority.  Rounds 1-4 created it with the lowest one (the chains' kernels were to win the
// dispatcher); measured in round 5 (profiles/r05_hw_queue_probe.txt): with five or more handles alive in a process, every odd one
// from the fifth on ran 32-45 % slower -- every kernel of its step 20-50 us longer -- as soon as a step touched its non-default-
// priority stream (lowest or highest alike; the serial order with only the backward tail on it: 1.83 -> 2.49 ms), and at the default
// priority nothing is lost anywhere (cfg-B / C / D / ref-default / 360-row batches, train and validation: within +-0.3 %).
// `aux` must run BESIDE the handle's main stream (fsmg_model::aux_tries).  The probe: a wave on the main stream polls a flag for up to
// 300 us, a kernel on the candidate stream sets it; if the waiter gives up the two streams share a hardware queue -- keep the
// candidate allocated (so that the next one the runtime hands out sits on another queue), draw another, at most `max_tries` times.
static int pick_concurrent_aux(fsmg_model* h, int priority, int max_tries) {
    if (max_tries <= 0) {       // FSMG_AUX_TRIES=0: no probe, the first stream is taken as it comes (counter passes: a profiler that serialises
        // dispatches makes every candidate look like a shared queue; tools/pmc_passes.sh wants the partitioned order's kernels on record)
        if (hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, priority) != hipSuccess) return fail(h, FSMG_ERR_HIP, "aux stream create failed");
        h->aux_tries = 0;
        return FSMG_OK;
    }
    int* d = nullptr;
    if (hipMalloc((void**)&d, 256) != hipSuccess) return fail(h, FSMG_ERR_NOMEM, "hipMalloc(queue probe) failed");
    // does `cand` run beside the main stream?  1 / 0, or -1 when the probe itself failed.  `ticks`: how long the waiter polls (100 MHz)
    auto probe = [&](hipStream_t cand, long long ticks) -> int {
        int seen = 0;
        const int init[2] = {0, -1};
        if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess ||
            launch_queue_probe(h->stream, d, d + 1, 0, ticks) != hipSuccess || launch_queue_probe(cand, d, d + 1, 1, 0) != hipSuccess ||
            hipStreamSynchronize(cand) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
            hipMemcpy(&seen, d + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return seen == 1 ? 1 : 0;
    };
    std::vector<hipStream_t> rejected;
    int rc = FSMG_OK, found = 0;
    for (int t = 1; t <= max_tries && rc == FSMG_OK && !found; ++t) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, priority) != hipSuccess) { rc = fail(h, FSMG_ERR_HIP, "aux stream create failed"); break; }
        const int seen = probe(cand, 30000);
        if (seen < 0) { hipStreamDestroy(cand); rc = fail(h, FSMG_ERR_HIP, "queue probe failed"); break; }
        if (seen == 1) { h->aux = cand; h->aux_tries = t; found = 1; }
        else rejected.push_back(cand);
    }
    // Nobody ran beside the waiter within 300 us: either the process's streams really share the hardware queues, or the GPU was busy
    // with somebody else's work at that moment (another rank or process on the same device, shared CI) and the setter simply was not
    // scheduled in time (a false negative here switched the XCD-partitioned order off for the handle's whole life).  Ask the
    // same candidates once more with a 5 ms window before settling for the serial order.
    for (size_t i = 0; rc == FSMG_OK && !found && i < rejected.size(); ++i) {
        const int seen = probe(rejected[i], 500000);
        if (seen < 0) { rc = fail(h, FSMG_ERR_HIP, "queue probe failed"); break; }
        if (seen == 1) { h->aux = rejected[i]; h->aux_tries = (int)(max_tries + i + 1); found = 1; rejected.erase(rejected.begin() + (long)i); }
    }
    if (rc == FSMG_OK && !found) {          // no stream of this process runs beside the main one: keep one for the API's sake, remember it is serial
        h->aux = rejected.back(); rejected.pop_back();
        h->aux_tries = -1;
    }
    for (hipStream_t s : rejected) hipStreamDestroy(s);
    hipFree(d);
    return rc;
}

// The second launch of the last clip + Adam update (softmax_w, softmax_b and their moments, on the auxiliary stream) may still be in
// flight: order the main stream behind it.  One event wait, no host synchronisation.
int settle_pending(fsmg_model* h) {
    if (!h->upd_pending) return FSMG_OK;
    h->upd_pending = false;
    HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_upd, 0));
    return FSMG_OK;
}

std::mutex& turn_mutex() { static std::mutex mu; return mu; }
namespace {
fsmg_model* g_turn_last[16] = {};         // per device: the handle whose call returned last
bool turnstile_on() { static const bool on = !(std::getenv("FSMG_TURNSTILE") && std::atoi(std::getenv("FSMG_TURNSTILE")) == 0); return on; }
}
// (begin_call) behind what the handle that had the device before this one has issued so far
static int take_turn(fsmg_model* h) {
    if (!turnstile_on() || h->ev_turn == nullptr) return FSMG_OK;
    std::lock_guard<std::mutex> lk(turn_mutex());
    fsmg_model*& last = g_turn_last[h->device & 15];
    // a stream that is capturing takes no event record from outside its capture: the library's own captures hold turn_mutex() from
    // BeginCapture to EndCapture (run_graphed), a caller capturing the stream it handed over in fsmg_config is asked here
    auto capturing = [](hipStream_t s) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        return s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
    };
    if (last != nullptr && last != h && last->stream != nullptr && !last->capturing.load() && !capturing(last->stream) && !capturing(last->aux)) {
        HIPCK(h, hipEventRecord(last->ev_turn, last->stream));
        HIPCK(h, hipEventRecord(last->ev_turn_aux, last->aux != nullptr ? last->aux : last->stream));
        HIPCK(h, hipStreamWaitEvent(h->stream, last->ev_turn, 0));
        HIPCK(h, hipStreamWaitEvent(h->stream, last->ev_turn_aux, 0));
        if (h->aux != nullptr) {             // (this handle's second stream forks behind its first inside a call; what an earlier call left on it does not)
            HIPCK(h, hipStreamWaitEvent(h->aux, last->ev_turn, 0));
            HIPCK(h, hipStreamWaitEvent(h->aux, last->ev_turn_aux, 0));
        }
    }
    last = h;
    return FSMG_OK;
}
static void forget_turn(fsmg_model* h) {          // fsmg_destroy: nobody may wait on this handle's events any more
    std::lock_guard<std::mutex> lk(turn_mutex());
    if (g_turn_last[h->device & 15] == h) g_turn_last[h->device & 15] = nullptr;
}

int begin_call(fsmg_model* h, bool keep_pending) {
    HIPCK(h, hipSetDevice(h->device));
    const int rc = take_turn(h);
    if (rc != FSMG_OK) return rc;
    return keep_pending ? FSMG_OK : settle_pending(h);
}

}  // namespace fsmg_host

// =========================================================================== C ABI
extern "C" {

int fsmg_version(void) { return FSMG_VERSION; }

const char* fsmg_last_error(fsmg_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

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
        if (const char* e = std::getenv("FSMG_XOV_PARTS")) h->xov_parts = std::max(1, std::min(7, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_EAGER")) h->eager = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_MERGE_DK")) h->merge_dk = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_UPD_SPLIT")) h->upd_split = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_TAIL_ASIDE")) h->tail_aside = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_LAZY_CS")) h->lazy_cs = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_FUSED_SOFTMAX")) h->fused_softmax = (e[0] != '0');

        if (const char* e = std::getenv("FSMG_INPLACE_DLOGITS")) h->inplace_dlogits = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_XOV_SELFCHECK")) h->xov_selfcheck_left = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XOV_SELFCHECK_EVERY")) h->xov_selfcheck_every = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("FSMG_PERSISTENT")) h->persist = (e[0] != '0');
        h->persist_cfg = h->persist;
        if (const char* e = std::getenv("FSMG_FALLBACK_STEPS")) h->fallback_steps = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XCD")) h->xcd = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_XCD_PAIR")) h->pair_mode = std::max(0, std::min(2, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_FWD_RT")) h->force_fwd_rt = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_CHAIN_SPIN_LIMIT")) h->chain_spin_limit = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XCD_VARIANT")) h->xcd_variant = std::atoi(e);      // XCD_* bits, both directions (tests: the non-default paths)
        if (const char* e = std::getenv("FSMG_XCD_VARIANT_BWD")) h->xcd_variant_bwd = std::atoi(e);
#ifdef FSMG_EXPERIMENTS         // settled A/Bs: tuning values and rejected alternatives, experiment builds only
        if (const char* e = std::getenv("FSMG_CE_TAIL")) h->ce_tail = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_CE_TAIL_BLOCKS")) h->ce_tail_blocks = std::max(1, std::min(4096, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_XOV_DW_SPLIT")) h->xov_dw_split = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("FSMG_XOV_PUB")) h->xov_pub = std::max(1, std::min(64, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_FILL_EARLY")) h->fill_early = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_FILLS_LATE")) h->fills_late = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_XOV_TAIL")) h->xov_tail = std::max(0, std::min(64, std::atoi(e)));
        if (const char* e = std::getenv("FSMG_BWD_RS")) h->bwd_rs = (e[0] != '0');
        if (const char* e = std::getenv("FSMG_DP_SPLIT")) h->dp_split = std::max(0, std::min(2, std::atoi(e)));
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
        if (hipEventCreateWithFlags(&h->ev_turn, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_turn_aux, hipEventDisableTiming) != hipSuccess) return bail(FSMG_ERR_HIP, "event create failed");
        if (turnstile_on()) {      // the probe below asks whether two streams of this handle run side by side: not while another handle's pass has the chip
            std::lock_guard<std::mutex> lk(turn_mutex());
            fsmg_model* last = g_turn_last[h->device & 15];
            if (last != nullptr && last->stream != nullptr && !last->capturing.load()) { (void)hipStreamSynchronize(last->stream); if (last->aux) (void)hipStreamSynchronize(last->aux); }
        }
        {
            static const int tries = std::getenv("FSMG_AUX_TRIES") ? std::max(0, std::min(32, std::atoi(std::getenv("FSMG_AUX_TRIES")))) : 8;
            if (pick_concurrent_aux(h, 0, tries) != FSMG_OK) { std::string e = h->err; return bail(FSMG_ERR_HIP, e); }
            if (h->aux_tries < 0) {
                fprintf(stderr, "[fsmg] no second stream of this process runs beside the handle's stream (%d candidates, probed for 300 us and again for 5 ms: "
                                "they share its hardware queue -- GPU_MAX_HW_QUEUES? -- or the device was busy with another process): serial order, no overlapped "
                                "tails for this handle (fsmg_stats.aux_stream_tries = -1; fsmg_debug_set(\"reprobe_aux\", 1) asks again)\n", tries);
                h->overlap = false; h->overlap_forced = true; h->tail_aside = false; h->upd_split = false;
            }
        }
#ifdef FSMG_EXPERIMENTS
        if (h->ce_tail && (hipStreamCreateWithPriority(&h->aux2, hipStreamNonBlocking, 0) != hipSuccess ||
                           hipEventCreateWithFlags(&h->ev_ce_fork, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&h->ev_ce, hipEventDisableTiming) != hipSuccess)) return bail(FSMG_ERR_HIP, "aux2 stream create failed");
#endif
        for (int c = 0; c < fsmg_model::NCHUNK; ++c)
            if (hipEventCreateWithFlags(&h->ev_chunk[c], hipEventDisableTiming) != hipSuccess) return bail(FSMG_ERR_HIP, "event create failed");
        if (hipEventCreateWithFlags(&h->ev_bucket[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_bucket[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_upd_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_upd, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_side_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming) != hipSuccess ||

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
                               sizeof(int) * 2 * tok_words + sizeof(int) * prog_words + sizeof(int) * fsmg_model::XOV_DONE;
    if (hipMalloc((void**)&small, small_bytes) != hipSuccess) return bail(FSMG_ERR_NOMEM, "hipMalloc(scalars) failed");
    hipMemsetAsync(small, 0, small_bytes, h->stream);
    h->d_step = (long long*)small; h->d_err = (int*)(small + 256); h->d_gnorm = (float*)(small + 512);
    h->d_inbox_dirty = (int*)(small + 768);
    h->d_decided = (float*)(small + 896);
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
    h->xov_done = h->xov_prog + prog_words;
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
            // Hidden 1024 (round 6): the order EXISTS there (FSMG_SCHEDULE_XCD_PARTITIONED / FSMG_XCD_OVERLAP=1: the top layer's chains on three
            // XCD pairs, 15 rows each, the projection / dW on the fourth; FSMG_XOV_PARTS bit 4: dK of layer l + 1 beside the BPTT chain of
            // layer l) and is NOT what AUTO takes: measured at cfg-C against the serial order on the same kernels (profiles/r06_cfgC_xov_ab.txt)
            // 408.6 -> 403.6 (dW pair only), 392.4 (+ dK pair), 385.3 (+ forward pair).  One pair is a quarter of the chip: it absorbs at
            // most a quarter of a chain's length in GEMM time (~90 us), the packed chain pays a fourth row group for the 15th row (+30 us),
            // and what the pair has not drawn when the chain ends is rounds of whole 256 x 256 tiles (95-190 us) on the whole chip.
            if (cfg->schedule == FSMG_SCHEDULE_AUTO && std::getenv("FSMG_XCD_OVERLAP") == nullptr && !h->overlap_forced) h->xov = eligible;
            if (h->Hp == 1024 && std::getenv("FSMG_XOV_PARTS") == nullptr) h->xov_parts = 2;
        }
        h->xov_eligible = h->xov;
        if (h->aux_tries < 0) h->xov = false;           // its two launches would run one after the other (pick_concurrent_aux)
        // the XCD-partitioned schedule packs the rows on ceil(B / 16) XCDs: only the bf16-split kernels take 16 rows per XCD at one
        // MFMA phase's cost
        if (h->xov && h->bx3 && h->Hp == 512) h->xcd_bx3 = true;
        // (hidden 1024: the bf16-split pair kernels share the arithmetic of the bf16-split GEMMs; a handle on the fp32 MFMA keeps the fp32 chains)
        if (h->Hp == 1024 && !h->bx3) h->xcd_bx3 = false;
        if (const char* e = std::getenv("FSMG_XCD_BX3")) { h->xcd_bx3 = std::atoi(e) != 0 && (h->Hp == 512 || h->Hp == 1024); h->xcd_bx3_forced = true; }
        // (hidden 1024: room for either format -- the one in use follows the row count of the train passes, select_xcd_format)
        if (hipMalloc((void**)&h->khx, sizeof(float) * (size_t)h->L * 2 * lstm_xcd_weight_floats(h->Hp, h->xcd_bx3 || h->Hp == 1024)) != hipSuccess)
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
    (void)begin_call(h);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->aux) hipStreamSynchronize(h->aux);
    forget_turn(h);
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
    if (h->ev_upd_fork) hipEventDestroy(h->ev_upd_fork);
    if (h->ev_upd) hipEventDestroy(h->ev_upd);
    if (h->ev_side_fork) hipEventDestroy(h->ev_side_fork);
    if (h->ev_side) hipEventDestroy(h->ev_side);
    if (h->ev_turn) hipEventDestroy(h->ev_turn);
    if (h->ev_turn_aux) hipEventDestroy(h->ev_turn_aux);
#ifdef FSMG_EXPERIMENTS
    if (h->ev_ce_fork) hipEventDestroy(h->ev_ce_fork);
    if (h->ev_ce) hipEventDestroy(h->ev_ce);
    if (h->aux2) { hipStreamSynchronize(h->aux2); hipStreamDestroy(h->aux2); }
#endif
    comm_destroy(h);
    if (h->probe) { hipStreamSynchronize(h->probe); hipStreamDestroy(h->probe); }
    if (h->d_probe) hipFree(h->d_probe);
    if (h->aux) hipStreamDestroy(h->aux);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return FSMG_OK;
}

int fsmg_synchronize(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    return FSMG_OK;
}

int fsmg_sample(fsmg_handle h, int32_t num, int32_t* out_tokens) {
    if (!h || num < 0 || (num > 0 && !out_tokens)) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
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
    BEGIN_CALL(h);
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
    BEGIN_CALL(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    poll_skipped(h);
    std::memset(out, 0, sizeof(*out));
    out->timeouts = h->n_timeouts;
    out->steps_skipped_timeout = h->host_counters ? h->host_counters[0] : 0;
    out->steps_skipped_token_range = h->host_counters ? h->host_counters[1] : 0;
    out->steps_skipped_peer_failure = h->host_counters ? h->host_counters[2] : 0;
    out->xov_selfcheck_mismatches = h->host_counters ? h->host_counters[3] : 0;
    out->softmax_range_rows = h->host_counters ? h->host_counters[4] : 0;
    out->steps_skipped_softmax_range = h->host_counters ? h->host_counters[5] : 0;
    out->aux_stream_tries = h->aux_tries; out->reserved0 = 0;
    out->xcd_launches = h->n_xcd_launches;
    out->persistent_launches = h->n_persist_launches;
    out->step_launches = h->n_step_launches;
    out->persistent_path = h->persist ? 1 : 0;
    out->fallback_steps_left = h->fallback_left;
    return FSMG_OK;
}

int fsmg_debug_set(fsmg_handle h, const char* what, int64_t value) {
    if (!h || !what) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->aux) HIPCK(h, hipStreamSynchronize(h->aux));
    if (!std::strcmp(what, "chain_spin_limit")) h->chain_spin_limit = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (!std::strcmp(what, "fallback_steps")) h->fallback_steps = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30));
    else if (!std::strcmp(what, "eager")) h->eager = value != 0;
    else if (!std::strcmp(what, "persistent")) { h->persist = h->persist_cfg = value != 0; h->fallback_left = 0; }
    else if (!std::strcmp(what, "inplace_dlogits")) h->inplace_dlogits = value != 0;
    else if (!std::strcmp(what, "upd_split")) h->upd_split = value != 0;
    else if (!std::strcmp(what, "reprobe_aux")) {
        // the create-time probe found no stream running beside the handle's own (aux_stream_tries = -1), possibly because the device was
        // busy with somebody else's work at that moment: ask again.  On success the overlapped tails come back; the XCD-partitioned order
        // only where the handle was created with the recurrent-kernel format it needs (xov_eligible: decided at creation)
        if (value != 0 && h->aux_tries < 0) {
            HIPCK(h, hipStreamSynchronize(h->stream));
            if (h->aux) HIPCK(h, hipStreamSynchronize(h->aux));
            hipStream_t old = h->aux; h->aux = nullptr;
            const int rc = pick_concurrent_aux(h, 0, 8);
            if (rc != FSMG_OK) { if (h->aux == nullptr) h->aux = old; return rc; }
            if (old) hipStreamDestroy(old);
            if (h->aux_tries > 0) {
                h->overlap = true; h->overlap_forced = false; h->tail_aside = true;
                if (h->xov_eligible && h->xcd_bx3) h->xov = true;
                drop_graphs(h);
            }
        }
    }
    else if (!std::strcmp(what, "tail_aside")) h->tail_aside = value != 0;
    else if (!std::strcmp(what, "fused_softmax")) h->fused_softmax = value != 0;

    else if (!std::strcmp(what, "xov_selfcheck")) h->xov_selfcheck_left = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else if (!std::strcmp(what, "xov_selfcheck_fault")) h->xov_selfcheck_fault = value != 0;
    else if (!std::strcmp(what, "xov_selfcheck_every")) h->xov_selfcheck_every = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else return fail(h, FSMG_ERR_NAME, std::string("unknown knob '") + what + "'");
    drop_graphs(h);
    return FSMG_OK;
}

}  // extern "C"
