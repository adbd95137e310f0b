// Clip + Adam update, inner-loop SGD, recurrent-weight repack, time-out bookkeeping, loss read-back.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

// host-side parameter writes (init / set_param / restore) leave the fragment-ordered weight copies stale
int repack_recurrent_weights(fsmg_model* h, hipStream_t s);
int ensure_khf(fsmg_model* h) {
    if (!h->khf_dirty) return FSMG_OK;
    const int rc = repack_recurrent_weights(h, h->stream);
    if (rc != FSMG_OK) return rc;
    h->khf_dirty = false;
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
        // the XCD-local kernels are this handle's recurrence: the column-split copies wait until somebody needs them (ensure_cs)
        // (never from inside a graph capture: a replay would not run this host code, and cs_stale would lie -- the repack that closes a
        // train step, inc != nullptr, is the one that can be captured; run_graphed's bypass condition says when it is not)
        const bool captured = inc != nullptr && h->cfg.use_graph && !(h->timing || h->ov_call || h->xov_call || h->eager_call);
        const bool lazy = h->lazy_cs && h->khx != nullptr && h->persist && h->xcd && !captured;
        a.mode = lazy ? 1 : 0;
        h->cs_stale = lazy;
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
    h->cs_stale = false;
    for (int l = 0; l < h->L; ++l) {
        HIPCK(h, launch_repack_kh(s, h->P + h->off_kh[l], h->khf + (size_t)(2 * l) * h->Hp * h->G4,
                                  h->khf + (size_t)(2 * l + 1) * h->Hp * h->G4, h->Hp));
        if (h->khx) HIPCK(h, launch_repack_kh_xcd(s, h->P + h->off_kh[l], h->khx + (size_t)(2 * l) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3),
                                                  h->khx + (size_t)(2 * l + 1) * lstm_xcd_weight_floats((int)h->Hp, h->xcd_bx3), h->Hp, h->xcd_bx3));
    }
    return FSMG_OK;
}
int repack_recurrent_weights(fsmg_model* h, hipStream_t s) { return repack_recurrent_weights(h, s, nullptr, nullptr); }

int ensure_cs(fsmg_model* h) {
    if (!h->cs_stale) return FSMG_OK;
    RepackAllArgs a{};
    a.n = h->L; a.Hp = h->Hp; a.bx3 = h->xcd_bx3 ? 1 : 0; a.mode = 2;
    for (int l = 0; l < h->L; ++l) {
        a.Kh[l] = h->P + h->off_kh[l];
        a.cf[l] = h->khf + (size_t)(2 * l) * h->Hp * h->G4; a.cb[l] = h->khf + (size_t)(2 * l + 1) * h->Hp * h->G4;
    }
    HIPCK(h, launch_repack_kh_all(h->stream, a, nullptr));
    h->cs_stale = false;
    return FSMG_OK;
}

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
    // Two launches where the pass is eager (nothing is being captured): [embedding .. the LSTM layers] here, [softmax_w, softmax_b] --
    // 56 % of the bytes at cfg-B -- on the auxiliary stream, where it runs beside the repack below and the NEXT step's token_prep,
    // x-part GEMM and fills, none of which touches those parameters.  The first launch publishes (go, clip scale, alpha); the
    // second takes them from there, because by the time it runs k_step_increment may have moved the step counter and cleared the
    // flags.  Same arithmetic on the same values: bit-identical to the single launch.  Whoever reads or writes the softmax
    // parameters / moments / gradients next goes through settle_pending() first (fsmg_model.h).
    const bool split = h->upd_split && h->eager_call && h->aux != nullptr && !h->timing && h->off_w > 0 && h->off_w < h->n_flat;
    if (split) {
        GEMMCK(settle_pending(h));           // (an earlier update's second half: cannot be pending here, but costs nothing to rule out)
        a.n = h->off_w; a.publish = h->d_decided;
        HIPCK(h, launch_adam_update(s, a));
        HIPCK(h, hipEventRecord(h->ev_upd_fork, s));
        HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_upd_fork, 0));
        UpdateArgs b = a;
        b.p = h->P + h->off_w; b.m = h->M + h->off_w; b.v = h->Vv + h->off_w; b.g = h->G + h->off_w; b.n = h->n_flat - h->off_w;
        b.publish = nullptr; b.consume = h->d_decided; b.gnorm_out = nullptr;
        HIPCK(h, launch_adam_update(h->aux, b));
        HIPCK(h, hipEventRecord(h->ev_upd, h->aux));
        h->upd_pending = true;
    } else {
        HIPCK(h, launch_adam_update(s, a));
    }
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
    HIPCK(h, launch_copy_words(h->stream, h->P_saved, h->P, h->n_flat));        // (n_flat is a multiple of FLAT_ALIGN = 64 floats)
    return FSMG_OK;
}
int restore_theta(fsmg_model* h) {
    HIPCK(h, launch_copy_words(h->stream, h->P, h->P_saved, h->n_flat));
    h->khf_dirty = true;
    return ensure_khf(h);
}

// a persistent recurrent kernel gave up waiting for its peers (its blocks were not co-resident): one launch per time step
// for the next `fallback_steps` train steps, then the persistent path is tried again
void on_timeout(fsmg_model* h) {
    ++h->n_timeouts;
    h->retry_armed = true;
    if (h->persist) {
        // reached from the asynchronous path too (after_update with loss == NULL polls the host-mapped tallies): later replays
        // of the same execs may still be queued or running, so drain both streams before the execs are destroyed
        h->persist = false;
        hipStreamSynchronize(h->stream);
        if (h->aux) hipStreamSynchronize(h->aux);
        drop_graphs(h);
    }
    h->fallback_left = h->fallback_steps;
    // ... or the self-check of the gated projection found words that differ from the serial recomputation (xov_selfcheck,
    // api_forward.hip): this handle keeps the serial order, at once
    if (h->host_counters && h->host_counters[3] != h->seen_selfcheck_mismatch) {
        h->seen_selfcheck_mismatch = h->host_counters[3];
        if (h->xov) {
            h->xov = false;
            fprintf(stderr, "[fsmg] XCD-partitioned order: %lld 16-byte words of the gated projection differed from the serial recomputation "
                            "(stale operand rows in an XCD's L2?): the step is repeated, serial order from here on\n", (long long)h->host_counters[3]);
        }
    }
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

// A row's sum of exp(logit) or its target's exp(logit) left the range the shift-free fused softmax is used for (k_ce_finish raised
// flag 4, k_step_increment tallied the skipped step in counters[5] -- on every rank of a data-parallel job: the indicator travels in the
// reduced tail): the handle takes the cross-entropy pass with the shifted softmax from here on.  Nothing timed out: the persistent
// kernels, the fallback period, the time-out tallies and the BPTT inboxes are left alone.
void on_softmax_range(fsmg_model* h) {
    h->retry_armed = true;
    if (h->host_counters) h->seen_softmax_range = h->host_counters[4];
    if (!h->fused_softmax) return;
    h->fused_softmax = false;
    hipStreamSynchronize(h->stream);
    if (h->aux) hipStreamSynchronize(h->aux);
    drop_graphs(h);                       // (a captured pass holds the fused path)
    fprintf(stderr, "[fsmg] fused softmax: a row's sum of exp(logit) or its target's exp(logit) left the fp32 range of the shift-free form (%lld rows so far, "
                    "this rank's): the step is repeated, cross-entropy pass with the shifted softmax from here on\n",
            h->host_counters ? (long long)h->host_counters[4] : 0LL);
}

// Compares the host-mapped tallies of k_step_increment with what this handle has already seen (no synchronisation: the
// caller decides whether the stream has been drained).  0 = nothing new, 2 = a train step was skipped after a time-out,
// 1 = after a token-range error.
int poll_skipped(fsmg_model* h) {
    if (!h->host_counters) return 0;
    const long long to = h->host_counters[0], tk = h->host_counters[1];
    int what = 0;
    const long long pf = h->host_counters[2], rg = h->host_counters[5];
    if (pf != h->seen_peer_failures) { h->seen_peer_failures = pf; what = 3; }
    if (rg != h->seen_range_skips) { h->seen_range_skips = rg; on_softmax_range(h); what = 4; }
    if (tk != h->seen_token_errors) { h->seen_token_errors = tk; what = 1; }
    if (to != h->seen_timeouts) { h->seen_timeouts = to; on_timeout(h); what = 2; }
    return what;
}

int report(fsmg_model* h, int what) {
    if (what == 2)
        return fail(h, FSMG_ERR_TIMEOUT, "persistent recurrent kernel timed out waiting for a peer block (blocks not co-resident); "
                                         "the step was skipped, this handle now uses one launch per time step: repeat the step");
    if (what == 4)
        return fail(h, FSMG_ERR_SOFTMAX_RANGE, "a row's sum of exp(logit) or its target's exp(logit) left the fp32 range of the shift-free fused softmax; "
                                               "the step was skipped, this handle now takes the cross-entropy pass with the shifted softmax: repeat the step");
    if (what == 1) return fail(h, FSMG_ERR_TOKEN_RANGE, "token id outside [0, input_size)");
    if (what == 3) return fail(h, FSMG_ERR_STATE, "a rank of the episode-parallel job failed before the gradient exchange: the step was skipped on every rank");
    return FSMG_OK;
}

int check_tokens_and_read(fsmg_model* h, const float* d_src, float scale, float* host_out, int n, bool train_tail) {
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
        else if (err == 4) on_softmax_range(h);
    }
    if (err) return report(h, err);
    for (int i = 0; i < n; ++i) host_out[i] = tmp[i] * scale;
    return FSMG_OK;
}

int after_update(fsmg_model* h, float grad_scale, float* loss) {
    h->have_grads = false;
    if (loss) return check_tokens_and_read(h, h->G + h->n_flat + 1, grad_scale, loss, 1, true);
    // no read-back: skipped steps of EARLIER calls that have retired by now are noticed here (a time-out switches the
    // handle to per-step launches; the skipped episodes stay skipped -- fsmg_get_stats counts them)
    const int what = poll_skipped(h);
    if (what == 1 || what == 3) return report(h, what);
    return FSMG_OK;         // (2 / 4: the handle has switched already; the skipped episode stays skipped -- fsmg_get_stats counts it)
}

}  // namespace fsmg_host
