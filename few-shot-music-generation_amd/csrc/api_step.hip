// Train / eval / MAML-style step entry points and the device-resident episode table.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

int validate_shape(fsmg_model* h, int N, int K, int Q) {
    if (N <= 0 || K < 0 || Q < 0 || (int64_t)N * (K + Q) <= 0 || (int64_t)N * (K + Q) > (1 << 20))
        return fail(h, FSMG_ERR_INVALID, "bad episode shape N/K/Q");
    return FSMG_OK;
}

// forward + backward of one episode whose tokens `stage` puts into the handle's staging buffer ([n_sup + n_qry][T], support rows first)
// with_update: the clip + Adam update (grad_scale 1) rides in the same captured graph -- the single-GPU train step; a
// gradient exchange between the two halves (episode-parallel training) needs them as separate calls
template <class Stage>
int forward_backward_core(fsmg_model* h, int32_t N, int32_t K, int32_t Q, Stage&& stage, bool with_update = false) {
    int rc = validate_shape(h, N, K, Q);
    if (rc != FSMG_OK) return rc;
    const int B = N * (K + Q);
    if (h->fallback_left > 0 && --h->fallback_left == 0 && h->persist != h->persist_cfg) {   // try the persistent path again
        h->persist = h->persist_cfg;
        drop_graphs(h);
    }
    if ((rc = ensure_scratch(h, B)) != FSMG_OK) return rc;
    if ((rc = select_xcd_format(h, B)) != FSMG_OK) return rc;
    choose_schedule(h, B, true);
    h->xov_last = h->xov_call;
    // an eager pass orders itself behind a pending update half inside forward(); a pass that is captured or replayed cannot hold
    // a wait on an event recorded outside the graph
    if (!h->eager_call && (rc = settle_pending(h)) != FSMG_OK) return rc;
    if ((rc = ensure_khf(h)) != FSMG_OK) return rc;
    if (pass_reads_cs(h, B, true) && (rc = ensure_cs(h)) != FSMG_OK) return rc;
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
        if (is_retry(rc) && h->retry_armed && attempt == 0) { h->retry_armed = false; continue; }
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
    if (is_retry(rc) && h->retry_armed) {
        // a persistent step kernel could not get all of its blocks resident (another workload holds the CUs): the
        // update kernels saw the flag and left parameters, Adam state and step counter alone, and the handle has
        // fallen back to one launch per time step -- repeat the step that way
        h->retry_armed = false;
        rc = forward_backward_core(h, N, K, Q, stage, true);
        if (rc == FSMG_OK) rc = after_update(h, 1.0f, loss);
    }
    return rc;
}

}  // namespace fsmg_host

// =========================================================================== C ABI
extern "C" {

int fsmg_forward_backward(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K,
                          int32_t Q, int32_t tokens_on_device) {
    if (!h || !support || !query) return FSMG_ERR_INVALID;
    BEGIN_CALL(h, true);        // the pass orders itself behind a pending update where it first reads the softmax parameters (forward())
    return forward_backward_core(h, N, K, Q, [&]() { return stage_tokens(h, support, N * K, query, N * Q, tokens_on_device); });
}

// ---- device-resident episode table: a split's packed [n_songs][T] token table lives in HBM and an
// episode is an index gather on the GPU (reference src/data/episode.py:62-74, src/data/dataset.py:187-199 fill the same
// rows from the host cache): a step uploads N*(K+Q) indices (180 B at cfg-B) instead of 23 KB of tokens.
int fsmg_upload_table(fsmg_handle h, int32_t table_id, const int32_t* host_table, int64_t n_songs) {
    if (!h || !host_table || table_id < 0 || table_id >= fsmg_model::MAX_TABLES || n_songs <= 0 || n_songs > (1LL << 30) / std::max(1, h->T))
        return h ? fail(h, FSMG_ERR_INVALID, "bad table id / size") : FSMG_ERR_INVALID;
    BEGIN_CALL(h);
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
    BEGIN_CALL(h, true);        // the pass orders itself behind a pending update where it first reads the softmax parameters (forward())
    return forward_backward_core(h, N, K, Q, [&]() { return stage_indexed(h, table_id, support_idx, N * K, query_idx, N * Q); });
}

int fsmg_train_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                            int32_t N, int32_t K, int32_t Q, float* loss) {
    if (!h || !support_idx || !query_idx) return FSMG_ERR_INVALID;
    BEGIN_CALL(h, true);        // the pass orders itself behind a pending update where it first reads the softmax parameters (forward())
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
    BEGIN_CALL(h);
    if (!h->have_grads) return fail(h, FSMG_ERR_STATE, "no backward pass is pending");
    HIPCK(h, hipStreamWaitEvent((hipStream_t)stream, h->ev_bucket[bucket == 0 ? 0 : 1], 0));
    return FSMG_OK;
}

int fsmg_apply_update(fsmg_handle h, float grad_scale, float* loss) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    if (!h->have_grads) return fail(h, FSMG_ERR_STATE, "fsmg_apply_update without a preceding fsmg_forward_backward");
    if (!(grad_scale > 0.f)) return fail(h, FSMG_ERR_INVALID, "grad_scale must be > 0");
    uint32_t bits; std::memcpy(&bits, &grad_scale, 4);
    int rc = run_graphed(h, "up:" + std::to_string(bits) + (h->last_bwd_xcd ? "x" : "s"), [&]() -> int { return apply_update(h, grad_scale); });
    if (rc != FSMG_OK) return rc;
    return after_update(h, grad_scale, loss);
}

int fsmg_train_step(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                    int32_t tokens_on_device, float* loss) {
    if (!h || !support || !query) return FSMG_ERR_INVALID;
    BEGIN_CALL(h, true);        // the pass orders itself behind a pending update where it first reads the softmax parameters (forward())
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
    BEGIN_CALL(h);
    int rc = maml_adapt(h, support, N, K, inner_steps, inner_lr, tokens_on_device);
    if (rc == FSMG_OK) rc = fsmg_forward_backward(h, query, query, N, Q, 0, tokens_on_device);     // query rows at theta'
    const int rc2 = h->P_saved ? restore_theta(h) : FSMG_OK;                                         // theta comes back whatever happened
    return rc != FSMG_OK ? rc : rc2;
}

int fsmg_maml_step(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* loss) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    if (h->comm != nullptr)           // per-rank inner loop (no communication), query gradients exchanged like a plain step's
        return dp_train_step(h, loss, [&]() { return fsmg_maml_forward_backward(h, support, query, N, K, Q, inner_steps, inner_lr, tokens_on_device); });
    int rc = fsmg_maml_forward_backward(h, support, query, N, K, Q, inner_steps, inner_lr, tokens_on_device);
    if (rc != FSMG_OK) return rc;
    rc = fsmg_apply_update(h, 1.0f, loss);
    if (is_retry(rc) && h->retry_armed) {        // same recovery as fsmg_train_step: nothing was updated, repeat per step
        h->retry_armed = false;
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
    BEGIN_CALL(h);
    const int rc = gather_episode(h, table_id, support_idx, N * K, query_idx, N * Q);
    if (rc != FSMG_OK) return rc;
    return fsmg_maml_forward_backward(h, h->d_gather, h->d_gather + (size_t)N * K * h->T, N, K, Q, inner_steps, inner_lr, 1);
}

int fsmg_maml_step_indexed(fsmg_handle h, int32_t table_id, const int32_t* support_idx, const int32_t* query_idx,
                           int32_t N, int32_t K, int32_t Q, int32_t inner_steps, float inner_lr, float* loss) {
    if (!h || !support_idx || !query_idx || N <= 0 || K <= 0 || Q <= 0) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    const int rc = gather_episode(h, table_id, support_idx, N * K, query_idx, N * Q);
    if (rc != FSMG_OK) return rc;
    return fsmg_maml_step(h, h->d_gather, h->d_gather + (size_t)N * K * h->T, N, K, Q, inner_steps, inner_lr, 1, loss);
}

int fsmg_maml_eval(fsmg_handle h, const int32_t* support, const int32_t* query, int32_t N, int32_t K, int32_t Q,
                   int32_t inner_steps, float inner_lr, int32_t tokens_on_device, float* nll) {
    if (!h || !support || !query || !nll) return FSMG_ERR_INVALID;
    if (inner_steps < 0 || inner_steps > 64 || !(inner_lr >= 0.f) || K <= 0 || Q <= 0) return fail(h, FSMG_ERR_INVALID, "bad inner_steps / inner_lr / K / Q");
    BEGIN_CALL(h);
    int rc = FSMG_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = maml_adapt(h, support, N, K, inner_steps, inner_lr, tokens_on_device);
        // the flag of a time-out / token error inside the adaptation is still set: fsmg_eval_step reads and reports it
        if (rc == FSMG_OK) rc = fsmg_eval_step(h, query, N, Q, tokens_on_device, nll);
        const int rc2 = h->P_saved ? restore_theta(h) : FSMG_OK;
        h->have_grads = false;
        if (rc == FSMG_OK) rc = rc2;
        if (!(is_retry(rc) && h->retry_armed)) break;
        h->retry_armed = false;                        // adapted with garbage (skipped) steps: repeat on per-step launches
    }
    return rc;
}

int fsmg_eval_batch(fsmg_handle h, const int32_t* queries, int32_t n_episodes, int32_t N, int32_t Q,
                    int32_t tokens_on_device, float* nll) {
    if (!h || !queries || !nll || n_episodes <= 0) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
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
        if (pass_reads_cs(h, B, false) && (rc = ensure_cs(h)) != FSMG_OK) return rc;
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
        if (is_retry(rc) && h->retry_armed && !retried) {
            // a persistent kernel could not get its blocks resident: the handle has switched to one launch per time
            // step; repeat this chunk that way (validation must not abort a training run, nor leave peer ranks hanging)
            h->retry_armed = false;
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

}  // extern "C"
