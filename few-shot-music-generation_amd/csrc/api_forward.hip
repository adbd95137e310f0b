// Forward pass builder: token staging, x-part GEMMs, recurrent chains, projection + cross entropy in the order choose_schedule picked.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

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
int token_prep(fsmg_model* h, int n_sup, int n_qry, bool train) {
    const bool table = train && h->tok_first != nullptr;
    HIPCK(h, launch_token_prep(h->stream, h->cur_sup, n_sup, h->cur_qry, n_qry, h->T, h->V, h->V,
                               h->X, h->Y, h->d_err, table ? h->tok_first : nullptr, table ? h->tok_count : nullptr));
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
// fused softmax of a train pass (fsmg_model::fused_softmax): the projection's epilogue leaves E = exp(logit) + the per-slice partials ...
inline void fused_softmax_args(fsmg_model* h, GemmArgs& g) {
    g.ce_part = h->ce_part; g.ce_nvocab = h->V1; g.ce_store = 1;      // (no target lookup in the epilogue: k_ce_finish reads the target's E)
}
// ... and one kernel turns them into lse, ce, the row scales and c_r * h_r, and patches E[r][y_r] (all rows of the pass)
int ce_finish(fsmg_model* h, hipStream_t s, int B, int64_t rows) {
    ScopedTimer tm(h, "ce");
    HIPCK(h, launch_ce_finish(s, h->ce_part, h->ce_nparts, h->Y, (int)rows, (float)(1.0 / ((double)rows + 1e-12)), h->logits, h->V1p,
                              h->lse, h->ce, h->crow, h->Hs[h->L - 1] + (size_t)B * h->Hp, h->Hsc, h->Hp, h->d_err, h->d_counters + 4));
    return FSMG_OK;
}
int ce_rows(fsmg_model* h, hipStream_t s, int B, int t0, int t1, int64_t rows_total) {
    ScopedTimer tm(h, "ce");
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    // (a row is read into registers as a whole before any of it is written back: dlogits may be the logits buffer itself)
    HIPCK(h, launch_ce_rows(s, h->logits + (size_t)r0 * h->V1p, h->V1p, (int)m, h->V1, h->Y + r0, h->lse + r0,
                            h->ce + r0, dlogits_buf(h) + (size_t)r0 * h->V1p, (float)(1.0 / ((double)rows_total + 1e-12))));
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

// Self-check of the gated projection.  Under the XCD-partitioned order a tile's A rows are written by a kernel on
// OTHER XCDs while this kernel runs; the consumer reads them with ordinary loads behind a relaxed poll of the row's progress counter,
// the producer publishes with write-through (sc1) stores drained by s_waitcnt vmcnt(0) in front of a relaxed atomic (lstm_xcd.hip).
// That is correct as long as (a) an XCD's L2 holds no line of those rows from before the launch (the runtime's acquire at kernel
// start) and (b) no line enters it before its row's gate (nothing in the kernel touches a row early; rows are 2 KB, tiles 512 KB:
// no shared lines) -- properties of this runtime and firmware (validated on ROCm 7.2.0 / gfx950), not of the
// programming model, and a stale line would be SILENT.  So the first passes of every handle under this order compute the logits a
// second time with the same kernel as a plain launch behind the chain (same k order, same bits: tests/test_gpu_parity.py
// test_xcd_partitioned_schedule_gives_the_same_bits) into the spare dlogits buffer and compare the words: a difference raises the
// time-out flag (the step is skipped and repeated on per-step launches like any timed-out step), is tallied in
// fsmg_stats.xov_selfcheck_mismatches, and on_timeout() parks the order for the handle.  ~0.4 ms per checked pass.
int xov_selfcheck(fsmg_model* h, int B) {
    --h->xov_selfcheck_left;
    ++h->xov_selfcheck_runs;
    const int64_t rows = (int64_t)h->T * B;
    if (!h->xov_selfcheck_fault) {
        GemmArgs g = logits_args(h, B, 0, h->T);
        if (h->fs_call) fused_softmax_args(h, g);        // (the same epilogue as the queue launch: E values; the partials it rewrites are the same numbers)
        g.C = h->dlogits; g.bx3 = 3; g.ksplit = 1;
        HIPCK(h, launch_gemm(h->stream, OP_KC, OP_XC, g, 0));
    }
    HIPCK(h, launch_compare_words(h->stream, h->logits, h->dlogits, rows * h->V1p, h->d_err, h->d_counters + 3));
    return FSMG_OK;
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
    if (!xcd && h->cs_stale) return fail(h, FSMG_ERR_STATE, "internal: forward pass on the column-split kernels with stale fragment copies of K_h (ensure_cs not called)");
    const int nch_ov = ov ? ((chain || xcd) ? h->nchunk_persist : h->nchunk) : 1;
    // XCD-partitioned schedule: the chain packed on the first XCDs publishes the time steps it has finished, the projection's
    // row tiles are drawn by the other XCDs as their rows arrive (and by the whole chip once the chain is over)
    // The queue takes the rows of the time steps [0, t_cut); the last few steps' rows (complete only when the chain is) go to a
    // chip-wide launch of the 128-tile kernel behind it: a 256 x 256 tile is 85 us of latency with 1/6 of the CUs busy, the same
    // rows as 128 x 128 tiles are one under-full round of ~40 us (same bits: the kernels share k order and term order, K = H is never split)
    // (t_cut is a multiple of the publishing period: the queue's last row tile then waits for a step that IS published)
    const int t_cut = (T >= 4 * h->xov_tail && h->xov_tail > 0) ? (T - h->xov_tail) / h->xov_pub * h->xov_pub : T;
    GemmArgs ghead = logits_args(h, B, 0, t_cut);
    // Fused softmax: where dlogits would be written in place, nothing overlaps on a second stream chunk by chunk, and the weight
    // gradient of the projection runs on the 256 x 256-tile kernel (the one with weighted column sums).  The projection itself may be
    // any of the bf16-split kernels: they share the epilogue (store_tile_at) and the layout of the partials.
    h->fs_call = false;
    if (want_dlogits && h->fused_softmax && !ov && h->bx3 && h->inplace_dlogits && h->Hsc != nullptr && t_cut == T && mainl.lds_pad == 0) {
        h->fs_call = true;                   // (dw_args reads it)
        h->fs_call = use_h_gemm(h, OP_XC, OP_XC, dw_args(h, B), mainl);
    }
    if (h->fs_call) fused_softmax_args(h, ghead);
    const bool xov = h->xov_call && (h->xov_parts & 1) && xcd && want_dlogits && !ov && xov_fits(ghead);
#ifdef FSMG_EXPERIMENTS
    // the cross entropy under the pair's tail (fsmg_model::ce_tail, measured and rejected): not in a pass that self-checks the logits first
    const int tiles_m = (int)((rows + 255) / 256);
    const bool ce_tail = xov && !h->fs_call && h->ce_tail && h->aux2 != nullptr && h->xov_selfcheck_left <= 0 && t_cut == T && !h->timing &&
                         h->V1p <= 12 * 1024 && tiles_m < fsmg_model::XOV_DONE;      // (the last word of xov_done is the row counter)
    if (ce_tail) ghead.done = h->xov_done;
#else
    constexpr bool ce_tail = false;
#endif
    const int xfree = xov_first_free(B, Hp);
    const int rpx = xov ? lstm_xcd16_packed_rows(B, Hp) : 0;
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
                    if (ce_tail) GEMMCK(fills.add(h->xov_done, 0u, fsmg_model::XOV_DONE));
                }
            }
            return fills.flush();
        };
        if (!h->fill_early || xov) GEMMCK(chain_fills());
        // the softmax half of the previous update may still be running on the auxiliary stream (apply_update): what has been issued
        // so far -- token_prep, the bottom layer's x-part GEMM, the fills -- reads none of it; everything from here on may
        if (l == 0) GEMMCK(settle_pending(h));
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
                a.variant = h->xcd_variant >= 0 ? h->xcd_variant : lstm_xcd_default_variant(B, true, Hp, a.rpx, h->xcd_bx3);
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
#ifdef FSMG_EXPERIMENTS
            if (ce_tail) {         // the gated cross entropy starts when the chain is over, beside what is left of the queue
                HIPCK(h, hipEventRecord(h->ev_ce_fork, s));
                HIPCK(h, hipStreamWaitEvent(h->aux2, h->ev_ce_fork, 0));
            }
#endif
            GEMMCK(gemm_cleanup(h, s, OP_KC, OP_XC, ghead, h->xov_ctl));
#ifdef FSMG_EXPERIMENTS
            if (ce_tail) {
                const int tiles_n = (h->V1p + 255) / 256;
                HIPCK(h, launch_ce_rows_gated(h->aux2, h->logits, h->V1p, (int)rows, h->V1, h->Y, h->lse, h->ce, dlogits_buf(h),
                                              (float)(1.0 / ((double)rows + 1e-12)), h->xov_done, tiles_n, 256, h->d_err,
                                              h->chain_spin_limit > 0 ? 200000 : 0, h->ce_tail_blocks, h->xov_done + fsmg_model::XOV_DONE - 1));
                HIPCK(h, hipEventRecord(h->ev_ce, h->aux2));
                HIPCK(h, hipStreamWaitEvent(s, h->ev_ce, 0));
            }
#endif
            HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));    // the restricted launch and its tiles in flight
        }
        // ... and one pass in every `xov_selfcheck_every` (1000: ~0.4 ms per 1.6 s of training, 0.03 %) for the handle's whole life: the
        // property the unfenced read rests on belongs to the runtime / firmware, and a stale line on step 40 000 would be as silent
        // as one on step 2.  Re-armed here, on the host, by the count of passes that took this order.
        ++h->xov_passes;
        if (h->xov_selfcheck_every > 0 && h->xov_selfcheck_left <= 0 && h->xov_passes % h->xov_selfcheck_every == 0) h->xov_selfcheck_left = 1;
        if (h->xov_selfcheck_left > 0) GEMMCK(xov_selfcheck(h, B));
        if (h->fs_call) GEMMCK(ce_finish(h, s, B, rows));
        else if (!ce_tail) GEMMCK(ce_rows(h, s, B, 0, T, rows));
    } else if (h->fs_call) {
        {
            ScopedTimer tm(h, "gemm_logits");                                  // the kernel the shape would get anyway (K = Hp: never split)
            ghead.bx3 = h->bx3;
            if (use_h_gemm(h, OP_KC, OP_XC, ghead, mainl)) ghead.bx3 = 3;
            else if (use_ws_gemm(h, OP_KC, OP_XC, ghead, mainl)) { ghead.bx3 = 2; ghead.group_m = 4; }
            HIPCK(h, launch_gemm(s, OP_KC, OP_XC, ghead, 0));
        }
        GEMMCK(ce_finish(h, s, B, rows));
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

}  // namespace fsmg_host
