// Backward pass builder: projection gradients, BPTT chains, weight / input gradients, embedding gradient.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

int dhout_chunk(fsmg_model* h, const Lane& ln, int B, int t0, int t1, OpBatch* defer = nullptr) {
    ScopedTimer tm(h, "gemm_dhout");     // dH = dlogits * W^T for the rows of time steps [t0, t1)
    const int64_t r0 = (int64_t)t0 * B, m = (int64_t)(t1 - t0) * B;
    GemmArgs g{};
    g.A = dlogits_buf(h) + (size_t)r0 * h->V1p; g.lda = h->V1p; g.B = h->P + h->off_w; g.ldb = h->V1p;
    g.C = h->dH + (size_t)r0 * h->Hp; g.ldc = h->Hp; g.M = (int)m; g.N = h->Hp; g.K = h->V1p; g.ksplit = 1;
    if (h->fs_call) g.row_scale = h->crow + r0;          // fused softmax: A holds E', row r of the product times c_r is dH
    return gemm(h, ln, OP_KC, OP_KC, g, defer);
}

GemmArgs dw_args(fsmg_model* h, int B) {      // dW = Hout^T * dlogits, dd = colsum(dlogits)
    GemmArgs g{};
    g.A = h->Hs[h->L - 1] + (size_t)B * h->Hp; g.lda = h->Hp; g.B = dlogits_buf(h); g.ldb = h->V1p;
    g.C = h->G + h->off_w; g.ldc = h->V1p; g.M = h->Hp; g.N = h->V1p; g.K = (int)((int64_t)h->T * B);
    g.colsum = h->G + h->off_d; g.ksplit = 1;
    if (h->fs_call) { g.A = h->Hsc; g.colsum_w = h->crow; }      // fused softmax: (diag(c) Hout)^T E', dd = sum_r c_r E'[r]
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
int backward(fsmg_model* h, int B, int part) {
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
    if (!xcd && h->cs_stale) return fail(h, FSMG_ERR_STATE, "internal: backward pass on the column-split kernels with stale fragment copies of K_h (ensure_cs not called)");
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
    const int rpx = xov ? lstm_xcd16_packed_rows(B, Hp) : 0;
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
    OpBatch late2(h);                       // tail_aside beside a 128-tile dK: the bottom layer's weight-gradient sums, on the main stream behind the join
    OpBatch* const d_now = defer_ok ? &fills : nullptr;
    OpBatch* const d_late = defer_ok ? &late : nullptr;
    // (eager passes only: a captured pass would have to join the auxiliary stream inside the graph; event timing wants one stream)
    const bool aside = h->tail_aside && defer_ok && h->eager_call && h->aux != nullptr && !h->timing && !ov;
    h->side_pending = false;
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
        GEMMCK(gemm_restricted(h, h->aux, OP_XC, OP_XC, gdw, xov_first_free(B, Hp), h->xov_ctl + fsmg_model::XOV_CTL));
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
    // the bandwidth-bound tail of the pass on stream `ts`: the deferred slab sums + the mean loss (one launch), the embedding gradient,
    // the embedding-slice norm.  ts == aux: forked behind what has been issued on the main stream so far, joined by ev_side.
    auto tail_kernels = [&](hipStream_t ts) -> int {
        ScopedTimer tm(h, "embed_grad");
        if (ts != s) {
            HIPCK(h, hipEventRecord(h->ev_side_fork, s));
            HIPCK(h, hipStreamWaitEvent(ts, h->ev_side_fork, 0));
            late.s = ts;
        }
        // tail[1] = the mean loss of the pass: nobody reads it before the step's last kernels, so it rides with the slab sums (one
        // block of a launch that keeps the rest of the chip busy) instead of costing a launch behind the cross entropy
        const bool loss_in_batch = late.r.count > 0;
        if (loss_in_batch) GEMMCK(late.mean(h->ce, rows, h->G + h->n_flat + 1));
        GEMMCK(late.flush());
        // tail[0] = squared norm of the embedding-slice gradients, tail[1] = mean loss of the pass, tail[2] / tail[3] = time-out /
        // token-range indicators: one block's work, which rides in the embedding gradient's first launch where the partials are there already
        const int nb = sqnorm_blocks(rows * h->Ep);
        const SumPartialsArgs sp{h->partials, nb, h->G + h->n_flat + 0, h->d_err, loss_in_batch ? nullptr : h->ce, (int)rows, h->G + h->n_flat + 1};
        const bool sum_rides = dx_sq_done && h->dXpart != nullptr && h->tok_first != nullptr;
        HIPCK(h, launch_embed_grad(ts, h->X, (int)rows, h->dXemb, h->Ep, h->G + h->off_emb, h->tok_first, h->tok_count, h->dXpart, sum_rides ? &sp : nullptr));
        h->tok_table_open = false;
        if (!dx_sq_done) HIPCK(h, launch_sqnorm_partials(ts, h->dXemb, rows * h->Ep, h->partials));
        if (!sum_rides) HIPCK(h, launch_sum_partials(ts, sp.partials, sp.n, sp.dst, sp.flag_src, sp.ce, sp.ce_n, sp.loss_out));
        if (ts != s) {
            HIPCK(h, hipEventRecord(h->ev_side, ts));
            late.s = s;
            h->side_pending = true;
        }
        return FSMG_OK;
    };
    // XCD-partitioned order, stacked layers (hidden 1024, round 6): the weight gradient of layer l + 1 -- its operands are complete once
    // that layer's chain is over -- waits as a work-queue GEMM and runs on the free XCD pair beside layer l's chain (dx of layer l + 1,
    // which that chain reads, goes first); what the pair has not drawn when the chain ends is drained chip-wide.  Same K split, same
    // slab sums in the same order whoever computes an item.
    GemmArgs pend{};
    bool pend_on = false;
    int* const pend_ctl = h->xov_ctl;          // (the forward pair's queue words: free during the backward pass)
    for (int l = h->L - 1; l >= 0; --l) {
        const bool top = l == h->L - 1;
        if (part == 2 && cut_late && l > 0) continue;                       // done in part 1
        const bool skip_chain = part == 2 && cut_late;                      // layer 0: its chain ran in part 1
        if (!skip_chain) {
        if (top && ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_chunk[nch - 1], 0));
        PHASE(4);
        const bool packed = xov && (top || pend_on);      // this layer's chain sits on the first XCDs only
        if (!(top && top_fills_done)) {
            GEMMCK(bptt_fills(h, fills, B, xcd, rs, chain, packed ? rpx : 0));
            GEMMCK(fills.flush());
        }
        for (int c = nch - 1; c >= 0; --c) {
            const int t0 = chunk_begin(h, c, nch), t1 = chunk_begin(h, c + 1, nch);
            if (top && ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_chunk[c], 0));
            ScopedTimer tm(h, "lstm_bwd");
            if (xcd) {
                LstmBwdXcdArgs a{};
                a.rpx = packed ? rpx : 0; a.Hp = Hp; a.bx3 = h->xcd_bx3 ? 1 : 0; a.variant = h->xcd_variant_bwd >= 0 ? h->xcd_variant_bwd : (h->xcd_variant >= 0 ? h->xcd_variant : lstm_xcd_default_variant(B, false, Hp, 0, h->xcd_bx3));
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
                    GEMMCK(late.room(2));
                    GEMMCK(late.reduce(mainl.slabs, mn, dw_split, h->G + h->off_w, mn));
                    GEMMCK(late.reduce(mainl.colsum_slabs, gdw.N, dw_split, h->G + h->off_d, gdw.N));
                    h->slabs_owner = &late;       // (gemm() flushes `late` before anything else writes the main lane's slabs)
                } else {
                    HIPCK(h, launch_reduce_slabs(s, mainl.slabs, mn, dw_split, h->G + h->off_w, mn));
                    HIPCK(h, launch_reduce_slabs(s, mainl.colsum_slabs, gdw.N, dw_split, h->G + h->off_d, gdw.N));
                    HIPCK(h, hipEventRecord(h->ev_bucket[0], s));
                    h->bucket0_recorded = true;
                }
            }
        }
        if (pend_on) {                        // the rest of the waiting weight gradient chip-wide (its slab sums ride in `late`)
            ScopedTimer tm(h, "gemm_dk");
            GEMMCK(gemm_cleanup(h, s, OP_XC, OP_XC, pend, pend_ctl));
            HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));
            pend_on = false;
        }
        }   // !skip_chain
        if (part == 1 && cut_late && l == 0) return FSMG_OK;                // bucket 0 travels beside what follows
        const int in_p = h->in_dim[l];
        PHASE(5);
        // dK_l (weight gradient) and dx_l (input gradient) contract the same dZ and do not depend on each other.  Layer 0 with the tail
        // moved aside: dx first, so that its slab sum, the embedding gradient and the other deferred sums run on the auxiliary stream
        // beside the dK GEMM (below); everywhere else dK first (the layer below waits for dx only).
        OpBatch* dk_defer = d_late;
        // probe != nullptr: nothing is launched, *probe = whether dK_h runs on the 256 x 256-tile kernel
        auto dk_gemm = [&](bool* probe, bool* wait_for_chain = nullptr) -> int {
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
                if (merged && wait_for_chain != nullptr) {       // leave it to the queue beside the next chain
                    GEMMCK(gemm_prepare_queue(h, m, h->xov_dw_split, dk_defer, wait_for_chain));
                    if (*wait_for_chain) { pend = m; return FSMG_OK; }
                }
                if (merged) {
                    if (probe) { *probe = true; return FSMG_OK; }
                    GEMMCK(gemm(h, mainl, OP_XC, OP_XC, m, dk_defer));
                } else {
                GemmArgs g{};                     // dKh = Hprev^T * dZ, db = colsum(dZ)
                g.A = h->Hs[l]; g.lda = Hp; g.B = h->Z[l]; g.ldb = G4;
                g.C = h->G + h->off_kh[l]; g.ldc = G4; g.M = Hp; g.N = G4; g.K = (int)rows;
                g.colsum = h->G + h->off_b[l]; g.ksplit = 1;
                if (probe) { *probe = use_h_gemm(h, OP_XC, OP_XC, g, mainl); return FSMG_OK; }
                GEMMCK(gemm(h, mainl, OP_XC, OP_XC, g, dk_defer));
                GemmArgs k{};                     // dKx = in^T * dZ
                if (l == 0) { k.A = h->P + h->off_emb; k.lda = h->Ep; k.gather = h->X; }
                else { k.A = h->Hs[l - 1] + (size_t)B * Hp; k.lda = Hp; }
                k.B = h->Z[l]; k.ldb = G4; k.C = h->G + h->off_kx[l]; k.ldc = G4;
                k.M = in_p; k.N = G4; k.K = (int)rows; k.ksplit = 1;
                GEMMCK(gemm(h, mainl, OP_XC, OP_XC, k, dk_defer));
                }
            }
            return FSMG_OK;
        };
        auto dx_gemm = [&]() -> int {
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
            return FSMG_OK;
        };
        // Beside a 128-tile dK only (small hidden sizes: the reference's shipped dimensions, +2.7 %).  The 256 x 256-tile kernel holds every
        // CU's whole register file (two 256-VGPR waves per SIMD): a kernel of another stream gets no wave in until its blocks retire
        // -- measured at cfg-B, dx's 12 MB slab sum took 110 us beside dK and ended 13 us after it, the embedding gradient behind it,
        // then the join; with only dW's slab sums aside (beside the 128-tile dx): cfg-B +0.2 %, cfg-C +0.2 %, cfg-D -0.3 %, cfg-E -0.6 %
        // = nothing, so those passes keep the tail in line.
        bool dk_h = false;
        if (aside && l == 0) GEMMCK(dk_gemm(&dk_h));
        if (aside && l == 0 && dk_h) {
            GEMMCK(dk_gemm(nullptr));
            GEMMCK(dx_gemm());
            GEMMCK(tail_kernels(s));
        } else if (aside && l == 0) {
            dk_defer = &late2;
            GEMMCK(dx_gemm());
            GEMMCK(tail_kernels(h->aux));
            GEMMCK(dk_gemm(nullptr));
            HIPCK(h, hipStreamWaitEvent(s, h->ev_side, 0));
            h->side_pending = false;
            GEMMCK(late2.flush());
        } else if (xov && l > 0 && defer_ok && (h->xov_parts & 4) && Hp == 1024) {
            // dK_l as a queue launch on the free XCD pair, STARTED here: the launch must be running before the chain below is -- its blocks
            // are dealt to all eight XCDs in turn, and behind a chain that holds every CU of the first six the dealing stops at the first
            // block meant for one of them (measured: forked right in front of the chain it started 0.6 us behind it and the pair did
            // nothing).  Beside dx_l it shares the pair's CUs with that GEMM's blocks; dx_l goes first on the main stream because the chain
            // below reads dH.
            GEMMCK(dk_gemm(nullptr, &pend_on));
            if (pend_on) {
                GEMMCK(fills.add(pend_ctl, 0u, 4 + gemm_items(pend)));
                GEMMCK(fills.flush());
                HIPCK(h, hipEventRecord(h->ev_fork, s));
                HIPCK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
                GEMMCK(gemm_restricted(h, h->aux, OP_XC, OP_XC, pend, xov_first_free(B, Hp), pend_ctl));
                HIPCK(h, hipEventRecord(h->ev_join, h->aux));
            }
            GEMMCK(dx_gemm());
        } else {
            GEMMCK(dk_gemm(nullptr));
            GEMMCK(dx_gemm());
        }
    }
    if (!aside) GEMMCK(tail_kernels(s));
    if (ov) HIPCK(h, hipStreamWaitEvent(s, h->ev_join, 0));     // dW / dd landed
    PHASE(6);
    h->have_grads = true;
    return FSMG_OK;
}

}  // namespace fsmg_host
