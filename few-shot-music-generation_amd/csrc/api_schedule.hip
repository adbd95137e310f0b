// Which kernel a GEMM takes, which order a pass takes (serial / two-stream / XCD-partitioned), the gated work-queue launches.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

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
        if (g.row_scale != nullptr) HIPCK(h, launch_scale_rows(s, g.C, g.row_scale, g.M, g.N));
        return FSMG_OK;
    }
    // not deferred: this GEMM writes the lane's own slabs -- if a batch still holds a REDUCE over them (the XCD-partitioned order's
    // dW sums wait in `late`), that sum goes out first (an arena that is too small must not cost a gradient)
    if (!deferred && slabs == h->slabs && h->slabs_owner != nullptr) GEMMCK(h->slabs_owner->flush());
    if (!deferred && slabs == h->slabs && h->side_pending) {       // ... or is going out on the auxiliary stream right now (backward(): tail_aside)
        HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_side, 0));
        h->side_pending = false;
    }
    if (deferred && sq != nullptr && g.row_scale != nullptr) return fail(h, FSMG_ERR_STATE, "internal: row-scaled GEMM with squared-norm partials");
    float* C = g.C; float* colsum = g.colsum;
    g.C = slabs; g.c_slab = mn; g.ksplit = S;
    if (colsum) { g.colsum = cslabs; g.colsum_slab = g.N; }
    HIPCK(h, launch_gemm(s, amode, bmode, g, ln.lds_pad));
    if (deferred) {
        if (sq != nullptr && g.row_scale != nullptr) return fail(h, FSMG_ERR_STATE, "internal: row-scaled GEMM with squared-norm partials");
        GEMMCK(defer->room(colsum ? 2 : 1));
        GEMMCK(defer->reduce(slabs, mn, S, C, mn, sq, g.row_scale, g.N));
        if (colsum) GEMMCK(defer->reduce(cslabs, g.N, S, colsum, g.N));
        if (sq_done) *sq_done = sq != nullptr;
        return FSMG_OK;
    }
    HIPCK(h, launch_reduce_slabs2(s, slabs, mn, S, C, mn, cslabs, g.N, colsum, colsum ? g.N : 0));
    if (g.row_scale != nullptr) HIPCK(h, launch_scale_rows(s, C, g.row_scale, g.M, g.N));
    return FSMG_OK;
}

// Two-stream (eager) or single-stream (hipGraph replay) order for a pass over B sequences.  The XCD-local recurrent kernels
// put a high-priority wave on every SIMD of the chip and spend half of their time in hand-offs; GEMM waves beside them
// stretch both (measured at cfg-B: 374-379 episodes/s two-stream with 1-4 chunks against 383-385 single-stream), so a pass
// that takes them runs single-stream; the per-step kernels of big validation batches keep the overlap.
// Hidden 1024: which pair kernels a handle's TRAIN passes run follows their row count -- the bf16-split ones (k_lstm_*_pair16) from three
// row groups per weight copy on, the fp32 row-group chains below (lstm_xcd_bx3_pays) -- not the episode size the handle was created
// for: a MAML-style step (cfg-E) is created for 45 sequences and runs passes of 25 and 20 rows (measured with the format fixed at
// creation: 266 -> 255-261 episodes/s).  A change of format rewrites the weight images (one repack, ~35 us) and drops the graphs;
// evaluation passes take the format they find.
int select_xcd_format(fsmg_model* h, int B) {
    if (h->Hp != 1024 || !h->bx3 || h->xcd_bx3_forced || h->khx == nullptr || !(h->persist && h->xcd)) return FSMG_OK;
    const bool want = lstm_xcd_bx3_pays(B, (int)h->Hp) && B <= h->xcd_max_rows;
    if (want == h->xcd_bx3) return FSMG_OK;
    h->xcd_bx3 = want;
    drop_graphs(h);
    return repack_recurrent_weights(h, h->stream);
}

void choose_schedule(fsmg_model* h, int B, bool train) {
    h->ov_call = h->overlap && (h->overlap_forced || !use_xcd(h, B));
    h->xov_call = false;
    // a pass whose recurrence is one persistent launch per direction is short enough to issue eagerly; per-step kernels (big
    // validation batches, the fallback after a time-out) keep the graph
    h->eager_call = h->eager && !h->ov_call && h->persist && h->persist_fwd && h->persist_bwd &&
                    (use_xcd(h, B) || lstm_fwd_chain_supported(B, h->Hp) || lstm_fwd_chain_rt_supported(B, h->Hp));
    // XCD-partitioned schedule (round 4 form): the bf16-split chains packed on ceil(B / 16) XCDs, the 256-tile work-queue GEMMs of
    // the projection / its weight gradient on the others
    // hidden 1024 (round 6): the TOP layer's pair on three XCD pairs (the bf16-split pair kernels take 16 rows per pair at one MFMA
    // phase's cost), the projection / its weight gradient on the fourth
    if (train && h->xov && ((h->Hp == 512 && h->L == 1) || h->Hp == 1024) && h->xcd_bx3 && h->bx3 && !h->ov_call && h->aux != nullptr && use_xcd(h, B) && h->persist_fwd && h->persist_bwd &&
        (!h->timing || h->timing_only == "lstm_fwd" || h->timing_only == "lstm_bwd")) {
        const int rpx = lstm_xcd16_packed_rows(B, h->Hp);
        h->xov_call = rpx > 0 && xov_first_free(B, h->Hp) <= (h->Hp == 1024 ? 6 : 5);          // at least three XCDs (hidden 1024: a pair) for the GEMMs
        if (h->xov_call) h->eager_call = true;
    }
}
void xov_gate(fsmg_model* h, GemmArgs& g, int B) {     // the projection's A rows arrive time step by time step
    const int rpx = lstm_xcd16_packed_rows(B, h->Hp);
    g.gate = h->xov_prog; g.gate_expect = lstm_xcd_active_blocks(B, rpx, h->Hp); g.gate_rows = B; g.gate_last = h->T - 1;       // (blocks below xcd_first join when the CHAIN is over)
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
// A split-K GEMM for the work queue: slabs in the pass arena, the fixed-order slab sums (same order as gemm()'s) queued in `defer`;
// the launches are the caller's (gemm_restricted beside a chain, gemm_cleanup behind it).  *ok = false: no room, nothing changed.
int gemm_prepare_queue(fsmg_model* h, GemmArgs& g, int split, OpBatch* defer, bool* ok) {
    *ok = false;
    const int S = std::max(1, std::min(std::min(split, MAX_SPLIT), g.K / 256));
    const int64_t mn = (int64_t)g.M * g.N;
    const int64_t need = round_up((int64_t)S * mn, 64) + (g.colsum ? round_up((int64_t)S * g.N, 64) : 0);
    if (S < 2 || defer == nullptr || h->arena == nullptr || g.ldc != g.N || h->arena_off + need > h->arena_cap || g.row_scale != nullptr) return FSMG_OK;
    float* slabs = h->arena + h->arena_off; float* cslabs = slabs + round_up((int64_t)S * mn, 64);
    h->arena_off += need;
    float* C = g.C; float* colsum = g.colsum;
    g.C = slabs; g.c_slab = mn; g.ksplit = S; g.bx3 = 3;
    if (colsum) { g.colsum = cslabs; g.colsum_slab = g.N; }
    if (!xov_fits(g)) { h->arena_off -= need; g.C = C; g.colsum = colsum; g.ksplit = 1; return FSMG_OK; }
    GEMMCK(defer->room(colsum ? 2 : 1));
    GEMMCK(defer->reduce(slabs, mn, S, C, mn));
    if (colsum) GEMMCK(defer->reduce(cslabs, g.N, S, colsum, g.N));
    *ok = true;
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

}  // namespace fsmg_host
