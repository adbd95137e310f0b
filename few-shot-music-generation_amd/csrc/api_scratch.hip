// Activation scratch of a handle (sized for its largest batch so far) and the split-K policy of its GEMMs.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

// ------------------------------------------------------------------ split-K policy
// A 128x128-tile GEMM with few output tiles leaves most of the 256 CUs (2 resident blocks each)
// idle; splitting K multiplies the block count.  Cost model: MFMA time at ~100 TF/s divided by the
// slot efficiency of tiles*S blocks over the resident-block slots, plus S slabs of C written and read back.
int pick_split(int64_t M, int64_t N, int64_t K, int64_t slots, bool bx3, int tile_mn) {
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
    const int64_t n_hx = xrows ? std::max(lstm_xcd_hx_floats(xrows, (int)T, (int)Hp, h->xcd_bx3), Hp == 1024 ? lstm_xcd_hx_floats(xrows, (int)T, (int)Hp, !h->xcd_bx3) : 0LL) : 0;
    const int64_t n_inx = (xrows && (Hp != 1024 || h->pair_mode >= 2)) ? lstm_xcd_inbox_floats(xrows, (int)Hp) : 0;
    const int64_t o_hx = place(xrows ? 4 * n_hx : 256), o_inx = place(n_inx ? 4 * n_inx : 256);
    const int64_t o_dc = place(4 * (int64_t)B * Hp), o_dh = place(4 * rows * Hp);
    const int64_t o_lg = place(4 * rows * h->V1p), o_dlg = place(4 * rows * h->V1p), o_lse = place(4 * rows), o_ce = place(4 * rows);
    const int64_t o_dx = place(4 * rows * h->Ep);
    const int64_t o_dxp = place(4 * rows * h->Ep);      // per-chunk partial rows of the heavy tokens' embedding gradient (launch_embed_grad)
    const int64_t o_hsc = place(4 * rows * Hp), o_crow = place(4 * (rows + 32));       // fused softmax: c_r * h_r, c_r
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
    // grow-only: pick_split may choose MORE slabs for a small batch than for the one the scratch is growing for (fewer tiles to fill
    // the chip with), so a handle that has grown for a validation batch keeps what its train shapes were given
    slab_need = std::max(slab_need, h->slab_cap);
    arena_need = std::max(arena_need, h->arena_cap);
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
    h->lse = (float*)(s + o_lse); h->ce = (float*)(s + o_ce); h->dXemb = (float*)(s + o_dx); h->dXpart = (float*)(s + o_dxp);
    h->Hsc = (float*)(s + o_hsc); h->crow = (float*)(s + o_crow);
    HIPCK(h, hipMemsetAsync(h->crow, 0, 4 * (size_t)(rows + 32), h->stream));          // (the weights past the last row are read by a partial k tile, times zero)
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

}  // namespace fsmg_host
