// Debug reads, per-class event timers, the shader-clock probe.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;


// =========================================================================== C ABI
extern "C" {

int fsmg_debug_clock_begin(fsmg_handle h, int32_t microseconds) {
    if (!h || microseconds <= 0 || microseconds > 1000000) return FSMG_ERR_INVALID;
    BEGIN_CALL(h, true);         // (the probe has a stream of its own and touches nothing of the model)
    if (!h->probe) HIPCK(h, hipStreamCreateWithFlags(&h->probe, hipStreamNonBlocking));
    if (!h->d_probe) HIPCK(h, hipMalloc((void**)&h->d_probe, 64));
    HIPCK(h, hipMemsetAsync(h->d_probe, 0, 64, h->probe));
    HIPCK(h, launch_clock_probe(h->probe, (long long)microseconds * 100, h->d_probe));
    return FSMG_OK;
}
int fsmg_debug_clock_end(fsmg_handle h, float* ghz) {
    if (!h || !ghz) return FSMG_ERR_INVALID;
    if (!h->probe || !h->d_probe) return fail(h, FSMG_ERR_STATE, "fsmg_debug_clock_end without fsmg_debug_clock_begin");
    BEGIN_CALL(h, true);
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
    BEGIN_CALL(h);
    const int64_t B = h->lastB, T = h->T, Hp = h->Hp, G4 = h->G4, rows = T * B;
    const float* src = nullptr; int64_t cap = 0;
    auto layer_of = [&](const char* prefix) -> int {
        const size_t n = std::strlen(prefix);
        if (std::strncmp(what, prefix, n) != 0) return -1;
        const int l = std::atoi(what + n);
        return (l >= 0 && l < h->L && std::strlen(what) > n) ? l : -1;
    };
    int l;
    if (!std::strcmp(what, "aux_tries")) { host[0] = (float)h->aux_tries; return FSMG_OK; }       // streams drawn until one ran beside the handle's (-1: none did)
    if (!std::strcmp(what, "fused_softmax")) { host[0] = h->fused_softmax ? 1.0f : 0.0f; if (count > 1) host[1] = h->fs_call ? 1.0f : 0.0f; return FSMG_OK; }   // [0] the knob, [1] whether the last train pass took it
    if (!std::strcmp(what, "xcd_bx3")) { host[0] = h->xcd_bx3 ? 1.0f : 0.0f; return FSMG_OK; }      // a host-side fact: which XCD-local kernel family this handle runs
    if (!std::strcmp(what, "xov_selfcheck")) {        // [0] passes that recomputed and compared the gated projection, [1] passes that took the XCD-partitioned order, [2] the period
        const float v[3] = {(float)h->xov_selfcheck_runs, (float)h->xov_passes, (float)h->xov_selfcheck_every};
        for (int64_t i = 0; i < count && i < 3; ++i) host[i] = v[i];
        return FSMG_OK;
    }
    if (!std::strcmp(what, "xcd_partitioned")) {      // ... and whether its train passes take the XCD-partitioned order: [0] yes / no, [1] XCDs the chains occupy, [2] the last pass
        host[0] = h->xov ? 1.0f : 0.0f;
        if (count > 1) { const int b = h->lastB > 0 ? h->lastB : 45, rpx = lstm_xcd16_packed_rows(b, h->Hp); host[1] = (h->xov && rpx > 0) ? (float)xov_first_free(b, h->Hp) : 8.0f; }
        if (count > 2) host[2] = h->xov_last ? 1.0f : 0.0f;       // [2] whether the LAST train pass took it (its row count decides per call)
        return FSMG_OK;
    }
    if (!std::strcmp(what, "logits")) { src = h->logits; cap = rows * h->V1p; }
    else if (!std::strcmp(what, "dlogits")) { src = dlogits_buf(h); cap = rows * h->V1p; }
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
    BEGIN_CALL(h);
    { const int rc_cs = ensure_cs(h); if (rc_cs != FSMG_OK) return rc_cs; }
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
    BEGIN_CALL(h);
    drain_timers(h);
    h->timing = on != 0;
    return FSMG_OK;
}
int fsmg_timing_select(fsmg_handle h, const char* kernel_class) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    drain_timers(h);
    h->timing_only = kernel_class ? kernel_class : "";
    return FSMG_OK;
}
int fsmg_timing_read(fsmg_handle h, const char* kernel_class, double* total_ms, int64_t* launches) {
    if (!h || !kernel_class) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    drain_timers(h);
    auto it = h->timers.find(kernel_class);
    if (total_ms) *total_ms = it == h->timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == h->timers.end() ? 0 : it->second.launches;
    return FSMG_OK;
}
int fsmg_timing_reset(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    drain_timers(h);
    h->timers.clear();
    return FSMG_OK;
}

}  // extern "C"
