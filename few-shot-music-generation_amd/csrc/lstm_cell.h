// Device helpers shared by the recurrent kernels (lstm_step.hip, lstm_xcd.hip): the cell arithmetic of
// BasicLSTMCell(forget_bias=1), gate order i, j, f, o (reference src/models/lstm_baseline.py:44-55)
// -- ONE definition, floating-point contraction off, so that every kernel variant produces the same bits for the same
// inputs -- and the write-through / L1-bypassing memory operations of the in-launch hand-offs.
#pragma once
#include <hip/hip_runtime.h>

namespace fsmg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate nonlinearities on v_exp_f32 (exp2 of a pre-scaled argument, ~1 ulp) instead of the libm call chains:
// absolute error <= ~1.5e-7 on outputs in [-1, 1], far inside the 1e-4 NLL bound; the small-|x| branch of
// tanh is a Taylor polynomial so tanh(x) ~ x keeps full relative accuracy where 1 - 2/(1+e^2x) cancels.
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float x2 = x * x;
    const float poly = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.05396825f + x2 * 0.02186949f))));
    const float big = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x));
    return fabsf(x) < 0.25f ? poly : big;
}

// The cell update of one (row, unit): one definition with floating-point contraction OFF, shared by every forward
// kernel (per step, patch, persistent, all-row-tiles persistent) so that they produce the same bits.
struct CellOut { float si, tj, sf, so, c, h; };
__device__ __forceinline__ CellOut cell_forward(const float (&zg)[4], float c_prev) {
#pragma clang fp contract(off)
    CellOut o;
    o.si = sigmoidf_(zg[0]);
    o.tj = tanhf_(zg[1]);
    o.sf = sigmoidf_(zg[2] + 1.0f);          // forget_bias = 1 added at run time
    o.so = sigmoidf_(zg[3]);
    o.c = c_prev * o.sf + o.si * o.tj;
    o.h = tanhf_(o.c) * o.so;
    return o;
}

// Gate gradients of one (row, unit) of one time step.  One definition with floating-point contraction OFF, shared by
// the per-step and the persistent kernels, so both produce the same bits whatever fusions the surrounding code invites.
struct CellGrad { float di, dj, df, dg, dc_out; };
__device__ __forceinline__ CellGrad cell_backward(float si, float tj, float sf, float so, float ct, float cp,
                                                  float dc_in, float dh) {
#pragma clang fp contract(off)
    CellGrad g;
    const float tc = tanhf_(ct);
    const float dc = dc_in + dh * so * (1.0f - tc * tc);
    g.di = dc * tj * si * (1.0f - si);
    g.dj = dc * si * (1.0f - tj * tj);
    g.df = dc * cp * sf * (1.0f - sf);
    g.dg = dh * tc * so * (1.0f - so);
    g.dc_out = dc * sf;
    return g;
}

__device__ __forceinline__ f32x4 load_sc1(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// The trailing s_nop covers the VMEM-store-data hazard (a VALU write to the data VGPRs of a store wider than 8 bytes
// needs a wait state after it): hipcc inserts that for its own stores, not behind inline asm -- and it did reuse the
// first two data registers as the next store's address (k_lstm_bwd_rs<8>: garbage in one tile per wave).
__device__ __forceinline__ void store_sc1(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// all 4 components of a fragment written (the buffer is pre-filled with 0xFFFFFFFF words; no h value has that pattern)
__device__ __forceinline__ bool frag_ready(const f32x4& v) {
    return __float_as_uint(v[0]) != 0xFFFFFFFFu && __float_as_uint(v[1]) != 0xFFFFFFFFu &&
           __float_as_uint(v[2]) != 0xFFFFFFFFu && __float_as_uint(v[3]) != 0xFFFFFFFFu;
}

// s_sleep argument between two polls of a hand-off word (units of 64 clocks)
#ifndef FSMG_POLL_SLEEP
#define FSMG_POLL_SLEEP 1
#endif

// Waits until the N hand-off fragments at af, af + 64, ... (one 16-byte word per lane each) have been written: polls the
// first one until none of its four components shows the fill pattern, then fetches the rest and re-fetches them until
// none does.  Returns false when the spin limit is hit or another wave has already raised the time-out flag.
template <int N>
__device__ __forceinline__ bool wait_fragments(const f32x4* af, f32x4 (&av)[N], int spin_limit, int* err_flag) {
    for (int spins = 0;; ++spins) {
        av[0] = load_sc1(af);
        drain_vmem();
        asm volatile("" : "+v"(av[0]));
        if (__all(frag_ready(av[0]))) break;
        __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
        if (spins >= spin_limit || ((spins & 255) == 255 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) return false;
    }
    if (N > 1) {
        for (int spins = 0;; ++spins) {
#pragma unroll
            for (int j = 1; j < N; ++j) av[j] = load_sc1(af + j * 64);
            drain_vmem();
            bool ok = true;
#pragma unroll
            for (int j = 1; j < N; ++j) { asm volatile("" : "+v"(av[j])); ok &= frag_ready(av[j]); }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
            if (spins >= spin_limit || ((spins & 255) == 255 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) return false;
        }
    }
    return true;
}

}  // namespace fsmg
