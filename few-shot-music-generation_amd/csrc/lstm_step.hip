// Fused LSTM cell steps for gfx950: recurrent contraction on v_mfma_f32_16x16x4_f32 with the
// gate nonlinearities / Hadamard products (forward) and the gate-gradient arithmetic
// (backward) fused behind it.  One launch per time step; the launches of a sequence are
// replayed as a hipGraph by the caller.  Cell semantics: BasicLSTMCell(forget_bias=1),
// gate order i, j, f, o (reference src/models/lstm_baseline.py:44-55).
//
// Gate columns are "packed": column pc = 16*(u/4) + 4*gate + (u%4) holds gate `gate` of
// hidden unit u, so the 16 columns of one MFMA N-tile are the four gates of four units and
// the cell update for those units is local to the block.
//
// K-order trick: an MFMA 16x16x4 step consumes k = 4 values, lane slot q = lane>>4 supplying
// A[m][k_q] and B[k_q][n].  Any bijection slot->k is valid as long as A and B agree, so within
// a group of 16 k's slot q takes k = 16g + 4q + s at sub-step s: every lane then reads its
// four k's as ONE 16-byte load from the row-major activations instead of four strided dwords.
//
// Fragment-ordered operands: a row-major operand makes every wave-load touch 16 rows x 64 B (half cache
// lines; measured 14-24 B/cycle/CU).  Both MFMA operands are therefore ALSO kept in "fragment order":
// [tile][k-group g][lane][4 floats] with lane = 16*q + (row or column), so the load for one k-group is a
// single contiguous 1 KiB per wave.  The weights are repacked once per Adam step (k_repack_kh), the
// activations (h_t forward, dz_t backward) are written in fragment order by the epilogue of the step
// that produces them, next to the row-major copy the big GEMMs consume.
#include "fsmg_kernels.h"
#include "lstm_cell.h"
#include "lstm_repack.h"

namespace fsmg {

namespace {

#ifndef FSMG_FWD_NW
#define FSMG_FWD_NW 4
#endif
#ifndef FSMG_BWD_NW
#define FSMG_BWD_NW 8
#endif
// wave priority of the step kernels.  Measured beside the cfg-B dW GEMM at 2 blocks/CU (tools/step_contention.cpp):
// forward 18.7 us/step at priority 0, 10.9 at 3 (4.5 alone); backward 24.4 / 13.3 (6.2 alone)
#ifndef FSMG_STEP_PRIO_LEVEL
#define FSMG_STEP_PRIO_LEVEL 3
#endif
#define FSMG_STEP_PRIO __builtin_amdgcn_s_setprio(FSMG_STEP_PRIO_LEVEL)
constexpr int FWD_NW = FSMG_FWD_NW;   // waves per forward-step block (split K = Hp)
constexpr int BWD_NW = FSMG_BWD_NW;   // waves per backward-step block (split K = 4Hp)

// ---------------------------------------------------------------- forward
// grid (4Hp/16, ceil(B/16)); 256 threads = 4 waves, wave w owns a quarter of the K = Hp range.
// Every operand of a wave's K range is requested before the first MFMA (the loop over groups is
// unrolled in chunks of FWD_CHUNK with all loads hoisted), so a step pays ONE memory latency, not
// one per k-group; the cell inputs of the epilogue (x-part pre-activations, c_{t-1}) are
// prefetched by wave 0 at kernel entry as well.
// N k-groups (16 k's each): all 2N 16-byte loads are issued before the first of the 4N MFMAs; no predicates.
template <int N>
__device__ __forceinline__ void fwd_chunk(const float4* __restrict__ af, const float4* __restrict__ bf, int g0,
                                          f32x4& acc0, f32x4& acc1) {
    float4 av[N], bv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        av[j] = af[(g0 + j) * 64];
        bv[j] = bf[(g0 + j) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);      // keep every load ahead of the first MFMA (one latency per chunk)
#pragma unroll
    for (int j = 0; j < N; ++j) {
        f32x4& acc = (j & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
    }
}

#define FSMG_STAMP(i)                                                                                  \
    if (PROF && lane == 0)                                                                              \
        prof[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime()

template <bool PROF, int NW>
__global__ __launch_bounds__(64 * NW, NW) void k_lstm_fwd_step(const LstmFwdArgs a, unsigned long long* prof) {
    __shared__ float red[NW][16][17];
    FSMG_STEP_PRIO;       // latency-critical chain: win issue arbitration against co-resident GEMM waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int nb = blockIdx.x;           // unit block: units 4nb..4nb+3, packed cols 16nb..16nb+15
    const int m0 = blockIdx.y * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp;
    FSMG_STAMP(0);

    // epilogue operands (wave 0: thread -> (row, unit))
    const int erow = tid >> 2, euu = tid & 3;
    const int eb = m0 + erow;
    const bool eact = (tid < 64) && (eb < a.B);
    float zin[4] = {0.f, 0.f, 0.f, 0.f};
    float cp = 0.f;
    float* zp = a.z + (long long)eb * G4 + 16 * nb + euu;
    const int eu = 4 * nb + euu;
    if (eact) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) zin[gi] = zp[4 * gi];
        cp = a.c_prev[(long long)eb * Hp + eu];
    }

    const int ngroups = Hp >> 4;
    const int g_beg = (wave * ngroups) / NW, g_end = ((wave + 1) * ngroups) / NW;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment-ordered operands: [tile][group][lane] float4 (pad rows of the last M tile feed discarded outputs)
    const float4* af = reinterpret_cast<const float4*>(a.hF_prev) + ((size_t)blockIdx.y * ngroups) * 64 + lane;
    const float4* bf = reinterpret_cast<const float4*>(a.KhF) + ((size_t)nb * ngroups) * 64 + lane;

    int g = g_beg;
    while (g + 8 <= g_end) { fwd_chunk<8>(af, bf, g, acc0, acc1); g += 8; }
    if (g + 4 <= g_end) { fwd_chunk<4>(af, bf, g, acc0, acc1); g += 4; }
    if (g + 2 <= g_end) { fwd_chunk<2>(af, bf, g, acc0, acc1); g += 2; }
    if (g < g_end) fwd_chunk<1>(af, bf, g, acc0, acc1);
    if (PROF) { __builtin_amdgcn_s_waitcnt(0); FSMG_STAMP(1); }        // all loads landed (profiling build only)
    // C/D layout 16x16: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * q + r][l15] = acc0[r] + acc1[r];
    FSMG_STAMP(2);
    __syncthreads();
    FSMG_STAMP(3);

    if (eact) {
        float zg[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int c = 4 * gi + euu;
            float zs = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) zs += red[w][erow][c];
            zg[gi] = zin[gi] + zs;
        }
        const CellOut co = cell_forward(zg, cp);
        const float si = co.si, tj = co.tj, sf = co.sf, so = co.so, cn = co.c, hn = co.h;
        a.c_next[(long long)eb * Hp + eu] = cn;
        a.h_next[(long long)eb * Hp + eu] = hn;
        // fragment-ordered copy for the next step's A operand: group eu/16, slot q = (eu%16)/4, sub-step eu%4
        a.hF_next[(((size_t)blockIdx.y * ngroups + (eu >> 4)) * 64 + 4 * (eu & 12) + erow) * 4 + (eu & 3)] = hn;
        zp[0] = si; zp[4] = tj; zp[8] = sf; zp[12] = so;  // activated gates kept for BPTT
    }
    FSMG_STAMP(4);
}


// ---------------------------------------------------------------- forward, persistent over a range of time steps
// Same decomposition as k_lstm_fwd_step (block = 16 packed gate columns = 4 hidden units x one 16-row tile, 4 waves
// split K = Hp), but the launch loops over t: the wave's slice of Kh is loaded ONCE into GPW float4 registers, c
// stays in a register, and the only per-step traffic is the h_t fragment (GPW KiB per wave) plus the cell's own
// inputs and outputs.  Hand-off of h_t between the 4Hp/16 blocks of a row tile (MI355X_MICROARCH.md "inter-workgroup
// visibility"): the data is its own flag.  The fragment buffer of every time index is pre-filled with 0xFFFFFFFF
// words (no h value has that bit pattern, not even a NaN produced by arithmetic); the producer writes its 256
// contiguous bytes with 16-byte sc1 (write-through) stores and nothing else; each consumer wave polls ONE of its
// fragments with sc1 loads until no component shows the fill pattern, then fetches the others and re-fetches until
// none does (4-byte words are never torn).  A first version with a per-(row tile, step) arrival counter (drain +
// atomic + one-lane poll + barrier + sc1 loads) took 5.3 us per step against 4.75 for one launch per step.
// Every spin is bounded: on a timeout (some block not resident) err_flag becomes 2 and all blocks leave.
template <int GPW>
__global__ __launch_bounds__(256, 2) void k_lstm_fwd_chain(const LstmFwdChainArgs a) {
    // two copies, alternating by time step: three of the four waves poll fragments that come from OTHER blocks only, so
    // they can be a whole step ahead of wave 0 (never two: the barrier below) and would otherwise overwrite the partial
    // sums wave 0 is still adding up -- seen as a 1e-3 relative glitch in one block's h about once in 300 passes, and
    // only beside GEMMs that delay wave 0
    __shared__ float red[2][4][16][17];
    __shared__ int s_fail;
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int nb = blockIdx.x, rt = blockIdx.y;
    const int m0 = rt * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp, B = a.B;
    const int ngroups = Hp >> 4;
    const size_t hf_step = (size_t)gridDim.y * 16 * Hp;          // floats per time index of HF
    if (tid == 0) s_fail = 0;

    // this wave's slice of the recurrent weights: resident for the whole launch
    f32x4 bw[GPW];
    {
        const f32x4* bf = reinterpret_cast<const f32x4*>(a.KhF) + ((size_t)nb * ngroups + wave * GPW) * 64 + lane;
#pragma unroll
        for (int j = 0; j < GPW; ++j) bw[j] = bf[j * 64];
    }
    // epilogue mapping (wave 0): thread -> (row, unit)
    const int erow = tid >> 2, euu = tid & 3;
    const int eb = m0 + erow, eu = 4 * nb + euu;
    const bool eact = (tid < 64) && (eb < B);
    float cp = eact ? a.Cs[((size_t)a.t0 * B + eb) * Hp + eu] : 0.0f;
    __syncthreads();

    for (int t = a.t0; t < a.t1; ++t) {
        // x-part pre-activations of this step do not depend on the recurrence: requested before the wait
        float zin[4] = {0.f, 0.f, 0.f, 0.f};
        float* zp = a.Z + ((size_t)t * B + eb) * G4 + 16 * nb + euu;
        if (eact) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) zin[gi] = zp[4 * gi];
        }
        // h_t fragments of this row tile, this wave's K range.  The data is its own flag: poll ONE fragment until
        // its producer has written it, then fetch the rest and re-fetch until none shows the fill pattern.
        f32x4 av[GPW];
        {
            const f32x4* af = reinterpret_cast<const f32x4*>(a.HF + (size_t)t * hf_step) + ((size_t)rt * ngroups + wave * GPW) * 64 + lane;
            const bool fail = !wait_fragments<GPW>(af, av, a.spin_limit, a.err_flag);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GPW; ++j) {
            f32x4& acc = (j & 1) ? acc1 : acc0;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][0], bw[j][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][1], bw[j][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][2], bw[j][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][3], bw[j][3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[t & 1][wave][4 * q + r][l15] = acc0[r] + acc1[r];
        __syncthreads();
        if (s_fail) return;                                          // block-uniform: a wave of this block timed out

        if (tid < 64) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (eact) {
                float zg[4];
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int c = 4 * gi + euu;
                    float zs = 0.0f;                                  // same summation order as k_lstm_fwd_step
#pragma unroll
                    for (int w = 0; w < 4; ++w) zs += red[t & 1][w][erow][c];
                    zg[gi] = zin[gi] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h;
                cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            // fragment-ordered h_{t+1}: the 4 units of a row are one float4 (group eu/16, lane slot 4*(eu&12) + row);
            // pad rows publish zeros, so every word of the buffer is written and the readers' test terminates
            f32x4 hv;
            hv[0] = __shfl(hn, (lane & ~3) + 0); hv[1] = __shfl(hn, (lane & ~3) + 1);
            hv[2] = __shfl(hn, (lane & ~3) + 2); hv[3] = __shfl(hn, (lane & ~3) + 3);
            if (euu == 0) {
                const int u0 = 4 * nb;
                f32x4* dst = reinterpret_cast<f32x4*>(a.HF + (size_t)(t + 1) * hf_step) +
                             ((size_t)rt * ngroups + (u0 >> 4)) * 64 + 4 * (u0 & 12) + erow;
                store_sc1(dst, hv);
            }
            if (eact) {      // the outputs nobody waits for go out behind the hand-off
                a.Cs[((size_t)(t + 1) * B + eb) * Hp + eu] = cp;
                a.Hs[((size_t)(t + 1) * B + eb) * Hp + eu] = hn;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;  // activated gates kept for BPTT
            }
        }
        // the other waves run ahead into step t+1 (into the other copy of `red`)
    }
}

// ---------------------------------------------------------------- forward, persistent, all row tiles in one block
// grid (4Hp/16): block = one column tile (4 hidden units) x ALL RT = ceil(B/16) row tiles, for the shapes whose
// (column tile, row tile) grid cannot be resident at once (Hp = 1024: 768 blocks at B = 45; 100-row episodes at
// Hp = 512: 896).  Same weights-in-registers / data-as-flag protocol as k_lstm_fwd_chain; inside a step every wave
// walks the row tiles with its GPW resident weight fragments, then wave w finishes the cell for row tiles w, w+4
// (c stays in registers there).  The partial sums live in dynamic LDS, two copies alternating by time step.
template <int GPW>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_chain_rt(const LstmFwdChainArgs a) {     // GPW = 16 needs > 256 registers
    extern __shared__ float red_rt[];                         // [2][RT][4 waves][16][17]
    __shared__ int s_fail;
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int nb = blockIdx.x;
    const int Hp = a.Hp, G4 = 4 * a.Hp, B = a.B;
    const int RT = (B + 15) / 16;
    const int ngroups = Hp >> 4;
    const size_t hf_step = (size_t)RT * 16 * Hp;
    if (tid == 0) s_fail = 0;
    auto red = [&](int buf, int rt, int w, int row, int col) -> float& { return red_rt[((((size_t)buf * RT + rt) * 4 + w) * 16 + row) * 17 + col]; };

    f32x4 bw[GPW];
    {
        const f32x4* bf = reinterpret_cast<const f32x4*>(a.KhF) + ((size_t)nb * ngroups + wave * GPW) * 64 + lane;
#pragma unroll
        for (int j = 0; j < GPW; ++j) bw[j] = bf[j * 64];
    }
    // epilogue: wave w owns row tiles w and w + 4; lane -> (row, unit)
    const int erow = lane >> 2, euu = lane & 3, eu = 4 * nb + euu;
    int eb[2]; bool eact[2]; float cp[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int rt = wave + 4 * e;
        eb[e] = rt * 16 + erow;
        eact[e] = rt < RT && eb[e] < B;
        cp[e] = eact[e] ? a.Cs[((size_t)a.t0 * B + eb[e]) * Hp + eu] : 0.0f;
    }
    __syncthreads();

    for (int t = a.t0; t < a.t1; ++t) {
        float zin[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
                zin[e][gi] = eact[e] ? a.Z[((size_t)t * B + eb[e]) * G4 + 16 * nb + euu + 4 * gi] : 0.0f;
        // row tile 0 pays the hop (poll); the fragments of row tile rt + 1 are requested before the MFMAs of row tile rt
        // and checked after them -- by then their producers have long published, so the check almost never fails
        const f32x4* af0 = reinterpret_cast<const f32x4*>(a.HF + (size_t)t * hf_step) + ((size_t)wave * GPW) * 64 + lane;
        f32x4 av[GPW], nx[GPW];
        bool fail = !wait_fragments<GPW>(af0, av, a.spin_limit, a.err_flag);
        for (int rt = 0; rt < RT; ++rt) {
            const bool more = rt + 1 < RT && !fail;
            const f32x4* afn = af0 + (size_t)(rt + 1) * ngroups * 64;
            if (more) {
#pragma unroll
                for (int j = 0; j < GPW; ++j) nx[j] = load_sc1(afn + j * 64);
            }
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                f32x4& acc = (j & 1) ? acc1 : acc0;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][0], bw[j][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][1], bw[j][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][2], bw[j][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][3], bw[j][3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red(t & 1, rt, wave, 4 * q + r, l15) = acc0[r] + acc1[r];
            if (more) {
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int j = 0; j < GPW; ++j) { asm volatile("" : "+v"(nx[j])); ok &= frag_ready(nx[j]); }
                if (!__all(ok)) fail = !wait_fragments<GPW>(afn, nx, a.spin_limit, a.err_flag);
#pragma unroll
                for (int j = 0; j < GPW; ++j) av[j] = nx[j];
            }
            if (fail) break;
        }
        if (fail && lane == 0) {
            __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_fail = 1;
        }
        __syncthreads();
        if (s_fail) return;

#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rt = wave + 4 * e;
            if (rt >= RT) break;                                      // wave-uniform
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (eact[e]) {
                float zg[4];
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int c = 4 * gi + euu;
                    float zs = 0.0f;                                  // same summation order as k_lstm_fwd_step
#pragma unroll
                    for (int w = 0; w < 4; ++w) zs += red(t & 1, rt, w, erow, c);
                    zg[gi] = zin[e][gi] + zs;
                }
                const CellOut co = cell_forward(zg, cp[e]);
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
                hn = co.h;
                cp[e] = co.c;
            }
            f32x4 hv;
            hv[0] = __shfl(hn, (lane & ~3) + 0); hv[1] = __shfl(hn, (lane & ~3) + 1);
            hv[2] = __shfl(hn, (lane & ~3) + 2); hv[3] = __shfl(hn, (lane & ~3) + 3);
            if (euu == 0) {
                const int u0 = 4 * nb;
                f32x4* dst = reinterpret_cast<f32x4*>(a.HF + (size_t)(t + 1) * hf_step) +
                             ((size_t)rt * ngroups + (u0 >> 4)) * 64 + 4 * (u0 & 12) + erow;
                store_sc1(dst, hv);
            }
            if (eact[e]) {
                a.Cs[((size_t)(t + 1) * B + eb[e]) * Hp + eu] = cp[e];
                a.Hs[((size_t)(t + 1) * B + eb[e]) * Hp + eu] = hn;
                float* zp = a.Z + ((size_t)t * B + eb[e]) * G4 + 16 * nb + euu;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;
            }
        }
    }
}

// ---------------------------------------------------------------- forward, many rows (validation batches, wide episodes)
// With R row tiles every (column group, row tile) block of the kernel above re-fetches fragments that its
// neighbours fetch too: R x the weights, 4Hp/16 x the activations (164 MB of L2->CU traffic per step at 320
// rows).  Here a block owns a CT x RT patch of output tiles and reuses each fragment CT (activations) or RT
// (weights) times from registers.  grid (4Hp/16/CT, ceil(tiles/RT)); same K split over the waves, same epilogue.
template <int N, int CT, int RT>
__device__ __forceinline__ void fwd_patch_chunk(const float4* const (&af)[RT], const float4* const (&bf)[CT], int g0,
                                                f32x4 (&acc)[RT][CT]) {
    float4 av[RT][N], bv[CT][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int ri = 0; ri < RT; ++ri) av[ri][j] = af[ri][(g0 + j) * 64];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) bv[ci][j] = bf[ci][(g0 + j) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int ri = 0; ri < RT; ++ri)
#pragma unroll
            for (int ci = 0; ci < CT; ++ci) {
                acc[ri][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ri][j].x, bv[ci][j].x, acc[ri][ci], 0, 0, 0);
                acc[ri][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ri][j].y, bv[ci][j].y, acc[ri][ci], 0, 0, 0);
                acc[ri][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ri][j].z, bv[ci][j].z, acc[ri][ci], 0, 0, 0);
                acc[ri][ci] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ri][j].w, bv[ci][j].w, acc[ri][ci], 0, 0, 0);
            }
    }
}

template <int NW, int CT, int RT>
__global__ __launch_bounds__(64 * NW, NW) void k_lstm_fwd_patch(const LstmFwdArgs a) {
    constexpr int PAIRS = RT * 16 * CT * 4;                 // (row, unit) pairs of the patch, one epilogue thread each
    static_assert(PAIRS <= 64 * NW, "one epilogue thread per (row, unit)");
    __shared__ float red[NW][RT * 16][CT * 16 + 1];
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int Hp = a.Hp, G4 = 4 * a.Hp, ngroups = Hp >> 4;
    const int ntiles = (a.B + 15) >> 4;
    const int nb0 = blockIdx.x * CT, mt0 = blockIdx.y * RT;

    const int prow = tid / (CT * 4), pu = tid % (CT * 4);
    const int eb = mt0 * 16 + prow, enb = nb0 + (pu >> 2), euu = pu & 3, eu = 4 * enb + euu;
    const bool eact = (tid < PAIRS) && (eb < a.B);
    float zin[4] = {0.f, 0.f, 0.f, 0.f};
    float cp = 0.f;
    float* zp = a.z + (long long)eb * G4 + 16 * enb + euu;
    if (eact) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) zin[gi] = zp[4 * gi];
        cp = a.c_prev[(long long)eb * Hp + eu];
    }

    const int g_beg = (wave * ngroups) / NW, g_end = ((wave + 1) * ngroups) / NW;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) acc[ri][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* af[RT];
    const float4* bf[CT];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)           // a row tile past the batch is clamped: its outputs are never stored
        af[ri] = reinterpret_cast<const float4*>(a.hF_prev) + ((size_t)min(mt0 + ri, ntiles - 1) * ngroups) * 64 + lane;
#pragma unroll
    for (int ci = 0; ci < CT; ++ci)
        bf[ci] = reinterpret_cast<const float4*>(a.KhF) + ((size_t)(nb0 + ci) * ngroups) * 64 + lane;

    int g = g_beg;
    while (g + 4 <= g_end) { fwd_patch_chunk<4, CT, RT>(af, bf, g, acc); g += 4; }
    if (g + 2 <= g_end) { fwd_patch_chunk<2, CT, RT>(af, bf, g, acc); g += 2; }
    if (g < g_end) fwd_patch_chunk<1, CT, RT>(af, bf, g, acc);
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
        for (int ci = 0; ci < CT; ++ci)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][ri * 16 + 4 * q + r][ci * 16 + l15] = acc[ri][ci][r];
    __syncthreads();

    if (eact) {
        float zg[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int c = (pu >> 2) * 16 + 4 * gi + euu;
            float zs = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) zs += red[w][prow][c];
            zg[gi] = zin[gi] + zs;
        }
        const CellOut co = cell_forward(zg, cp);
        const float si = co.si, tj = co.tj, sf = co.sf, so = co.so, cn = co.c, hn = co.h;
        a.c_next[(long long)eb * Hp + eu] = cn;
        a.h_next[(long long)eb * Hp + eu] = hn;
        a.hF_next[(((size_t)(eb >> 4) * ngroups + (eu >> 4)) * 64 + 4 * (eu & 12) + (eb & 15)) * 4 + (eu & 3)] = hn;
        zp[0] = si; zp[4] = tj; zp[8] = sf; zp[12] = so;
    }
}

// ---------------------------------------------------------------- backward
// grid (Hp/16, ceil(B/16)); 512 threads = 8 waves splitting K = 4Hp (packed gate columns).
// dh_rec[b][u] = sum_pc dz_{t+1}[b][pc] * Kh[u][pc]; then the gate gradients of step t for the
// block's 16 rows x 16 units.  Same load-everything-first structure as the forward step.
template <int N>
__device__ __forceinline__ void bwd_chunk(const float4* __restrict__ af, const float4* __restrict__ bf, int g0,
                                          f32x4& acc0, f32x4& acc1) {
    float4 av[N], bv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        av[j] = af[(g0 + j) * 64];
        bv[j] = bf[(g0 + j) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        f32x4& acc = (j & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
    }
}

// Producer-major accumulation: every quadruple of k-groups (= the 64 packed gate columns of 16 hidden units, what one
// block of the reduce-scatter kernel k_lstm_bwd_rs contributes) gets its own accumulator chain, and the chains are
// added in order -- the summation order of that kernel, so the two agree to the bit.  N groups, N % 4 == 0.
template <int N>
__device__ __forceinline__ void bwd_chunk_quads(const float4* __restrict__ af, const float4* __restrict__ bf, int g0,
                                                f32x4& wsum, bool& first) {
    float4 av[N], bv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        av[j] = af[(g0 + j) * 64];
        bv[j] = bf[(g0 + j) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[N / 4];
#pragma unroll
    for (int p = 0; p < N / 4; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)                 // interleave the N/4 independent chains
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < N / 4; ++p) {
                const float4 x = av[4 * p + jj], y = bv[4 * p + jj];
                const float xa = e == 0 ? x.x : e == 1 ? x.y : e == 2 ? x.z : x.w;
                const float ya = e == 0 ? y.x : e == 1 ? y.y : e == 2 ? y.z : y.w;
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, ya, acc[p], 0, 0, 0);
            }
#pragma unroll
    for (int p = 0; p < N / 4; ++p) {
        wsum = first ? acc[p] : wsum + acc[p];
        first = false;
    }
}

template <bool PROF, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void k_lstm_bwd_step(const LstmBwdArgs a, unsigned long long* prof) {
    __shared__ float red[NW][16][17];
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int u0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp;
    FSMG_STAMP(0);

    // epilogue operands (waves 0-3: thread -> (row, unit)), requested before the contraction
    const int erow = tid >> 4, eun = tid & 15;
    const int eb = m0 + erow, eu = u0 + eun;
    const bool eact = (tid < 256) && (eb < a.B);
    const long long hi = (long long)eb * Hp + eu;
    float* gp = a.gates + (long long)eb * G4 + 16 * (eu >> 2) + (eu & 3);
    float si = 0.f, tj = 0.f, sf = 0.f, so = 0.f, ct = 0.f, cp = 0.f, dcv = 0.f, dht = 0.f;
    if (eact) {
        si = gp[0]; tj = gp[4]; sf = gp[8]; so = gp[12];
        ct = a.c_t[hi]; cp = a.c_prev[hi]; dcv = a.dc[hi]; dht = a.dh_top[hi];
    }

    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.dzF_next != nullptr) {
        const int ngroups = G4 >> 4;
        const int g_beg = (wave * ngroups) / NW, g_end = ((wave + 1) * ngroups) / NW;
        const float4* af = reinterpret_cast<const float4*>(a.dzF_next) + ((size_t)blockIdx.y * ngroups) * 64 + lane;
        const float4* bf = reinterpret_cast<const float4*>(a.KhF) + ((size_t)blockIdx.x * ngroups) * 64 + lane;
        int g = g_beg;
        if (ngroups % (4 * NW) == 0) {       // whole quadruples per wave (Hp % 128 == 0): producer-major order
            bool first = true;
            while (g + 8 <= g_end) { bwd_chunk_quads<8>(af, bf, g, acc0, first); g += 8; }
            if (g + 4 <= g_end) { bwd_chunk_quads<4>(af, bf, g, acc0, first); g += 4; }
        } else {
            while (g + 8 <= g_end) { bwd_chunk<8>(af, bf, g, acc0, acc1); g += 8; }
            if (g + 4 <= g_end) { bwd_chunk<4>(af, bf, g, acc0, acc1); g += 4; }
            if (g + 2 <= g_end) { bwd_chunk<2>(af, bf, g, acc0, acc1); g += 2; }
            if (g < g_end) bwd_chunk<1>(af, bf, g, acc0, acc1);
        }
    }
    if (PROF) { __builtin_amdgcn_s_waitcnt(0); FSMG_STAMP(1); }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * q + r][l15] = acc0[r] + acc1[r];
    FSMG_STAMP(2);
    __syncthreads();
    FSMG_STAMP(3);

    if (eact) {
        float dh_rec = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) dh_rec += red[w][erow][eun];
        const CellGrad cg = cell_backward(si, tj, sf, so, ct, cp, dcv, dht + dh_rec);
        const float di = cg.di, dj = cg.dj, df = cg.df, dg = cg.dg;
        gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg;   // row-major dz for the weight-gradient GEMMs
        // fragment-ordered copy for step t-1: packed column 16*(u/4) + 4*gate + u%4 -> group u/4, slot q = gate
        float* fp = a.dzF_cur + (((size_t)blockIdx.y * (G4 >> 4) + (eu >> 2)) * 64 + erow) * 4 + (eu & 3);
        fp[0] = di; fp[64] = dj; fp[128] = df; fp[192] = dg;
        a.dc[hi] = cg.dc_out;
    }
    FSMG_STAMP(4);
}

// ---------------------------------------------------------------- backward, persistent over a range of time steps
// Block = 16 hidden units x one 16-row tile, 8 waves split K = 4Hp; the wave's slice of Kh (GPW float4 per lane)
// stays in registers, dc stays in a register, and dz_t travels between the Hp/16 blocks of a row tile exactly like
// h_t does in k_lstm_fwd_chain (one buffer per time step, pre-filled with the "not written" pattern).  The four
// gates of four units form one 16-byte fragment word, so the four lanes holding those units exchange their gate
// gradients with shuffles and each stores one float4.
template <int GPW>
__global__ __launch_bounds__(512, 2) void k_lstm_bwd_chain(const LstmBwdChainArgs a) {
    __shared__ float red[2][8][16][17];     // alternating by time step: see k_lstm_fwd_chain
    __shared__ int s_fail;
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int ug = blockIdx.x, rt = blockIdx.y;
    const int u0 = ug * 16, m0 = rt * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp, B = a.B;
    const int ngroups = G4 >> 4;
    const size_t dz_step = (size_t)gridDim.y * 16 * G4;          // floats per time index of dzF_all
    if (tid == 0) s_fail = 0;

    f32x4 bw[GPW];
    {
        const f32x4* bf = reinterpret_cast<const f32x4*>(a.KhF) + ((size_t)ug * ngroups + wave * GPW) * 64 + lane;
#pragma unroll
        for (int j = 0; j < GPW; ++j) bw[j] = bf[j * 64];
    }
    // epilogue mapping (waves 0-3): thread -> (row, unit)
    const int erow = tid >> 4, eun = tid & 15;
    const int eb = m0 + erow, eu = u0 + eun;
    const bool epi = tid < 256;
    const bool eact = epi && (eb < B);
    const long long hi = (long long)eb * Hp + eu;
    float dcv = eact ? a.dc[hi] : 0.0f;
    __syncthreads();

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        // everything the cell gradient needs besides dh_rec was produced by earlier launches: requested before the wait
        float si = 0.f, tj = 0.f, sf = 0.f, so = 0.f, ct = 0.f, cp = 0.f, dht = 0.f;
        float* gp = a.Z + ((size_t)t * B + eb) * G4 + 16 * (eu >> 2) + (eu & 3);
        if (eact) {
            si = gp[0]; tj = gp[4]; sf = gp[8]; so = gp[12];
            ct = a.Cs[(size_t)(t + 1) * B * Hp + hi]; cp = a.Cs[(size_t)t * B * Hp + hi];
            dht = a.dH[(size_t)t * B * Hp + hi];
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4 av[GPW];
            const f32x4* af = reinterpret_cast<const f32x4*>(a.dzF_all + (size_t)(t + 1) * dz_step) + ((size_t)rt * ngroups + wave * GPW) * 64 + lane;
            const bool fail = !wait_fragments<GPW>(af, av, a.spin_limit, a.err_flag);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                f32x4& acc = (j & 1) ? acc1 : acc0;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][0], bw[j][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][1], bw[j][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][2], bw[j][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][3], bw[j][3], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[t & 1][wave][4 * q + r][l15] = acc0[r] + acc1[r];
        __syncthreads();
        if (s_fail) return;

        if (epi) {
            float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
            if (eact) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) dh_rec += red[t & 1][w][erow][eun];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cp, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg;   // row-major dz for the weight-gradient GEMMs
                dcv = cg.dc_out;
            }
            // fragment word (group eu/4, lane 16*gate + row) = that gate of units 4*(eu/4) .. +3: lane k of the four
            // lanes that hold those units collects gate k of all four and stores it
            const int base = lane & ~3, k = lane & 3;
            f32x4 w4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float xi = __shfl(di, base + m), xj = __shfl(dj, base + m), xf = __shfl(df, base + m), xg = __shfl(dg, base + m);
                w4[m] = (k == 0) ? xi : (k == 1) ? xj : (k == 2) ? xf : xg;
            }
            f32x4* dst = reinterpret_cast<f32x4*>(a.dzF_all + (size_t)t * dz_step) + ((size_t)rt * ngroups + (eu >> 2)) * 64 + 16 * k + erow;
            store_sc1(dst, w4);
        }
    }
    if (eact) a.dc[hi] = dcv;
}

// ---------------------------------------------------------------- backward, persistent, reduce-scatter form
// See LstmBwdRsArgs.  Iteration t (descending) of block j = blockIdx.x, P = Hp/16 blocks per row tile, 8 waves:
//   A  consume: thread (wave w, lane) polls the partials of producers w*TPW .. w*TPW+TPW-1 for its own units (one
//      float4 = 4 rows of one unit column in MFMA C/D layout), puts the fill pattern back (the slot is reused two
//      steps later), adds them in producer order and leaves the wave's sum in LDS;
//   B  waves 0-3: dh_rec = sum over waves (in order) -> gate gradients of (row, unit) -> row-major dz for the GEMMs
//      and the block's own 16 x 64 dz slice in MFMA A-fragment order in LDS;
//   C  produce: wave w multiplies that slice with its TPW resident 64 x 16 weight tiles (one accumulator chain of
//      16 MFMAs per destination block) and stores the C/D registers straight into the destinations' inboxes with
//      16-byte write-through stores.  The resets of phase A are drained before the first of these stores, which is
//      what makes two slots enough: nobody can overwrite a slot before every reader of its previous content has put
//      the fill pattern back, because progress of any block depends on these stores.
// Summation order (producer-major, then wave-major) is the one k_lstm_bwd_step uses when Hp is a multiple of 128, so
// the two paths agree to the bit.
template <int TPW>
__global__ __launch_bounds__(512, 2) void k_lstm_bwd_rs(const LstmBwdRsArgs a) {
    __shared__ __attribute__((aligned(16))) float red[8][64][4];
    __shared__ __attribute__((aligned(16))) float frag[4][64][4];
    __shared__ int s_fail;
    FSMG_STEP_PRIO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x, rt = blockIdx.y, P = gridDim.x;
    const int u0 = j * 16, m0 = rt * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp, B = a.B;
    const int ngroups = G4 >> 4;
    if (tid == 0) s_fail = 0;

    // resident weights: for each of this wave's destination blocks i, the 4 k-groups of this block's 64 columns
    f32x4 bw[TPW][4];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bw[ti][g] = reinterpret_cast<const f32x4*>(a.KhF)[((size_t)(wave * TPW + ti) * ngroups + 4 * j + g) * 64 + lane];

    const size_t slot_f = (size_t)gridDim.y * P * P * 256;                       // floats per slot
    float* const inbox_rt = a.inbox + (size_t)rt * P * P * 256;
    // epilogue mapping (waves 0-3): thread -> (row, unit)
    const int erow = tid >> 4, eun = tid & 15;
    const int eb = m0 + erow, eu = u0 + eun;
    const bool epi = tid < 256;
    const bool eact = epi && (eb < B);
    const long long hi = (long long)eb * Hp + eu;
    float dcv = eact ? a.dc[hi] : 0.0f;
    __syncthreads();

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        float si = 0.f, tj = 0.f, sf = 0.f, so = 0.f, ct = 0.f, cp = 0.f, dht = 0.f;
        float* gp = a.Z + ((size_t)t * B + eb) * G4 + 16 * (eu >> 2) + (eu & 3);
        if (eact) {
            si = gp[0]; tj = gp[4]; sf = gp[8]; so = gp[12];
            ct = a.Cs[(size_t)(t + 1) * B * Hp + hi]; cp = a.Cs[(size_t)t * B * Hp + hi];
            dht = a.dH[(size_t)t * B * Hp + hi];
        }
        // ---- A: consume the partials of dh_t
        f32x4 wsum = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4* in = reinterpret_cast<f32x4*>(inbox_rt + (size_t)((t + 1) & 1) * slot_f) + ((size_t)j * P + wave * TPW) * 64 + lane;
            f32x4 v[TPW];
            bool fail = false;
            for (int spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < TPW; ++k) v[k] = load_sc1(in + k * 64);
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int k = 0; k < TPW; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
            const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                store_sc1(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        }
        *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = wsum;
        __syncthreads();
        if (s_fail) return;

        // ---- B: gate gradients; the block's dz slice goes to LDS in A-fragment order
        if (epi) {
            float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
            if (eact) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) dh_rec += red[w][eun + 16 * (erow >> 2)][erow & 3];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cp, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg;   // row-major dz for the weight-gradient GEMMs
                dcv = cg.dc_out;
            }
            // fragment [group eun/4][lane = 16*gate + row][eun%4]
            float* f = &frag[eun >> 2][erow][eun & 3];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
        }
        __syncthreads();

        // ---- C: partials of dh_{t-1} for every block of the row tile
        if (t > 0) {
            f32x4 af[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const f32x4*>(&frag[g][lane][0]);
            f32x4 acc[TPW];
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) acc[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti)
                        acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][e], bw[ti][g][e], acc[ti], 0, 0, 0);
            // the stores below are inline asm: hipcc does not know that they read the MFMA results, so the wait states
            // between an XDL write and a VMEM read of the same VGPRs (up to 19 for these MFMAs) are inserted by hand --
            // without them the stores picked up stale accumulator registers now and then (gradients off by ~1e-3)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            drain_vmem();                                             // the resets of phase A have landed
            f32x4* out = reinterpret_cast<f32x4*>(inbox_rt + (size_t)(t & 1) * slot_f) + (size_t)j * 64 + lane;
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) store_sc1(out + (size_t)(wave * TPW + ti) * P * 64, acc[ti]);
        }
    }
    if (eact) a.dc[hi] = dcv;
}

// Kh [Hp][4Hp] (packed gate columns) -> the two fragment-ordered copies the step kernels stream:
//   fwd: B[k][n = packed col], block nb = 16 cols:  KhF_fwd[nb][g][lane=16q+n][s] = Kh[16g+4q+s][16nb+n]
//   bwd: B[k = packed col][n = unit], block ug:      KhF_bwd[ug][g][lane=16q+n][s] = Kh[16ug+n][16g+4q+s]
__global__ void k_repack_kh(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int Hp) {
    repack_kh_chunked(Kh, fwd, bwd, Hp, blockIdx.x, gridDim.x);
}

}  // namespace

hipError_t launch_repack_kh(hipStream_t s, const float* Kh, float* fwd, float* bwd, int Hp) {
    const long long total = (long long)Hp * 4 * Hp / 4;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_repack_kh, dim3(blocks), dim3(256), 0, s, Kh, fwd, bwd, Hp);
    return hipGetLastError();
}

hipError_t launch_lstm_fwd_step(hipStream_t s, const LstmFwdArgs& a, unsigned long long* prof) {
    const int ncg = (4 * a.Hp) / 16, ntiles = (a.B + 15) / 16;
    if (prof == nullptr && ntiles >= 4 && (ncg & 1) == 0) {       // many rows: 2 x 2 patches of output tiles per block
        hipLaunchKernelGGL((k_lstm_fwd_patch<4, 2, 2>), dim3(ncg / 2, (ntiles + 1) / 2), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    dim3 grid(ncg, ntiles);
    if (prof) hipLaunchKernelGGL((k_lstm_fwd_step<true, FWD_NW>), grid, dim3(64 * FWD_NW), 0, s, a, prof);
    else hipLaunchKernelGGL((k_lstm_fwd_step<false, FWD_NW>), grid, dim3(64 * FWD_NW), 0, s, a, nullptr);
    return hipGetLastError();
}

// shapes the persistent forward kernel takes: K groups split evenly over the 4 waves into 1..16 per wave, and every
// block of the grid resident at once (with a margin: co-resident GEMM blocks only delay residency, but another
// persistent launch could hold slots for good)
bool lstm_fwd_chain_supported(int B, int Hp) {
    const int ngroups = Hp >> 4;
    if ((Hp & 15) || (ngroups & 3)) return false;
    const int gpw = ngroups >> 2;
    if (gpw != 1 && gpw != 2 && gpw != 4 && gpw != 8 && gpw != 16) return false;
    const long long blocks = (long long)(4 * Hp / 16) * ((B + 15) / 16);
    const int per_cu = gpw >= 16 ? 2 : 4;                      // register-limited residency of 256-thread blocks
    return blocks <= (long long)256 * per_cu * 9 / 10;
}

hipError_t launch_lstm_fwd_chain(hipStream_t s, const LstmFwdChainArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    dim3 grid((4 * a.Hp) / 16, (a.B + 15) / 16);
    switch ((a.Hp >> 4) >> 2) {
        case 1: hipLaunchKernelGGL((k_lstm_fwd_chain<1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lstm_fwd_chain<2>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_fwd_chain<4>), grid, dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_lstm_fwd_chain<8>), grid, dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_lstm_fwd_chain<16>), grid, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool lstm_fwd_chain_rt_supported(int B, int Hp) {
    const int ngroups = Hp >> 4;
    if ((Hp & 15) || (ngroups & 3)) return false;
    const int gpw = ngroups >> 2;
    if (gpw != 1 && gpw != 2 && gpw != 4 && gpw != 8 && gpw != 16) return false;
    const int rt = (B + 15) / 16;
    if (rt < 2 || rt > 4) return false;        // 7 row tiles (cfg-D) measured 12 us per step against 6.1 with one launch per step
    return (long long)(4 * Hp / 16) <= (long long)256 * (gpw >= 16 ? 1 : 3);
}

template <int GPW>
static hipError_t launch_fwd_chain_rt_t(hipStream_t s, const LstmFwdChainArgs& a) {
    const int rt = (a.B + 15) / 16;
    const size_t lds = (size_t)2 * rt * 4 * 16 * 17 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lstm_fwd_chain_rt<GPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 16 * 17 * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_lstm_fwd_chain_rt<GPW>), dim3((4 * a.Hp) / 16), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_lstm_fwd_chain_rt(hipStream_t s, const LstmFwdChainArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    switch ((a.Hp >> 4) >> 2) {
        case 1: return launch_fwd_chain_rt_t<1>(s, a);
        case 2: return launch_fwd_chain_rt_t<2>(s, a);
        case 4: return launch_fwd_chain_rt_t<4>(s, a);
        case 8: return launch_fwd_chain_rt_t<8>(s, a);
        case 16: return launch_fwd_chain_rt_t<16>(s, a);
        default: return hipErrorInvalidValue;
    }
}

bool lstm_bwd_chain_supported(int B, int Hp) {
    const int ngroups = (4 * Hp) >> 4;
    if ((Hp & 15) || (ngroups & 7)) return false;
    const int gpw = ngroups >> 3;
    if (gpw != 2 && gpw != 4 && gpw != 8 && gpw != 16) return false;
    const long long blocks = (long long)(Hp / 16) * ((B + 15) / 16);
    return blocks <= (long long)256 * 3 / 4;                   // one 512-thread block per CU, with a margin
}

hipError_t launch_lstm_bwd_chain(hipStream_t s, const LstmBwdChainArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    dim3 grid(a.Hp / 16, (a.B + 15) / 16);
    switch (((4 * a.Hp) >> 4) >> 3) {
        case 2: hipLaunchKernelGGL((k_lstm_bwd_chain<2>), grid, dim3(512), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_bwd_chain<4>), grid, dim3(512), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_lstm_bwd_chain<8>), grid, dim3(512), 0, s, a); break;
        case 16: hipLaunchKernelGGL((k_lstm_bwd_chain<16>), grid, dim3(512), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool lstm_bwd_rs_supported(int B, int Hp) {
    if (Hp % 128) return false;
    const int tpw = Hp / 128;
    if (tpw != 1 && tpw != 2 && tpw != 4 && tpw != 8) return false;
    return (long long)(Hp / 16) * ((B + 15) / 16) <= (long long)256 * 9 / 10;  // one 512-thread block per CU, with a margin
}
long long lstm_bwd_rs_inbox_floats(int B, int Hp) {
    const long long P = Hp / 16;
    return 2LL * ((B + 15) / 16) * P * P * 256;
}
hipError_t launch_lstm_bwd_rs(hipStream_t s, const LstmBwdRsArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    dim3 grid(a.Hp / 16, (a.B + 15) / 16);
    switch (a.Hp / 128) {
        case 1: hipLaunchKernelGGL((k_lstm_bwd_rs<1>), grid, dim3(512), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lstm_bwd_rs<2>), grid, dim3(512), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_bwd_rs<4>), grid, dim3(512), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_lstm_bwd_rs<8>), grid, dim3(512), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lstm_bwd_step(hipStream_t s, const LstmBwdArgs& a, unsigned long long* prof) {
    dim3 grid(a.Hp / 16, (a.B + 15) / 16);
    if (prof) hipLaunchKernelGGL((k_lstm_bwd_step<true, BWD_NW>), grid, dim3(64 * BWD_NW), 0, s, a, prof);
    else hipLaunchKernelGGL((k_lstm_bwd_step<false, BWD_NW>), grid, dim3(64 * BWD_NW), 0, s, a, nullptr);
    return hipGetLastError();
}

}  // namespace fsmg
