// XCD-local persistent recurrence for hidden size 512 (cfg-B / cfg-D of BASELINE.json): the fused LSTM cell
// (reference src/models/lstm_baseline.py:44-55) restructured around the chip instead of the GEMM.
//
// The column-split persistent kernels (lstm_step.hip) spread the 2048 gate columns of ONE row tile over the whole
// chip, so every time step is a cross-XCD hand-off through write-through stores (~1.1 us).  Here the ROWS are split
// instead: each of the 8 XCDs keeps a FULL copy of K_h (4 MiB fp32) in the registers of its 32 CUs and owns
// ceil(B/8) sequences.  h_t of those sequences is produced and consumed inside one XCD, so the hand-off goes through
// that XCD's L2 with plain stores (no write-through, no fabric hop) and no block ever waits on another XCD.
//
//   CU c of an XCD  : hidden units 16c .. 16c+15 = packed gate columns 64c .. 64c+63 (all four gates of its units)
//   wave w of a CU  : K range 128w .. 128w+127 of the contraction, 128 weight VGPRs per lane, resident for the launch
//   MFMA            : v_mfma_f32_4x4x1_16B_f32 -- sixteen 4x4 outer products per instruction.  Block = 4 columns; the
//                     four rows of a row group are BROADCAST from one block of the A register (cbsz = 4, abid = b), so
//                     one A VGPR carries 16 k's x 4 rows and a row group costs 8 cycles per k: a 6-row slice of the
//                     batch pays for 8 rows instead of the 16 a 16x16x4 tile would charge.
//   per step        : 128 k x RG row groups MFMAs per wave (RG = 2 at B = 45: 2048 cycles), K-split partials meet in
//                     LDS, the cell update of (row, unit) runs on 64*RG threads, h_{t+1} goes out as 16-byte words in
//                     the exact order the consumers' A registers want them.
//
// What bounds a step (tools/xcd_chain_bench.cpp; B = 45, ticks of the 2.35 GHz shader clock): 2176 MFMA + ~260 LDS hand-
// over and barrier + ~870 cell update + the hand-off.  The hand-off is a store into L2 and a poll out of it (0.26 us
// between two idle CUs) -- but a CU's vector-memory pipeline returns loads IN ORDER, so a poll queues behind every
// slower load the same CU has in flight.  Hence the two rules of these kernels: the inputs of the cell update (x-part
// pre-activations forward; gates, cell states and dH backward) are requested only AFTER a poll has succeeded, i.e. at the
// start of the MFMA phase that hides them, never in front of a poll; and a poll fetches every fragment at once (one
// round trip) instead of probing one and then fetching the rest.
// Tried and dropped (profiles/r02_xcd_probe3..10.log): running the row groups as interleaved chains with dedicated
// cell waves beside the MFMA waves (software pipeline, LDS counters instead of barriers) -- an MFMA wave and a VALU
// wave on one SIMD share the issue port, the cell update took 2-3x longer and the MFMA phase 1.3x, the sum was a wash
// (2.5-2.9 us per step against 2.6); wave priorities, write-through stores, sleeping or not between polls: +-2 %.
//
// Round 3 (profiles/r03_xcd_probe1..6*.log).  Kept: (a) XCD_DEFER_OUTPUTS -- the six output stores of a cell thread (c, h,
// four gates) leave the queue between the hand-off store and the next poll (whose vmcnt(0) waited for them) and are issued
// behind that poll's success instead, under the MFMAs (forward 2.21 -> 2.14 us per step at B = 45; +4 % at B = 100, so only up
// to two row groups); (b) XCD_NO_POLL_SLEEP (backward 2.40 -> 2.34).  Measured and dropped, each a full rewrite that passed the
// fp64 check: the row groups of an XCD as TWO chains advanced alternately by the same four waves, the poll of one chain issued
// under the other chain's MFMAs with hand-counted `s_waitcnt vmcnt(N)` and fixed landing registers for the in-flight loads
// (hand-off latency hidden, but every fixed cost of a step -- readiness check, LDS exchange + barrier, the cell update's
// dependent chain -- is paid per phase, i.e. twice: 2.65 / 2.85, and still 2.32 / 2.71 with all waits and all slow traffic
// compiled out); the cell update spread over all 256 threads, one gate per lane with DPP row shifts (more instructions per
// wave than two cell waves doing whole cells: 2.48-2.52 / 2.80-2.90); a leaner tanh without the small-|x| polynomial
// (no change); v_rcp_f32 instead of the correctly rounded reciprocal in the five nonlinearities of a cell update (45 fewer
// dependent VALU instructions per step: no change either -- 2.14-2.17 / 2.35-2.43, profiles/r03_xcd_probe7_rcp.log -- and one
// Adam-step comparison moves to 1.002e-4 against its 1e-4 bound, so the exact one stays).  The step is now ~2176 ticks of MFMA + ~330 LDS/barrier + ~800 cell + ~1300 from hand-off store to
// the consumers' first successful poll (646 between idle CUs + poll granularity + skew of 32 CUs), and the last term is
// what a one-chain-per-XCD design cannot hide.
//
// Placement is discovered, not assumed: a block reads its XCC id and takes a ticket from that XCD's counter; the
// (xcd, ticket) pair is its role.  HIP promises nothing about block -> XCD placement, so a ticket >= 32 (an XCD that
// received more than its share) raises the time-out flag like any other failed wait and the caller falls back to the
// column-split kernels: placement decides speed, never results.
#include "fsmg_kernels.h"
#include "lstm_cell.h"
#include "lstm_repack.h"
#include <type_traits>
#include <cstdlib>
#include <algorithm>

namespace fsmg {
namespace {

constexpr int XH = 512;            // padded hidden size these kernels are built for
constexpr int XG4 = 4 * XH;
constexpr int NXCD = 8, NCU = 32;  // XCDs per chip, CUs (= blocks) per XCD

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

// 16 B store that STAYS in the XCD's L2 (plain scope): the consumers are CUs of the same XCD reading with sc1 (L1
// bypass), so the L2 is the point of coherence.  Same hand-written-asm hazard padding as store_sc1.
__device__ __forceinline__ void store_l2(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// row-major h output of a cell thread.  wt: agent-scope write-through -- a GEMM on ANOTHER XCD reads these rows while the launch
// still runs (the overlapped step: LstmFwdXcdArgs::progress), so they must not linger in this XCD's L2
__device__ __forceinline__ void store_h_row(float* p, float v, bool wt) {
    if (wt) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else *p = v;
}

template <int ABID>
__device__ __forceinline__ f32x4 mfma44(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);       // A of block ABID broadcast to all 16 blocks
}

// value of lane (quad base + K) of every quad
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, true));
}

// acc[rg] += A[rg][.] x W[.]: NK k-groups of 16 k's; A register j*NRG.. see callers.  One macro per abid because the
// broadcast selector is an immediate.
#define XCD_MFMA_B(B_, AV, WV, ACC)                                                     \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                    \
        ACC = mfma44<B_>((AV)[e_], (WV)[e_], ACC);

// One-round-trip poll: fetches all N hand-off fragments (af, af + 64, ...) every round until none shows the fill pattern.
template <int N>
__device__ __forceinline__ bool wait_all_fragments(const f32x4* af, f32x4 (&av)[N], int spin_limit, int* err_flag, bool nosleep = false) {
    for (int spins = 0;; ++spins) {
#pragma unroll
        for (int j = 0; j < N; ++j) av[j] = load_sc1(af + j * 64);
        drain_vmem();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N; ++j) { asm volatile("" : "+v"(av[j])); ok &= frag_ready(av[j]); }
        if (__all(ok)) return true;
        if (!nosleep) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
        if (spins >= spin_limit || ((spins & 255) == 255 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) return false;
    }
}

struct Role { int xcd, cu; };

// every block: XCC id + a ticket from that XCD's counter.  Returns false (and raises the flag) when the XCD is over-subscribed.
__device__ __forceinline__ bool take_role(int* tickets, int* err_flag, int* s_role, Role& r) {
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        const int slot = atomicAdd(&tickets[x], 1);
        s_role[0] = x; s_role[1] = slot;
        if (slot >= NCU) __hip_atomic_store(err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    r.xcd = s_role[0]; r.cu = s_role[1];
    return r.cu < NCU;
}

// ---------------------------------------------------------------- forward
// HX  [T+1][8 xcd][4 w][RG][2 q][64 lanes][4]: h in A-register order.  Lane 4b+i, component e of (w, rg, q) holds
//     h[row 4rg+i of the XCD][unit 128w + 64q + 16(b/4) + 4(b%4) + e]; CU c writes the 16 lanes 16(c%4) .. +15 of
//     (w = c/8, q = (c/4)%2) as 256 contiguous bytes.  Index 0 is the zero state, indices t0+1 .. t1 are pre-filled
//     with the "not written" pattern.
// KhX [32 cu][4 w][32][64 lanes][4]: register image of the weights, see k_repack_kh_xcd.
// PROF: per (block, wave) sums of s_memtime ticks over the steps: [0] wait for h_t, [1] MFMAs, [2] partials -> LDS + barrier,
// [3] cell update up to the hand-off store, [4] rest of the step (diagnostic build, tools/xcd_chain_bench.cpp)
#define XCD_STAMP(i) if (PROF) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; }
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_xcd(const LstmFwdXcdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][RG][4][64 * 4];   // [step parity][row group][wave][cell lane][gate]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B, T = a.T;
    (void)T;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;                        // XCD without rows (packed split): leave its CUs to other streams

    f32x4 W[32];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
    }
    // cell threads: wave rg < RG owns row group rg; lane = 16 i + 4 bb + e -> row 4rg+i, unit 16cu + 4bb + e
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * XH + unit] : 0.0f;
    const size_t hx_step = (size_t)NXCD * 4 * RG * 2 * 64;                  // f32x4 words per time index
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)xcd * 4 + wave) * RG) * 2 * 64 + lane;
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) + ((((size_t)xcd * 4 + (cu >> 3)) * RG + wave) * 2 + ((cu >> 2) & 1)) * 64 +
                    16 * (cu & 3) + 4 * cbb + ci;
    // D layout of a 4x4 block: lane = 4*block + column, register = row.  Lane l = local packed column: unit block l/16,
    // gate (l/4)%4, unit%4 = l%4; the cell lane of (row i, unit) is 16 i + 4 (l/16) + l%4 and reads its four gates as
    // one 16-byte word per wave
    const int wofs = ((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3);
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    float zq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // x-part pre-activations of steps t, t+1 (in flight)
    float o_c = 0.f, o_hh = 0.f, o_g[4] = {0.f, 0.f, 0.f, 0.f};          // variant 16: outputs of the last step, stored one step later
    bool o_have = false;
    if (act) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.t0 + k < a.t1) {
                const float* zn = a.Z + ((size_t)(a.t0 + k) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
#pragma unroll
                for (int g = 0; g < 4; ++g) zq[k][g] = zn[4 * g];
            }
    }

    for (int t = a.t0; t < a.t1; ++t) {
        XCD_STAMP(4)
        f32x4 av[2 * RG];
        {
            const bool fail = !wait_all_fragments<2 * RG>(hx_in + (size_t)t * hx_step, av, a.spin_limit, a.err_flag, (a.variant & XCD_NO_POLL_SLEEP) != 0);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {              // outputs of the step before: behind a successful poll, under the MFMAs
            a.Cs[((size_t)t * B + row) * XH + unit] = o_c;
            a.Hs[((size_t)t * B + row) * XH + unit] = o_hh;
            float* zo = a.Z + ((size_t)(t - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
        }
        // x-part pre-activations: the loads of step t+2 are requested now that the poll of step t is over (never in front
        // of a poll in the CU's in-order memory pipeline) -- two steps ahead because one of these scattered, TLB-cold
        // loads takes ~3200 ticks, longer than the MFMA phase (first version: the barrier waited ~1000 ticks for them)
        float zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { zin[g] = zq[0][g]; zq[0][g] = zq[1][g]; zq[1][g] = 0.0f; }
        float* zp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        if (act && t + 2 < a.t1) {
            const float* zn = zp + 2 * (size_t)B * XG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        XCD_STAMP(0)
        f32x4 acc[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#define XCD_FWD_B(B_)                                                                   \
        _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) { XCD_MFMA_B(B_, av[2 * rg + q], W[16 * q + B_], acc[rg]) }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            XCD_FWD_B(0) XCD_FWD_B(1) XCD_FWD_B(2) XCD_FWD_B(3) XCD_FWD_B(4) XCD_FWD_B(5) XCD_FWD_B(6) XCD_FWD_B(7)
            XCD_FWD_B(8) XCD_FWD_B(9) XCD_FWD_B(10) XCD_FWD_B(11) XCD_FWD_B(12) XCD_FWD_B(13) XCD_FWD_B(14) XCD_FWD_B(15)
        }
#undef XCD_FWD_B
        if (PROF) { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); acc[0][0] += 0.0f; }
        XCD_STAMP(1)
        {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                float* rp = &red[t & 1][rg][wave][0] + wofs;
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[64 * i] = acc[rg][i];
            }
        }
        __syncthreads();
        if (s_fail) return;
        XCD_STAMP(2)

        if (cellw) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[t & 1][wave][0][0]) + lane;
                const f32x4 r0 = rsrc[0], r1 = rsrc[64], r2 = rsrc[128], r3 = rsrc[192];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            // the four units of a quad form one 16-byte word of the hand-off; pad rows publish zeros so that every word
            // of the buffer is written and the readers' test terminates
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            if (ce == 0) store_l2(hx_out + (size_t)(t + 1) * hx_step, hv);
            XCD_STAMP(3)
            if (a.variant & XCD_DEFER_OUTPUTS) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * XH + unit] = cp;
                a.Hs[((size_t)(t + 1) * B + row) * XH + unit] = hn;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;       // activated gates kept for BPTT
            }
        }
    }
    if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
        a.Cs[((size_t)a.t1 * B + row) * XH + unit] = o_c;
        a.Hs[((size_t)a.t1 * B + row) * XH + unit] = o_hh;
        float* zo = a.Z + ((size_t)(a.t1 - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
    }
    if (PROF && lane == 0 && a.prof) {
        XCD_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}

// ---------------------------------------------------------------- backward (reduce-scatter inside the XCD)
// Block (xcd, cu) keeps the dz of its 16 units to itself; what travels is dh.  Iteration t (descending):
//   A  consume: the 32 partials of dh_t for its own units (inbox, 16-byte words = 4 rows of one unit), fill pattern put
//      back, fixed-order sums left in LDS;
//   B  cell threads: dh_rec -> gate gradients -> row-major dz for the GEMMs + the block's RG x (4 rows x 64 columns) dz
//      slice in A-register order in LDS;
//   C  produce: wave w multiplies the slice with its resident 64 x 128 slice of K_h^T (destination units 128w .. +127)
//      and stores the 4x4-block results straight from the MFMA registers into the destinations' inboxes (plain
//      stores: same XCD).  The resets of phase A are drained before these stores; two slots suffice because progress
//      of every block depends on every other block's publish (same argument as k_lstm_bwd_rs).
// inbox [2 slots][8 xcd][32 dest][32 producer][RG][16 units][4 rows].
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_xcd(const LstmBwdXcdArgs a) {
    constexpr int NG = 4 / RG;                    // lane groups of a wave that read different producers of one row group
    constexpr int LPW = 2 * RG;                   // inbox words per lane: 8 producers x RG x 16 units / 64 lanes
    __shared__ __attribute__((aligned(16))) float psum[4 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float dzA[RG][64][4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;

    f32x4 W[32];          // component e' of word i = weight register 4i + e' = (cg = /64, k = %64): Kh[128w + 64cg + lane][64cu + k]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const size_t hi = (size_t)row * XH + unit;
    float dcv = act ? a.dc[hi] : 0.0f;
    const size_t slot_w = (size_t)NXCD * NCU * NCU * RG * 16;               // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    // consumer side: this block's 32 x RG x 16 words; wave w takes producers 8w .. 8w+7, LPW words per lane
    const size_t in_base = (((size_t)xcd * NCU + cu) * NCU + 8 * wave) * RG * 16 + lane;
    // producer side: lane l of (rg, cg) -> destination 8w + 4cg + l/16, word (dest, producer = cu, rg, l%16)
    size_t out_ofs[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
        out_ofs[cg] = ((((size_t)xcd * NCU + 8 * wave + 4 * cg + (lane >> 4)) * NCU + cu) * RG) * 16 + (lane & 15);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;

    // cell inputs of the iteration to come: loaded during the MFMA phase of the iteration before (and here for the first
    // one), never between a block's publish and its next poll
    float n_si = 0.f, n_tj = 0.f, n_sf = 0.f, n_so = 0.f, n_ct = 0.f, n_cp = 0.f, n_dh = 0.f;
    if (act && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        const float* gp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        n_si = gp[0]; n_tj = gp[4]; n_sf = gp[8]; n_so = gp[12];
        n_ct = a.Cs[(size_t)(t + 1) * B * XH + hi]; n_cp = a.Cs[(size_t)t * B * XH + hi];
        n_dh = a.dH[(size_t)t * B * XH + hi];
    }

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        XCD_STAMP(4)
        const float si = n_si, tj = n_tj, sf = n_sf, so = n_so, ct = n_ct, cpv = n_cp, dht = n_dh;
        float* gp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        // ---- A: consume
        f32x4 wsum = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base;
            f32x4 v[LPW];
            bool fail = false;
            for (int spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < LPW; ++k) v[k] = load_sc1(in + k * 64);
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int k = 0; k < LPW; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                if (__all(ok)) break;
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
#pragma unroll
            for (int k = 0; k < LPW; ++k) {
                store_l2(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        }
        XCD_STAMP(0)
        *reinterpret_cast<f32x4*>(&psum[(wave * 64 + lane) * 4]) = wsum;
        __syncthreads();
        if (s_fail) return;
        XCD_STAMP(1)

        // ---- B: gate gradients (wave rg < RG: lane = 16 i + 4 bb + e)
        float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
        if (cellw) {
            if (act) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp)
                        dh_rec += psum[(w * 64 + (grp * RG + wave) * 16 + 4 * cbb + ce) * 4 + ci];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cpv, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                if (!(a.variant & XCD_DEFER_OUTPUTS)) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }       // row-major dz for the weight-gradient GEMMs
                dcv = cg.dc_out;
            }
            // A-register order: local column k = 16bb + 4g + e -> register v = bb, block b = 4g + e, lane 4b + i
            float* f = &dzA[wave][4 * ce + ci][cbb];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
        }
        __syncthreads();
        XCD_STAMP(2)
        // the resets of phase A (and the dz stores of phase B) must have landed before this block publishes anything: wait for
        // them HERE, before the prefetch below puts ~3200-cycle loads into the queue that a drain behind the MFMAs would also wait for
        drain_vmem();
        if ((a.variant & XCD_DEFER_OUTPUTS) && act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }   // variant: dz stores behind the drain, under the MFMAs
        if (act && t > a.t0) {             // prefetch for iteration t-1 (hidden by the MFMAs below)
            const float* gn = a.Z + ((size_t)(t - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            n_si = gn[0]; n_tj = gn[4]; n_sf = gn[8]; n_so = gn[12];
            n_ct = cpv; n_cp = a.Cs[(size_t)(t - 1) * B * XH + hi];
            n_dh = a.dH[(size_t)(t - 1) * B * XH + hi];
        }

        // ---- C: produce the partials of dh_{t-1}
        if (t > 0) {
            f32x4 av[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) av[rg] = *reinterpret_cast<const f32x4*>(&dzA[rg][lane][0]);
            f32x4 acc[RG][2];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) { acc[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // k = 16 v + b: A register av[rg][v], broadcast block b; weight register cg*64 + k = word cg*16 + 4v + b/4, component b%4
#define XCD_BWD_B(B_)                                                                                   \
            _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) {                                         \
                acc[rg][0] = mfma44<B_>(av[rg][v], W[4 * v + (B_ >> 2)][B_ & 3], acc[rg][0]);            \
                acc[rg][1] = mfma44<B_>(av[rg][v], W[16 + 4 * v + (B_ >> 2)][B_ & 3], acc[rg][1]);       \
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                XCD_BWD_B(0) XCD_BWD_B(1) XCD_BWD_B(2) XCD_BWD_B(3) XCD_BWD_B(4) XCD_BWD_B(5) XCD_BWD_B(6) XCD_BWD_B(7)
                XCD_BWD_B(8) XCD_BWD_B(9) XCD_BWD_B(10) XCD_BWD_B(11) XCD_BWD_B(12) XCD_BWD_B(13) XCD_BWD_B(14) XCD_BWD_B(15)
            }
#undef XCD_BWD_B
            // the stores are inline asm: the wait states between an MFMA writing VGPRs and a VMEM store reading them
            // are inserted by hand
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            XCD_STAMP(3)
            f32x4* out = inbox + (size_t)(t & 1) * slot_w;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                store_l2(out + out_ofs[0] + rg * 16, acc[rg][0]);
                store_l2(out + out_ofs[1] + rg * 16, acc[rg][1]);
            }
        }
    }
    if (act) a.dc[hi] = dcv;
    if (PROF && lane == 0 && a.prof) {
        XCD_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}
#undef XCD_STAMP

// Kh [512][2048] (packed gate columns) -> the register images of the two XCD-local kernels:
//   fwd word i (= 16q + b), component e of (cu, w), lane l:  Kh[128w + 64q + 16(b/4) + 4(b%4) + e][64cu + l]
//   bwd word i, component e' of (cu, w), lane l, r = 4i + e' = 64cg + k:  Kh[128w + 64cg + l][64cu + k]
__device__ __forceinline__ void repack_kh_xcd_body(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int b, int nb) {
    const int total = NCU * 4 * 32 * 64;            // f32x4 words per copy
    for (int idx = b * blockDim.x + threadIdx.x; idx < total; idx += nb * blockDim.x) {
        const int l = idx & 63, i = (idx >> 6) & 31, w = (idx >> 11) & 3, cu = idx >> 13;
        {
            const int q = i >> 4, b = i & 15;
            const int k0 = 128 * w + 64 * q + 16 * (b >> 2) + 4 * (b & 3);
            const float* src = Kh + (size_t)k0 * XG4 + 64 * cu + l;
            float4 v;
            v.x = src[0]; v.y = src[XG4]; v.z = src[2 * (size_t)XG4]; v.w = src[3 * (size_t)XG4];
            reinterpret_cast<float4*>(fwd)[idx] = v;
        }
        {
            const int r = 4 * i, cg = r >> 6, k = r & 63;
            reinterpret_cast<float4*>(bwd)[idx] =
                *reinterpret_cast<const float4*>(Kh + (size_t)(128 * w + 64 * cg + l) * XG4 + 64 * cu + k);
        }
    }
}
__global__ void k_repack_kh_xcd(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    repack_kh_xcd_body(Kh, fwd, bwd, blockIdx.x, gridDim.x);
}

// ================================================================ hidden size 256: two row slices per XCD (round 4)
// The reference's own default is hidden_size 200 (src/config/lstm_baseline.yaml:17), which pads to 256 here.  K_h of 256 units is
// 1 MiB: a copy fits the registers of 16 CUs (64 KiB each), so every XCD holds TWO copies and the batch is split SIXTEEN ways --
// 3 rows per slice at B = 45, one row group.  The kernels are the hidden-512 ones (k_lstm_fwd_xcd / k_lstm_bwd_xcd) with a 64-wide
// K range per wave (16 weight words instead of 32, one hand-off fragment per row group instead of two) and a hand-off that stays
// inside 16 CUs of one XCD's L2:
//   role      (xcd, ticket) -> slice 2 xcd + ticket / 16, CU ticket % 16 of the slice (hidden units 16 cu ..., 64 packed gate columns)
//   HX        [T+1][16 slices][4 w][RG][64 lanes][4]: lane 4b+i, component e of (w, rg) = h[row 4rg+i][unit 64w + 16(b/4) + 4(b%4) + e];
//             CU c writes lanes 16(c%4) .. +15 of w = c / 4
//   KhS       [16 cu][4 w][16 words][64 lanes][4] per direction (k_repack_kh_slice)
//   inbox     [2 slots][16 slices][16 dest][16 producers][RG][16 units][4 rows]
constexpr int SH = 256, SG4 = 4 * SH, SCU = 16, NSL = 2 * NXCD;

template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_slice(const LstmFwdXcdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][RG][4][64 * 4];   // [step parity][row group][wave][cell lane][gate]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int slice = 2 * role.xcd + (role.cu >> 4), cu = role.cu & 15;
    const int B = a.B;
    const int rps = (B + NSL - 1) / NSL, row0 = slice * rps;
    if (row0 >= B) return;

    f32x4 W[16];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rps && row < B;
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * SH + unit] : 0.0f;
    const size_t hx_step = (size_t)NSL * 4 * RG * 64;                       // f32x4 words per time index
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)slice * 4 + wave) * RG) * 64 + lane;
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) + (((size_t)slice * 4 + (cu >> 2)) * RG + wave) * 64 + 16 * (cu & 3) + 4 * cbb + ci;
    const int wofs = ((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3);
    float zq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float o_c = 0.f, o_hh = 0.f, o_g[4] = {0.f, 0.f, 0.f, 0.f};
    bool o_have = false;
    if (act) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.t0 + k < a.t1) {
                const float* zn = a.Z + ((size_t)(a.t0 + k) * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
#pragma unroll
                for (int g = 0; g < 4; ++g) zq[k][g] = zn[4 * g];
            }
    }

    for (int t = a.t0; t < a.t1; ++t) {
        f32x4 av[RG];
        {
            const bool fail = !wait_all_fragments<RG>(hx_in + (size_t)t * hx_step, av, a.spin_limit, a.err_flag, (a.variant & XCD_NO_POLL_SLEEP) != 0);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {              // outputs of the step before: behind a successful poll, under the MFMAs
            a.Cs[((size_t)t * B + row) * SH + unit] = o_c;
            a.Hs[((size_t)t * B + row) * SH + unit] = o_hh;
            float* zo = a.Z + ((size_t)(t - 1) * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
            zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
        }
        float zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { zin[g] = zq[0][g]; zq[0][g] = zq[1][g]; zq[1][g] = 0.0f; }
        float* zp = a.Z + ((size_t)t * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
        if (act && t + 2 < a.t1) {           // (behind the poll, two steps ahead: the rules of k_lstm_fwd_xcd)
            const float* zn = zp + 2 * (size_t)B * SG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        f32x4 acc[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#define SLICE_FWD_B(B_)                                                                 \
        _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) { XCD_MFMA_B(B_, av[rg], W[B_], acc[rg]) }
        SLICE_FWD_B(0) SLICE_FWD_B(1) SLICE_FWD_B(2) SLICE_FWD_B(3) SLICE_FWD_B(4) SLICE_FWD_B(5) SLICE_FWD_B(6) SLICE_FWD_B(7)
        SLICE_FWD_B(8) SLICE_FWD_B(9) SLICE_FWD_B(10) SLICE_FWD_B(11) SLICE_FWD_B(12) SLICE_FWD_B(13) SLICE_FWD_B(14) SLICE_FWD_B(15)
#undef SLICE_FWD_B
        {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                float* rp = &red[t & 1][rg][wave][0] + wofs;
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[64 * i] = acc[rg][i];
            }
        }
        __syncthreads();
        if (s_fail) return;

        if (cellw) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[t & 1][wave][0][0]) + lane;
                const f32x4 r0 = rsrc[0], r1 = rsrc[64], r2 = rsrc[128], r3 = rsrc[192];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            if (ce == 0) store_l2(hx_out + (size_t)(t + 1) * hx_step, hv);
            if (a.variant & XCD_DEFER_OUTPUTS) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * SH + unit] = cp;
                a.Hs[((size_t)(t + 1) * B + row) * SH + unit] = hn;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;
            }
        }
    }
    if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
        a.Cs[((size_t)a.t1 * B + row) * SH + unit] = o_c;
        a.Hs[((size_t)a.t1 * B + row) * SH + unit] = o_hh;
        float* zo = a.Z + ((size_t)(a.t1 - 1) * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
        zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
    }
}

template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_slice(const LstmBwdXcdArgs a) {
    constexpr int NG = 4 / RG;                    // lane groups of a wave that read different producers of one row group
    constexpr int LPW = RG;                       // inbox words per lane: 4 producers x RG x 16 units / 64 lanes
    __shared__ __attribute__((aligned(16))) float psum[4 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float dzA[RG][64][4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int slice = 2 * role.xcd + (role.cu >> 4), cu = role.cu & 15;
    const int B = a.B;
    const int rps = (B + NSL - 1) / NSL, row0 = slice * rps;
    if (row0 >= B) return;

    f32x4 W[16];          // component e' of word i = weight register k = 4i + e': Kh[64w + lane][64cu + k]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rps && row < B;
    const size_t hi = (size_t)row * SH + unit;
    float dcv = act ? a.dc[hi] : 0.0f;
    const size_t slot_w = (size_t)NSL * SCU * SCU * RG * 16;                // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    // consumer side: this block's 16 x RG x 16 words; wave w takes producers 4w .. 4w+3, LPW words per lane
    const size_t in_base = (((size_t)slice * SCU + cu) * SCU + 4 * wave) * RG * 16 + lane;
    // producer side: lane l -> destination CU 4w + l/16, word (dest, producer = cu, rg, l%16)
    const size_t out_ofs = ((((size_t)slice * SCU + 4 * wave + (lane >> 4)) * SCU + cu) * RG) * 16 + (lane & 15);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};

    float n_si = 0.f, n_tj = 0.f, n_sf = 0.f, n_so = 0.f, n_ct = 0.f, n_cp = 0.f, n_dh = 0.f;
    if (act && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        const float* gp = a.Z + ((size_t)t * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
        n_si = gp[0]; n_tj = gp[4]; n_sf = gp[8]; n_so = gp[12];
        n_ct = a.Cs[(size_t)(t + 1) * B * SH + hi]; n_cp = a.Cs[(size_t)t * B * SH + hi];
        n_dh = a.dH[(size_t)t * B * SH + hi];
    }

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        const float si = n_si, tj = n_tj, sf = n_sf, so = n_so, ct = n_ct, cpv = n_cp, dht = n_dh;
        float* gp = a.Z + ((size_t)t * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
        // ---- A: consume
        f32x4 wsum = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base;
            f32x4 v[LPW];
            bool fail = false;
            for (int spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < LPW; ++k) v[k] = load_sc1(in + k * 64);
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int k = 0; k < LPW; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                if (__all(ok)) break;
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
#pragma unroll
            for (int k = 0; k < LPW; ++k) {
                store_l2(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        }
        *reinterpret_cast<f32x4*>(&psum[(wave * 64 + lane) * 4]) = wsum;
        __syncthreads();
        if (s_fail) return;

        // ---- B: gate gradients (wave rg < RG: lane = 16 i + 4 bb + e)
        float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
        if (cellw) {
            if (act) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp)
                        dh_rec += psum[(w * 64 + (grp * RG + wave) * 16 + 4 * cbb + ce) * 4 + ci];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cpv, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg;          // row-major dz for the weight-gradient GEMMs
                dcv = cg.dc_out;
            }
            float* f = &dzA[wave][4 * ce + ci][cbb];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
        }
        __syncthreads();
        drain_vmem();                      // the resets of phase A and the dz stores: landed before this block publishes anything
        if (act && t > a.t0) {             // prefetch for iteration t-1 (hidden by the MFMAs below)
            const float* gn = a.Z + ((size_t)(t - 1) * B + row) * SG4 + 64 * cu + 16 * cbb + ce;
            n_si = gn[0]; n_tj = gn[4]; n_sf = gn[8]; n_so = gn[12];
            n_ct = cpv; n_cp = a.Cs[(size_t)(t - 1) * B * SH + hi];
            n_dh = a.dH[(size_t)(t - 1) * B * SH + hi];
        }

        // ---- C: produce the partials of dh_{t-1}
        if (t > 0) {
            f32x4 av[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) av[rg] = *reinterpret_cast<const f32x4*>(&dzA[rg][lane][0]);
            f32x4 acc[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
            // k = 16 v + b: A register av[rg][v], broadcast block b; weight register k = word 4v + b/4, component b%4
#define SLICE_BWD_B(B_)                                                                                 \
            _Pragma("unroll") for (int rg = 0; rg < RG; ++rg)                                           \
                acc[rg] = mfma44<B_>(av[rg][v], W[4 * v + (B_ >> 2)][B_ & 3], acc[rg]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                SLICE_BWD_B(0) SLICE_BWD_B(1) SLICE_BWD_B(2) SLICE_BWD_B(3) SLICE_BWD_B(4) SLICE_BWD_B(5) SLICE_BWD_B(6) SLICE_BWD_B(7)
                SLICE_BWD_B(8) SLICE_BWD_B(9) SLICE_BWD_B(10) SLICE_BWD_B(11) SLICE_BWD_B(12) SLICE_BWD_B(13) SLICE_BWD_B(14) SLICE_BWD_B(15)
            }
#undef SLICE_BWD_B
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // MFMA result -> VMEM store data: wait states by hand (inline-asm stores)
            f32x4* out = inbox + (size_t)(t & 1) * slot_w;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) store_l2(out + out_ofs + rg * 16, acc[rg]);
        }
    }
    if (act) a.dc[hi] = dcv;
}

// Kh [256][1024] (packed gate columns) -> the register images of the slice kernels:
//   fwd word b (< 16), component e of (cu, w), lane l:  Kh[64w + 16(b/4) + 4(b%4) + e][64cu + l]
//   bwd word i, component e' of (cu, w), lane l, k = 4i + e':  Kh[64w + l][64cu + k]
__device__ __forceinline__ void repack_kh_slice_body(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int b, int nb) {
    const int total = SCU * 4 * 16 * 64;            // f32x4 words per copy
    for (int idx = b * blockDim.x + threadIdx.x; idx < total; idx += nb * blockDim.x) {
        const int l = idx & 63, i = (idx >> 6) & 15, w = (idx >> 10) & 3, cu = idx >> 12;
        {
            const int k0 = 64 * w + 16 * (i >> 2) + 4 * (i & 3);
            const float* src = Kh + (size_t)k0 * SG4 + 64 * cu + l;
            float4 v;
            v.x = src[0]; v.y = src[SG4]; v.z = src[2 * (size_t)SG4]; v.w = src[3 * (size_t)SG4];
            reinterpret_cast<float4*>(fwd)[idx] = v;
        }
        reinterpret_cast<float4*>(bwd)[idx] = *reinterpret_cast<const float4*>(Kh + (size_t)(64 * w + l) * SG4 + 64 * cu + 4 * i);
    }
}
__global__ void k_repack_kh_slice(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    repack_kh_slice_body(Kh, fwd, bwd, blockIdx.x, gridDim.x);
}

// ================================================================ hidden size 512 on the bf16 matrix pipe (round 3)
// The recurrent product of the kernels above runs at the fp32 MFMA rate and pays for 4-row groups: a 6-row slice (B = 45) costs
// 2 x 128 `4x4x1` MFMAs = 2048 cycles per step and wave, 13 rows (B = 100) cost 4096.  The dense GEMMs of the step already
// compute fp32 products from an exact three-way bf16 split on the bf16 pipe (gemm.hip: six products per term, 16x the fp32
// rate per instruction); the same arithmetic here: K_h sits in the registers as THREE bf16 planes (192 VGPRs per lane instead
// of 128), h_t travels as three bf16 planes (split once, by the cell thread that produced it), and a step's product is
// 96 `v_mfma_f32_16x16x32_bf16` per wave = 1536 cycles for ANY number of rows up to 16 per XCD.
//   forward : wave w = K range 128 w ... (four k steps of 32) x the CU's 64 gate columns (four 16-column tiles, tile nt = unit
//             block nt: column 16 nt + 4 g + e).  A operand = rows of h: lane (row = l % 16, k group = l / 16) holds 8 bf16 of
//             one plane; only the lanes of real rows load anything, the others stay zero.
//   HX16    : [T+1][8 xcd][4 w][3 planes][4 k steps][4 RG rows][4 k groups] 16-byte words = 8 units of one row and plane: the
//             rows a load instruction fetches are contiguous (6 rows = 384 bytes; with the row outermost every poll round
//             was 6 scattered 64-byte pieces per instruction, 9000 requests per XCD, and the forward step 3.1 us).
//             Eight cell lanes pack their units' bf16 with DPP moves and one of them stores the word; the consumers' readiness
//             test is per 16-bit half (fill 0xFFFF is a bf16 NaN, no h or residual is).
//   backward: dz slice (rows x the CU's 64 gate columns) as A, split by the cell threads into LDS; wave w holds K_h^T for
//             destination units 128 w ... (eight 16-unit tiles = eight destination CUs) x 64 columns (two k steps); the D
//             registers of tile nt ARE the inbox words of destination 8 w + nt (lane = (row group, unit), 4 rows per lane).
// Term order and split are those of gemm.hip (smallest products first, fp32 accumulation).
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 xbf16x8;
typedef unsigned short xu16x2 __attribute__((ext_vector_type(2)));
constexpr int HXW16 = 4 * 3 * 4 * 4;          // 16-byte words per row of one XCD and time index

__device__ __forceinline__ unsigned bf16_rne(float x) {          // low half = bf16(x), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void split3(float x, unsigned (&p)[3]) {
    p[0] = bf16_rne(x);
    const float r1 = x - __uint_as_float(p[0] << 16);
    p[1] = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(p[1] << 16);
    p[2] = bf16_rne(r2);
}
template <int OFS>
__device__ __forceinline__ f32x4 load_sc1_ofs(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=&v"(v) : "v"(p), "n"(OFS) : "memory");
    return v;
}
template <int OFS>
__device__ __forceinline__ void store_l2_ofs(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2\n\ts_nop 1" : : "v"(p), "v"(v), "n"(OFS) : "memory");
}
// Eight consecutive lanes (units 8 h ... 8 h + 7 of one row: cell lane = 16 i + 4 bb + e) each hold one bf16 in the low half
// of `p`; the lane with bb even, e = 0 gets the 16-byte hand-off word of the eight.  (Three 2-byte stores per lane instead cost
// 860 cycles per step in the store queue and delayed the consumers by as much, profiles/r03_xcd16_probe1.log.)
__device__ __forceinline__ f32x4 pack8_bf16(unsigned p) {
    const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0xB1, 0xF, 0xF, true);             // lane ^ 1
    const unsigned d = __builtin_amdgcn_perm(nb, p, 0x05040100u);                                          // {own, neighbour}
    const unsigned q1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0xAA, 0xF, 0xF, true);            // lane 2 of the quad
    const unsigned n0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x104, 0xF, 0xF, true);           // row_shl:4 = next quad
    const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q1, 0x104, 0xF, 0xF, true);
    return f32x4{__uint_as_float(d), __uint_as_float(q1), __uint_as_float(n0), __uint_as_float(n1)};
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// no 16-bit half of the twelve fragments shows the fill pattern
__device__ __forceinline__ bool frags16_ready(const f32x4 (&av)[12]) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = pk_max_u16(m, __float_as_uint(av[j][e]));
    return (m & 0xffffu) != 0xffffu && (m >> 16) != 0xffffu;
}
__device__ __forceinline__ xbf16x8 as_bf16x8(const f32x4& v) { return *reinterpret_cast<const xbf16x8*>(&v); }

#define XCD_STAMP(i) if (PROF) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; }
#define X16_TERMS(DO) DO(2, 0) DO(0, 2) DO(1, 1) DO(1, 0) DO(0, 1) DO(0, 0)
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_xcd16(const LstmFwdXcdArgs a) {
    constexpr int HXR = 4 * RG;                  // rows per XCD the hand-off buffer is laid out for
    __shared__ __attribute__((aligned(16))) float red[2][4][16 * 16 * 4];   // [step parity][wave][row][unit][gate]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;
    const int rgc = min((rpx + 3) >> 2, RG);     // waves with cell threads: wave rg owns rows 4 rg ... 4 rg + 3

    xbf16x8 W[3][4][4];                          // [plane][k step][column tile]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * 48) * 64 + lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 wv = wp[((pl * 4 + ks) * 4 + nt) * 64];
                    asm volatile("" : "+a"(wv));          // the weights live in the accumulation half of the register file: the
                    W[pl][ks][nt] = as_bf16x8(wv);        // arch half is for the fragments in flight, the cell update and the addresses
                }
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < rgc;
    const bool act = cellw && lrow < rpx && row < B;
    const bool pub = cellw && lrow < rpx;         // rows past B on the last XCD publish zeros: the readers load every row < rpx
    // overlapped step: the projection GEMM on the other XCDs draws row tiles as the time steps complete.  progress[t] counts the
    // blocks whose row-major h of step t (Hs index t + 1) has reached memory: the stores are write-through, every wave drains its
    // vector-memory queue in the poll of a later step, and the block barrier of that step orders all four waves before thread 0
    // publishes -- `lag` steps behind (1, or 2 when the outputs are deferred behind the next poll); the tail is published at the end
    const bool wt = a.progress != nullptr;
    const bool pubs = wt && !(a.progress_lag & 256);                    // (diagnostic bit 256: write-through stores, nothing published)
    const int lag = ((a.variant & XCD_DEFER_OUTPUTS) ? 2 : 1) + (a.progress_lag & 7);
    const int pk = a.progress_every > 0 ? a.progress_every : 1;         // a step is published when it is the last of a group of pk (or the pass's last)
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * XH + unit] : 0.0f;
    const int arow = lane & 15, akg = lane >> 4;
    const bool ldl = arow < rpx;                  // lanes of pad rows never load: their A registers stay zero
    const size_t hx_step = (size_t)HXR * NXCD * HXW16;                      // 16-byte words per time index
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)xcd * 4 + wave) * 12) * (HXR * 4) + arow * 4 + akg;
    // XCD_STREAM: every lane loads (the lanes of pad rows row 0 again -- same cache lines, and no divergent loads for the compiler to merge)
    const f32x4* hx_in_all = hx_in - (ldl ? 0 : arow * 4);
    // unit u = 16 cu + 4 cbb + ce -> w = u / 128, k step = u % 128 / 32, k group = u % 32 / 8, position u % 8
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) +
        ((((size_t)xcd * 4 + (cu >> 3)) * 12 + ((cu & 7) >> 1)) * (HXR * 4) + lrow * 4 + 2 * (cu & 1) + (cbb >> 1));
    const bool stl = pub && ce == 0 && (cbb & 1) == 0;      // the lane that stores the eight units 8 (cbb / 2) ... of its row
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    float zq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float o_c = 0.f, o_hh = 0.f, o_g[4] = {0.f, 0.f, 0.f, 0.f};
    bool o_have = false;
    if (act) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.t0 + k < a.t1) {
                const float* zn = a.Z + ((size_t)(a.t0 + k) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
#pragma unroll
                for (int g = 0; g < 4; ++g) zq[k][g] = zn[4 * g];
            }
    }
    f32x4 av[12];                                 // [plane][k step]
#pragma unroll
    for (int j = 0; j < 12; ++j) av[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = a.t0; t < a.t1; ++t) {
        XCD_STAMP(4)
        // XCD_PROBE / XCD_STREAM as in k_lstm_fwd_pair16 (lstm_pair16.h, where both were measured first): a wave polls ONE word per lane -- plane 2
        // (stored last) of k step (row % 4): its lanes cover every (producer CU, cell wave) of the wave's K range -- and behind a successful
        // probe the twelve fragments are ordinary loads in k-step order that the MFMAs consume as they arrive; checked afterwards, a miss
        // redoes the step behind the sc1 poll
        bool stream = false;
        if (a.variant & XCD_PROBE) {
            const f32x4* sp = hx_in + (size_t)t * hx_step + (2 * 4 + (arow & 3)) * (HXR * 4);
            for (int spins = 0; spins < a.spin_limit; ++spins) {
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ldl) { sv = load_sc1(sp); drain_vmem(); }
                asm volatile("" : "+v"(sv));
                const unsigned m = pk_max_u16(pk_max_u16(__float_as_uint(sv[0]), __float_as_uint(sv[1])), pk_max_u16(__float_as_uint(sv[2]), __float_as_uint(sv[3])));
                if (__all((m & 0xffffu) != 0xffffu && (m >> 16) != 0xffffu)) { stream = (a.variant & XCD_STREAM) != 0; break; }
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) break;
            }
        }
        float zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { zin[g] = zq[0][g]; zq[0][g] = zq[1][g]; zq[1][g] = 0.0f; }
        float* zp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        const bool outs_early = !(a.variant & XCD_STREAM);
        f32x4 acc[4];
        for (;;) {
        if (stream) {
            const f32x4* af = hx_in_all + (size_t)t * hx_step;
#define X16_LDC(J) av[J] = af[((J) >> 2) * (4 * HXR * 4) + ((J) & 3) * (HXR * 4)];
#define X16_LD3(KS) X16_LDC(KS) X16_LDC(4 + KS) X16_LDC(8 + KS) __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_sched_barrier(0);
            X16_LD3(0) X16_LD3(1) X16_LD3(2) X16_LD3(3)
#undef X16_LD3
#undef X16_LDC
        } else {
            bool fail = false;
            const f32x4* af = hx_in + (size_t)t * hx_step;
            for (int spins = 0;; ++spins) {
                if (ldl) {
#define X16_LD(J) av[J] = load_sc1_ofs<((J) & 3) * HXR * 64>(af + ((J) >> 2) * (4 * HXR * 4));      /* (plane, k step) stride: HXR x 64 bytes */
                    X16_LD(0) X16_LD(1) X16_LD(2) X16_LD(3) X16_LD(4) X16_LD(5) X16_LD(6) X16_LD(7) X16_LD(8) X16_LD(9) X16_LD(10) X16_LD(11)
#undef X16_LD
                    drain_vmem();
                }
#pragma unroll
                for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(av[j]));
                const bool ok = !ldl || frags16_ready(av);
                if (__all(ok)) break;
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        // outputs of the step before: behind a successful poll, under the MFMAs -- streamed fetch: behind the MFMAs, whose waits then count the
        // fragments only
        if (outs_early && (a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
            a.Cs[((size_t)t * B + row) * XH + unit] = o_c;
            store_h_row(a.Hs + ((size_t)t * B + row) * XH + unit, o_hh, wt);
            float* zo = a.Z + ((size_t)(t - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
            o_have = false;
        }
        if (outs_early && act && t + 2 < a.t1) {
            const float* zn = zp + 2 * (size_t)B * XG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        XCD_STAMP(0)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#define X16_FWD(PA, PB)                                                                                         \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                    \
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[(PA) * 4 + ks]), W[PB][ks][nt], acc[nt], 0, 0, 0);
            X16_TERMS(X16_FWD)
#undef X16_FWD
        }
        if (!outs_early) {
            if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
                a.Cs[((size_t)t * B + row) * XH + unit] = o_c;
                store_h_row(a.Hs + ((size_t)t * B + row) * XH + unit, o_hh, wt);
                float* zo = a.Z + ((size_t)(t - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
                zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
            }
            o_have = false;
        }
        if (!stream) break;
        if (__all(frags16_ready(av))) break;
        stream = false;                                // a fragment was not there yet: redo the step behind the sc1 poll
        }
        if (!outs_early && act && t + 2 < a.t1) {
            const float* zn = zp + 2 * (size_t)B * XG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        if (PROF) { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); acc[0][0] += 0.0f; }
        XCD_STAMP(1)
        if (akg < rgc) {                           // D: lane = (column 4 g + e of the tile, rows 4 akg ... + 3)
            float* rp = &red[t & 1][wave][0] + ((4 * akg) * 16 + (lane & 3)) * 4 + ((lane >> 2) & 3);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rp[(r * 16 + 4 * nt) * 4] = acc[nt][r];
        }
        __syncthreads();
        if (s_fail) return;
        if (pubs && tid == 0 && t - lag >= a.t0 && (t - lag) % pk == pk - 1)
            __hip_atomic_fetch_add(a.progress + (t - lag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        XCD_STAMP(2)

        if (cellw) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[t & 1][0][0]) + lrow * 16 + 4 * cbb + ce;
                const f32x4 r0 = rsrc[0], r1 = rsrc[256], r2 = rsrc[512], r3 = rsrc[768];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            {
                unsigned hp[3];
                split3(hn, hp);                    // (every lane of a cell wave: the packing below crosses lanes)
                const f32x4 w0 = pack8_bf16(hp[0]), w1 = pack8_bf16(hp[1]), w2 = pack8_bf16(hp[2]);
                if (stl) {
                    f32x4* o = hx_out + (size_t)(t + 1) * hx_step;
                    store_l2_ofs<0>(o, w0); store_l2_ofs<0>(o + 4 * HXR * 4, w1); store_l2_ofs<0>(o + 8 * HXR * 4, w2);
                }
            }
            XCD_STAMP(3)
            if (a.variant & XCD_DEFER_OUTPUTS) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * XH + unit] = cp;
                store_h_row(a.Hs + ((size_t)(t + 1) * B + row) * XH + unit, hn, wt);
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;
            }
        }
    }
    if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
        a.Cs[((size_t)a.t1 * B + row) * XH + unit] = o_c;
        store_h_row(a.Hs + ((size_t)a.t1 * B + row) * XH + unit, o_hh, wt);
        float* zo = a.Z + ((size_t)(a.t1 - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
    }
    if (wt) {                                     // the last `lag` steps: drain, barrier, publish
        drain_vmem();
        __syncthreads();
        if (tid == 0 && pubs)
            for (int tp = max(a.t0, a.t1 - lag); tp < a.t1; ++tp)
                if (tp % pk == pk - 1 || tp == a.t1 - 1) __hip_atomic_fetch_add(a.progress + tp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PROF && lane == 0 && a.prof) {
        XCD_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}

// inbox as in k_lstm_bwd_xcd: [2 slots][8 xcd][32 dest][32 producer][RG][16 units][4 rows]
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_xcd16(const LstmBwdXcdArgs a) {
    constexpr int NG = 4 / RG;
    constexpr int LPW = 2 * RG;
    __shared__ __attribute__((aligned(16))) float psum[4 * 64 * 4];
    __shared__ __attribute__((aligned(16))) unsigned char dzA[3][2][64][16];      // [plane][k step][lane = (k group, row)][8 bf16]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    for (int i = tid; i < 3 * 2 * 64 * 4; i += 256) reinterpret_cast<unsigned*>(&dzA[0][0][0][0])[i] = 0u;     // rows >= 4 RG stay zero
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;

    xbf16x8 W[3][2][8];                           // [plane][k step][destination tile]: Kh[128 w + 16 nt + l % 16][64 cu + 32 ks + 8 (l / 16) + j]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * 48) * 64 + lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) {
                    f32x4 wv = wp[((pl * 2 + ks) * 8 + nt) * 64];
                    asm volatile("" : "+a"(wv));
                    W[pl][ks][nt] = as_bf16x8(wv);
                }
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const size_t hi = (size_t)row * XH + unit;
    float dcv = act ? a.dc[hi] : 0.0f;
    const size_t slot_w = (size_t)NXCD * NCU * NCU * RG * 16;
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    const size_t in_base = (((size_t)xcd * NCU + cu) * NCU + 8 * wave) * RG * 16 + lane;
    // producer side: tile nt of wave w -> destination 8 w + nt, word (dest, producer = cu, row group l / 16, unit l % 16)
    const size_t out_ofs = ((((size_t)xcd * NCU + 8 * wave) * NCU + cu) * RG + (lane >> 4)) * 16 + (lane & 15);
    const bool outl = (lane >> 4) < RG;
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    // dz of gate g, local column k = 16 cbb + 4 g + ce: k step cbb / 2, k group 2 (cbb % 2) + g / 2, position 4 (g % 2) + ce
    unsigned short* const dz_out = reinterpret_cast<unsigned short*>(&dzA[0][cbb >> 1][(2 * (cbb & 1)) * 16 + lrow][0]) + ce;

    float n_si = 0.f, n_tj = 0.f, n_sf = 0.f, n_so = 0.f, n_ct = 0.f, n_cp = 0.f, n_dh = 0.f;
    if (act && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        const float* gp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        n_si = gp[0]; n_tj = gp[4]; n_sf = gp[8]; n_so = gp[12];
        n_ct = a.Cs[(size_t)(t + 1) * B * XH + hi]; n_cp = a.Cs[(size_t)t * B * XH + hi];
        n_dh = a.dH[(size_t)t * B * XH + hi];
    }

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        XCD_STAMP(4)
        const float si = n_si, tj = n_tj, sf = n_sf, so = n_so, ct = n_ct, cpv = n_cp, dht = n_dh;
        float* gp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
        // ---- A: consume
        f32x4 wsum = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base;
            f32x4 v[LPW];
            bool fail = false;
            for (int spins = 0;; ++spins) {
#pragma unroll
                for (int k = 0; k < LPW; ++k) v[k] = load_sc1(in + k * 64);
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int k = 0; k < LPW; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                if (__all(ok)) break;
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
#pragma unroll
            for (int k = 0; k < LPW; ++k) {
                store_l2(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        }
        XCD_STAMP(0)
        *reinterpret_cast<f32x4*>(&psum[(wave * 64 + lane) * 4]) = wsum;
        __syncthreads();
        if (s_fail) return;
        XCD_STAMP(1)

        // ---- B: gate gradients
        float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
        if (cellw) {
            if (act) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp)
                        dh_rec += psum[(w * 64 + (grp * RG + wave) * 16 + 4 * cbb + ce) * 4 + ci];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cpv, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                if (!(a.variant & XCD_DEFER_OUTPUTS)) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }
                dcv = cg.dc_out;
            }
            const float dzv[4] = {di, dj, df, dg};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned p3[3];
                split3(dzv[g], p3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)        // plane stride 2 * 64 * 8 halves, k group stride 16 * 8
                    dz_out[pl * (2 * 64 * 8) + (g >> 1) * (16 * 8) + 4 * (g & 1)] = (unsigned short)p3[pl];
            }
        }
        __syncthreads();
        XCD_STAMP(2)
        drain_vmem();
        if ((a.variant & XCD_DEFER_OUTPUTS) && act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }
        if (act && t > a.t0) {
            const float* gn = a.Z + ((size_t)(t - 1) * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            n_si = gn[0]; n_tj = gn[4]; n_sf = gn[8]; n_so = gn[12];
            n_ct = cpv; n_cp = a.Cs[(size_t)(t - 1) * B * XH + hi];
            n_dh = a.dH[(size_t)(t - 1) * B * XH + hi];
        }

        // ---- C: produce the partials of dh_{t-1}
        if (t > 0) {
            xbf16x8 av[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) av[pl][ks] = *reinterpret_cast<const xbf16x8*>(&dzA[pl][ks][lane][0]);
            if (a.variant & XCD_GROUP_STORES) {
                // the eight destination tiles in two groups of four: the first group's partials leave (and reach their consumers) while the
                // second multiplies (k_lstm_bwd_pair16, where the grouping was measured first: 6.76 -> 5.85 us per step)
                f32x4* out = inbox + (size_t)(t & 1) * slot_w + out_ofs;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    f32x4 acc[4];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) acc[j4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
#define X16_BWD(PA, PB)                                                                                         \
                        _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                        \
                            acc[j4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA][ks], W[PB][ks][4 * g2 + j4], acc[j4], 0, 0, 0);
                        X16_TERMS(X16_BWD)
#undef X16_BWD
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    if (g2 == 1) { XCD_STAMP(3) }
                    if (outl) {
#pragma unroll
                        for (int j4 = 0; j4 < 4; ++j4) store_l2(out + (size_t)(4 * g2 + j4) * NCU * RG * 16, acc[j4]);
                    }
                }
            } else {
            f32x4 acc[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#define X16_BWD(PA, PB)                                                                                         \
                _Pragma("unroll") for (int nt = 0; nt < 8; ++nt)                                                \
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA][ks], W[PB][ks][nt], acc[nt], 0, 0, 0);
                X16_TERMS(X16_BWD)
#undef X16_BWD
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            XCD_STAMP(3)
            if (outl) {
                f32x4* out = inbox + (size_t)(t & 1) * slot_w + out_ofs;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) store_l2(out + (size_t)nt * NCU * RG * 16, acc[nt]);
            }
            }
        }
    }
    if (act) a.dc[hi] = dcv;
    if (PROF && lane == 0 && a.prof) {
        XCD_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}
#undef X16_TERMS
#undef XCD_STAMP

// Kh [512][2048] (packed gate columns) -> the three-plane bf16 register images of the two kernels above, 16-byte words
// [cu][w][48][64 lanes]:
//   fwd word (pl * 4 + ks) * 4 + nt, lane l, position j: plane pl of Kh[128 w + 32 ks + 8 (l / 16) + j][64 cu + 16 nt + l % 16]
//   bwd word (pl * 2 + ks) * 8 + nt, lane l, position j: plane pl of Kh[128 w + 16 nt + l % 16][64 cu + 32 ks + 8 (l / 16) + j]
__device__ __forceinline__ void repack_kh_xcd16_body(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int b, int nb) {
    const int total = NCU * 4 * 16 * 64;            // (cu, w, 16 operand slots, lane) per image
    uint4* const fo = reinterpret_cast<uint4*>(fwd);
    uint4* const bo = reinterpret_cast<uint4*>(bwd);
    for (int idx = b * blockDim.x + threadIdx.x; idx < total; idx += nb * blockDim.x) {
        const int l = idx & 63, slot = (idx >> 6) & 15, w = (idx >> 10) & 3, cu = idx >> 12;
        float x[8];
        unsigned pk[3][4];
        {
            const int ks = slot >> 2, nt = slot & 3;
            const float* src = Kh + (size_t)(128 * w + 32 * ks + 8 * (l >> 4)) * XG4 + 64 * cu + 16 * nt + (l & 15);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = src[(size_t)j * XG4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned p0[3], p1[3];
                split3(x[j], p0); split3(x[j + 1], p1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk[pl][j >> 1] = (p0[pl] & 0xffffu) | (p1[pl] << 16);
            }
            const size_t base = ((size_t)(cu * 4 + w) * 48) * 64 + l;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fo[base + (size_t)((pl * 4 + ks) * 4 + nt) * 64] = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
        }
        {
            const int ks = slot >> 3, nt = slot & 7;
            const float* src = Kh + (size_t)(128 * w + 16 * nt + (l & 15)) * XG4 + 64 * cu + 32 * ks + 8 * (l >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = src[j];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned p0[3], p1[3];
                split3(x[j], p0); split3(x[j + 1], p1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk[pl][j >> 1] = (p0[pl] & 0xffffu) | (p1[pl] << 16);
            }
            const size_t base = ((size_t)(cu * 4 + w) * 48) * 64 + l;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bo[base + (size_t)((pl * 2 + ks) * 8 + nt) * 64] = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
        }
    }
}
__global__ void k_repack_kh_xcd16(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    repack_kh_xcd16_body(Kh, fwd, bwd, blockIdx.x, gridDim.x);
}

// ================================================================ XCD-PAIR-local recurrence, hidden size 1024 (cfg-C / cfg-E)
// K_h of a 1024-unit layer is 16 MiB in fp32 -- exactly the register file of one XCD, so a copy cannot live in one XCD beside
// anything else.  It lives in the registers of an XCD PAIR instead: 64 CUs x 4 waves x 256 weight VGPRs.  The chip holds four
// copies, the batch is split four ways (ceil(B / 4) rows per pair: 12 at B = 45 = THREE full row groups, no padding), and h_t /
// the dh partials are handed around inside a pair only -- 2 of the 8 XCDs take part in a step's exchange instead of all 8
// as in the column-split kernels of lstm_step.hip (round 1: 7.0 / 9.2 us per step at cfg-C).
// Differences from the one-XCD kernels above: a hand-off crosses an XCD boundary for half of its producers, so every hand-off
// store is write-through (`sc1`: the line leaves the writer's L2) and every hand-off load an `sc1` load; a wave's K slice is 256
// wide (four fragments per row group, 256 MFMAs per row group and step); the BPTT inbox is laid out [dest][row group]
// [producer][unit], so that a lane's words of one row group differ in the producer only and three row groups need no padding.
// Measured (profiles/r03_pair_probe1.log, B = 45, T = 50): forward 6.38 us per step (37.6 % of the fp32-MFMA peak) against 7.0 for
// the column-split kernel, backward 9.26 against 9.2 -- the MFMA floor is 2.8 us, the rest is the cross-XCD hand-off, which costs
// the same three microseconds whether 2 or 8 XCDs take part.  A variant with separate flag words (producer: tile -> wait for
// the write-through acknowledgement -> one dword per consumer wave; consumer: poll 64 bytes, fetch the 12 KB once, no data
// resets) was built too and is slower (9.68 / 9.58, profiles/r03_pair_probe2.log): three serialised fabric round trips instead
// of one.  So these kernels are correct, tested and NOT the default at hidden size 1024 (fsmg_config.recurrence =
// FSMG_RECURRENCE_XCD_LOCAL selects them); what would pay here is several independent row-group chains per pair (the
// hand-off is longer than a chain's MFMA phase) -- see DESIGN.md section 4.
constexpr int PH = 1024, PG4 = 4 * PH;
constexpr int PNX = 2, PGRP = NXCD / PNX;       // XCDs per weight copy; copies (= row splits) on the chip
constexpr int PCU = NCU * PNX;                  // CUs per copy
constexpr int PKW = PH / 4;                     // K range of a wave
constexpr int PNQ = PKW / 64;                   // hand-off fragments per wave and row group
constexpr int PNW = PKW / 4;                    // 16-byte weight words per lane (64 -> 256 VGPRs)

// HX  [T+1][4 pairs][4 w][RG][4 q][64 lanes][4], KhX [64 cu][4 w][64][64 lanes][4] (k_repack_kh_pair)
template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_pair(const LstmFwdXcdArgs a) {
    constexpr int NF = PNQ * RG;
    __shared__ __attribute__((aligned(16))) float red[2][RG][4][64 * 4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int grp = role.xcd / PNX, cu = (role.xcd % PNX) * NCU + role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + PGRP - 1) / PGRP, row0 = grp * rpx;
    if (row0 >= B) return;

    f32x4 W[PNW];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * PNW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < PNW; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * PH + unit] : 0.0f;
    const size_t hx_step = (size_t)PGRP * 4 * RG * PNQ * 64;
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)grp * 4 + wave) * RG) * PNQ * 64 + lane;
    // CU c produces k = 16 c .. 16 c + 15: wave c / 16, fragment (c / 4) % 4, lanes 16 (c % 4) .. + 15 of it
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) + ((((size_t)grp * 4 + (cu >> 4)) * RG + wave) * PNQ + ((cu >> 2) & 3)) * 64 +
                    16 * (cu & 3) + 4 * cbb + ci;
    const int wofs = ((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3);
    float zq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float o_c = 0.f, o_hh = 0.f, o_g[4] = {0.f, 0.f, 0.f, 0.f};
    bool o_have = false;
    if (act) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (a.t0 + k < a.t1) {
                const float* zn = a.Z + ((size_t)(a.t0 + k) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
#pragma unroll
                for (int g = 0; g < 4; ++g) zq[k][g] = zn[4 * g];
            }
    }

    for (int t = a.t0; t < a.t1; ++t) {
        f32x4 av[NF];
        {
            const bool fail = !wait_all_fragments<NF>(hx_in + (size_t)t * hx_step, av, a.spin_limit, a.err_flag, (a.variant & XCD_NO_POLL_SLEEP) != 0);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {          // outputs of the step before: behind a successful poll, under the MFMAs
            a.Cs[((size_t)t * B + row) * PH + unit] = o_c;
            a.Hs[((size_t)t * B + row) * PH + unit] = o_hh;
            float* zo = a.Z + ((size_t)(t - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
            zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
        }
        float zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { zin[g] = zq[0][g]; zq[0][g] = zq[1][g]; zq[1][g] = 0.0f; }
        float* zp = a.Z + ((size_t)t * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        if (act && t + 2 < a.t1) {
            const float* zn = zp + 2 * (size_t)B * PG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        f32x4 acc[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#define PAIR_FWD_B(B_)                                                                   \
        _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) { XCD_MFMA_B(B_, av[PNQ * rg + q], W[16 * q + B_], acc[rg]) }
#pragma unroll
        for (int q = 0; q < PNQ; ++q) {
            PAIR_FWD_B(0) PAIR_FWD_B(1) PAIR_FWD_B(2) PAIR_FWD_B(3) PAIR_FWD_B(4) PAIR_FWD_B(5) PAIR_FWD_B(6) PAIR_FWD_B(7)
            PAIR_FWD_B(8) PAIR_FWD_B(9) PAIR_FWD_B(10) PAIR_FWD_B(11) PAIR_FWD_B(12) PAIR_FWD_B(13) PAIR_FWD_B(14) PAIR_FWD_B(15)
        }
#undef PAIR_FWD_B
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            float* rp = &red[t & 1][rg][wave][0] + wofs;
#pragma unroll
            for (int i = 0; i < 4; ++i) rp[64 * i] = acc[rg][i];
        }
        __syncthreads();
        if (s_fail) return;

        if (cellw) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[t & 1][wave][0][0]) + lane;
                const f32x4 r0 = rsrc[0], r1 = rsrc[64], r2 = rsrc[128], r3 = rsrc[192];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            if (ce == 0) store_sc1(hx_out + (size_t)(t + 1) * hx_step, hv);       // write-through: half of the consumers sit on the other XCD
            if (a.variant & XCD_DEFER_OUTPUTS) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * PH + unit] = cp;
                a.Hs[((size_t)(t + 1) * B + row) * PH + unit] = hn;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;
            }
        }
    }
    if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
        a.Cs[((size_t)a.t1 * B + row) * PH + unit] = o_c;
        a.Hs[((size_t)a.t1 * B + row) * PH + unit] = o_hh;
        float* zo = a.Z + ((size_t)(a.t1 - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
    }
}

// ---------------------------------------------------------------- row-group CHAINS on an XCD pair (hidden size 1024)
// The pair kernels above spend ~3 us of a 6.4 / 9.3 us step in the cross-XCD hand-off while the MFMAs of one row group take
// ~1 us: the regime in which independent chains pay (at hidden size 512 they did not: header of this file).  The RG row groups
// of a pair are independent sequences, so they become RG chains that the same four waves advance round-robin: phase (c, t) =
// fragments of row group c at time t -> 256 MFMAs -> K-split partials through LDS -> cell update by wave c -> hand-off store.
// While h_{t+1} of row group c crosses the fabric, the block runs the phases of the other row groups; the poll of a phase is
// ISSUED at the end of the previous phase's MFMA stream and only CHECKED when the phase starts.
// Memory-queue discipline (a CU returns vector-memory operations in order, and vmcnt counts loads and stores alike):
//   * a wait is `s_waitcnt vmcnt(N)` with N = the number of operations this wave issued AFTER the one it needs, so that a poll
//     never waits for the slower loads / stores behind it;
//   * a cell wave's outputs (c, h, gates / dz) and next inputs are issued one phase after its update, right behind that phase's
//     poll; all of them unconditionally (inactive lanes: exec-masked stores, clamped load addresses), so the counts depend on
//     the wave's role only; around the ends of a pass, where an operation is missing, waits fall back to vmcnt(0) (`cons`);
//   * a load that is in flight across phases must not target a register the compiler manages (hipcc copies / re-uses the
//     destination of an inline-asm load at will): such loads land in FIXED accumulation registers a[176:245], far above what
//     the compiler allocates in these kernels, and an empty "adopt" asm hands them over after the wait.
//     tools/check_xcd_asm.py verifies both properties on the final ISA.
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
#define CHAIN_LOAD_X4(REG, PTR) do { f32x4 d_; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "={" REG "}"(d_) : "v"(PTR) : "memory"); } while (0)
#define CHAIN_LOAD_DW(REG, PTR, OFS) do { float d_; asm volatile("global_load_dword %0, %1, off offset:" #OFS : "={" REG "}"(d_) : "v"(PTR) : "memory"); } while (0)
#define CHAIN_ADOPT(REG, VAR) asm volatile("" : "={" REG "}"(VAR))

// the four hand-off words of chain C: a[176 + 16 C : + 15]
template <int C>
__device__ __forceinline__ void chain_issue_fragments(const f32x4* p0, const f32x4* p1, const f32x4* p2, const f32x4* p3) {
    static_assert(C >= 0 && C < 4, "chain");
    if constexpr (C == 0) { CHAIN_LOAD_X4("a[176:179]", p0); CHAIN_LOAD_X4("a[180:183]", p1); CHAIN_LOAD_X4("a[184:187]", p2); CHAIN_LOAD_X4("a[188:191]", p3); }
    if constexpr (C == 1) { CHAIN_LOAD_X4("a[192:195]", p0); CHAIN_LOAD_X4("a[196:199]", p1); CHAIN_LOAD_X4("a[200:203]", p2); CHAIN_LOAD_X4("a[204:207]", p3); }
    if constexpr (C == 2) { CHAIN_LOAD_X4("a[208:211]", p0); CHAIN_LOAD_X4("a[212:215]", p1); CHAIN_LOAD_X4("a[216:219]", p2); CHAIN_LOAD_X4("a[220:223]", p3); }
    if constexpr (C == 3) { CHAIN_LOAD_X4("a[224:227]", p0); CHAIN_LOAD_X4("a[228:231]", p1); CHAIN_LOAD_X4("a[232:235]", p2); CHAIN_LOAD_X4("a[236:239]", p3); }
}
template <int C>
__device__ __forceinline__ void chain_adopt_fragments(f32x4 (&v)[4]) {
    if constexpr (C == 0) { CHAIN_ADOPT("a[176:179]", v[0]); CHAIN_ADOPT("a[180:183]", v[1]); CHAIN_ADOPT("a[184:187]", v[2]); CHAIN_ADOPT("a[188:191]", v[3]); }
    if constexpr (C == 1) { CHAIN_ADOPT("a[192:195]", v[0]); CHAIN_ADOPT("a[196:199]", v[1]); CHAIN_ADOPT("a[200:203]", v[2]); CHAIN_ADOPT("a[204:207]", v[3]); }
    if constexpr (C == 2) { CHAIN_ADOPT("a[208:211]", v[0]); CHAIN_ADOPT("a[212:215]", v[1]); CHAIN_ADOPT("a[216:219]", v[2]); CHAIN_ADOPT("a[220:223]", v[3]); }
    if constexpr (C == 3) { CHAIN_ADOPT("a[224:227]", v[0]); CHAIN_ADOPT("a[228:231]", v[1]); CHAIN_ADOPT("a[232:235]", v[2]); CHAIN_ADOPT("a[236:239]", v[3]); }
}
__device__ __forceinline__ bool chain_fragments_ready(f32x4 (&v)[4]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(v[j])); ok &= frag_ready(v[j]); }
    return __all(ok);
}
// a cell wave's own inputs (every wave is the cell wave of at most one chain): four dwords 16 bytes apart -> a[240:243], two
// more scalars -> a244, a245
__device__ __forceinline__ void chain_issue_dw4(const float* p) {
    CHAIN_LOAD_DW("a240", p, 0); CHAIN_LOAD_DW("a241", p, 16); CHAIN_LOAD_DW("a242", p, 32); CHAIN_LOAD_DW("a243", p, 48);
}
__device__ __forceinline__ void chain_adopt_dw4(float (&z)[4]) {
    CHAIN_ADOPT("a240", z[0]); CHAIN_ADOPT("a241", z[1]); CHAIN_ADOPT("a242", z[2]); CHAIN_ADOPT("a243", z[3]);
}
__device__ __forceinline__ void chain_issue_dw_a(const float* p) { CHAIN_LOAD_DW("a244", p, 0); }
__device__ __forceinline__ void chain_issue_dw_b(const float* p) { CHAIN_LOAD_DW("a245", p, 0); }
__device__ __forceinline__ void chain_adopt_dw2(float& x, float& y) { CHAIN_ADOPT("a244", x); CHAIN_ADOPT("a245", y); }
// a wave-uniform 64-bit value as an SGPR pair (the "s" constraint of an inline asm does not move a VGPR-resident value itself)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// exec-masked stores that are issued whatever the mask (an instruction with exec = 0 still takes its slot in the vmcnt order)
__device__ __forceinline__ void store_dw4_stride16_masked(unsigned long long mask, float* p, float g0, float g1, float g2, float g3) {
    mask = uniform64(mask);
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dword %2, %3, off\n\t"
                 "global_store_dword %2, %4, off offset:16\n\t"
                 "global_store_dword %2, %5, off offset:32\n\t"
                 "global_store_dword %2, %6, off offset:48\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(p), "v"(g0), "v"(g1), "v"(g2), "v"(g3) : "memory");
}
__device__ __forceinline__ void store_dw2_masked(unsigned long long mask, float* p0, float v0, float* p1, float v1) {
    mask = uniform64(mask);
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dword %2, %3, off\n\t"
                 "global_store_dword %4, %5, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(p0), "v"(v0), "v"(p1), "v"(v1) : "memory");
}

// forward, RG >= 2 chains.  Buffers exactly as k_lstm_fwd_pair.
template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_pair_chains(const LstmFwdXcdArgs a) {
    static_assert(RG >= 2 && RG <= 4, "chains");
    __shared__ __attribute__((aligned(16))) float red[RG][4][64 * 4];        // [chain][wave][cell lane][gate]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int grp = role.xcd / PNX, cu = (role.xcd % PNX) * NCU + role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + PGRP - 1) / PGRP, row0 = grp * rpx;
    if (row0 >= B) return;

    f32x4 W[PNW];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * PNW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < PNW; ++i) W[i] = wp[i * 64];
    }
    // wave c < RG is the cell wave of chain (= row group) c: lane = 16 i + 4 bb + e -> row 4c + i, unit 16 cu + 4 bb + e
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const unsigned long long actmask = uniform64(__builtin_amdgcn_ballot_w64(act));
    const int rowc = row < B ? row : B - 1;                          // inactive lanes load a valid row and discard it
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * PH + unit] : 0.0f;
    const size_t hx_step = (size_t)PGRP * 4 * RG * PNQ * 64;
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)grp * 4 + wave) * RG) * PNQ * 64 + lane;   // + chain * PNQ * 64
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) + ((((size_t)grp * 4 + (cu >> 4)) * RG + wave) * PNQ + ((cu >> 2) & 3)) * 64 +
                    16 * (cu & 3) + 4 * cbb + ci;
    const int wofs = ((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3);
    const float* zload = a.Z + (size_t)rowc * PG4 + 64 * cu + 16 * cbb + ce;             // + t * B * PG4
    float* zstore = a.Z + (size_t)rowc * PG4 + 64 * cu + 16 * cbb + ce;
    const size_t hofs = (size_t)rowc * PH + unit;
    float o_h = 0.f, o_g[4] = {0.f, 0.f, 0.f, 0.f};
    int o_t = a.t0;
    unsigned long long omask = 0;                                    // nothing to write back yet
    if (cellw) chain_issue_dw4(zload + (size_t)a.t0 * B * PG4);     // x-part of every chain's first update
    vm_wait<0>();
    // the compiler's own loads (weights, cell state) are waited for HERE: its waitcnt bookkeeping must be empty inside the loop
#pragma unroll
    for (int i = 0; i < PNW; ++i) asm volatile("" : "+v"(W[i]));
    asm volatile("" : "+v"(cp));
    int cons = RG + 1;                                               // phases left whose waits must be vmcnt(0)

    auto phase = [&](auto Cc, const int t) __attribute__((always_inline)) -> bool {
        constexpr int C = decltype(Cc)::value, CN = (C + 1) % RG;
        constexpr int W10 = (C + RG - 2) % RG, W1 = (C + RG - 1) % RG;      // issued 10 operations / 1 store behind this phase's poll
        const int tn = C + 1 < RG ? t : t + 1;                               // time index of the next phase
        const bool pending = !(C == 0 && t == a.t0);
        const bool issue_next = !(C == RG - 1 && t + 1 >= a.t1);
        // ---- 1: the fragments of (C, t)
        f32x4 av[4];
        {
            bool ready = false;
            if (pending) {
                if (cons > 0) vm_wait<0>();
                else if (wave == W10) vm_wait<10>();
                else if (wave == W1) vm_wait<1>();
                else vm_wait<0>();
                chain_adopt_fragments<C>(av);
                ready = chain_fragments_ready(av);
                if (a.prof != nullptr && lane == 0) {          // diagnostics: early polls checked / found not ready, per wave
                    atomicAdd(a.prof + wave, 1ull);
                    if (!ready) atomicAdd(a.prof + 4 + wave, 1ull);
                }
            }
            if (!ready) {
                const bool fail = !wait_all_fragments<4>(hx_in + (size_t)t * hx_step + C * PNQ * 64, av, a.spin_limit, a.err_flag, (a.variant & XCD_NO_POLL_SLEEP) != 0);
                if (fail && lane == 0) {
                    __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_fail = 1;
                }
            }
        }
        // ---- 2: 256 MFMAs, two accumulators (fragment pairs) issued alternately
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#define CHAIN_FWD_B(B_)                                                                             \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                          \
            acc0 = mfma44<B_>(av[2 * qp][e_], W[32 * qp + B_][e_], acc0);                           \
            acc1 = mfma44<B_>(av[2 * qp + 1][e_], W[32 * qp + 16 + B_][e_], acc1);                  \
        }
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            CHAIN_FWD_B(0) CHAIN_FWD_B(1) CHAIN_FWD_B(2) CHAIN_FWD_B(3) CHAIN_FWD_B(4) CHAIN_FWD_B(5) CHAIN_FWD_B(6) CHAIN_FWD_B(7)
            CHAIN_FWD_B(8) CHAIN_FWD_B(9) CHAIN_FWD_B(10) CHAIN_FWD_B(11) CHAIN_FWD_B(12) CHAIN_FWD_B(13) CHAIN_FWD_B(14) CHAIN_FWD_B(15)
        }
#undef CHAIN_FWD_B
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3: the poll of the next phase goes out now (checked when it starts); behind it the slow traffic of the cell wave
        // that updated in the phase before: its outputs, and the x-part of its next update
        if (issue_next) {
            const f32x4* pn = hx_in + (size_t)tn * hx_step + CN * PNQ * 64;
            chain_issue_fragments<CN>(pn, pn + 64, pn + 128, pn + 192);
        }
        if (wave == W1) {
            store_dw2_masked(omask, a.Cs + (size_t)(o_t + 1) * B * PH + hofs, cp, a.Hs + (size_t)(o_t + 1) * B * PH + hofs, o_h);
            store_dw4_stride16_masked(omask, zstore + (size_t)o_t * B * PG4, o_g[0], o_g[1], o_g[2], o_g[3]);
            const int tz = C == 0 ? t : (t + 1 < a.t1 ? t + 1 : t);         // time index of that wave's next update
            chain_issue_dw4(zload + (size_t)tz * B * PG4);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4: K-split partials meet in LDS
        {
            const f32x4 s = acc0 + acc1;
            float* rp = &red[C][wave][0] + wofs;
#pragma unroll
            for (int i = 0; i < 4; ++i) rp[64 * i] = s[i];
        }
        __syncthreads();
        if (s_fail) return false;
        // ---- 5: cell update of chain C by wave C.  Its x-part was issued (last) in step 3 of the phase after its previous update;
        // behind it: the polls of the phases in between and of this one, 4 loads each
        if (wave == C) {
            if (cons > 0 || !issue_next) vm_wait<0>(); else vm_wait<4 * (RG - 1)>();
            float zin[4];
            chain_adopt_dw4(zin);
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[C][0][0]) + lane;
                const f32x4 r0 = rsrc[0], r1 = rsrc[64], r2 = rsrc[128], r3 = rsrc[192];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                const CellOut co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
                g_si = co.si; g_tj = co.tj; g_sf = co.sf; g_so = co.so;
            }
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            if (ce == 0) store_sc1(hx_out + (size_t)(t + 1) * hx_step, hv);
            o_h = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_t = t; omask = actmask;
        }
        if (cons > 0) --cons;
        return true;
    };

    for (int t = a.t0; t < a.t1; ++t) {
        if (!phase(std::integral_constant<int, 0>{}, t)) return;
        if (!phase(std::integral_constant<int, 1>{}, t)) return;
        if constexpr (RG > 2) { if (!phase(std::integral_constant<int, 2 % RG>{}, t)) return; }
        if constexpr (RG > 3) { if (!phase(std::integral_constant<int, 3 % RG>{}, t)) return; }
    }
    // the last chain's last outputs (every other chain's went out in the phase behind its update)
    if (wave == RG - 1) {
        store_dw2_masked(omask, a.Cs + (size_t)(o_t + 1) * B * PH + hofs, cp, a.Hs + (size_t)(o_t + 1) * B * PH + hofs, o_h);
        store_dw4_stride16_masked(omask, zstore + (size_t)o_t * B * PG4, o_g[0], o_g[1], o_g[2], o_g[3]);
    }
}

// inbox [2 slots][4 pairs][64 dest][RG][64 producers][16 units][4 rows]; KhXb [64 cu][4 w][64][64 lanes][4]
template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_pair(const LstmBwdXcdArgs a) {
    constexpr int LPR = 4;                        // inbox words per lane and row group: 16 producers x 16 units / 64 lanes
    __shared__ __attribute__((aligned(16))) float psum[RG][4][64][4];
    __shared__ __attribute__((aligned(16))) float dzA[RG][64][4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int grp = role.xcd / PNX, cu = (role.xcd % PNX) * NCU + role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + PGRP - 1) / PGRP, row0 = grp * rpx;
    if (row0 >= B) return;

    f32x4 W[PNW];         // component e' of word i = weight register 4i + e' = (cg = /64, k = %64): Kh[256w + 64cg + lane][64cu + k]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * PNW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < PNW; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const size_t hi = (size_t)row * PH + unit;
    float dcv = act ? a.dc[hi] : 0.0f;
    const size_t slot_w = (size_t)PGRP * PCU * RG * PCU * 16;                // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);

    // consumer: (dest = cu, row group rg): 64 producers x 16 units; wave w takes producers 16w .. 16w+15, 4 words per lane
    const size_t in_base = (((size_t)grp * PCU + cu) * RG) * PCU * 16 + (size_t)(16 * wave) * 16 + lane;
    // producer: lane l of column group cg -> destination 16w + 4cg + l/16, word (dest, rg, producer = cu, l % 16)
    size_t out_ofs[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
        out_ofs[cg] = (((size_t)grp * PCU + 16 * wave + 4 * cg + (lane >> 4)) * RG) * PCU * 16 + (size_t)cu * 16 + (lane & 15);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};

    float n_si = 0.f, n_tj = 0.f, n_sf = 0.f, n_so = 0.f, n_ct = 0.f, n_cp = 0.f, n_dh = 0.f;
    if (act && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        const float* gp = a.Z + ((size_t)t * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        n_si = gp[0]; n_tj = gp[4]; n_sf = gp[8]; n_so = gp[12];
        n_ct = a.Cs[(size_t)(t + 1) * B * PH + hi]; n_cp = a.Cs[(size_t)t * B * PH + hi];
        n_dh = a.dH[(size_t)t * B * PH + hi];
    }

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        const float si = n_si, tj = n_tj, sf = n_sf, so = n_so, ct = n_ct, cpv = n_cp, dht = n_dh;
        float* gp = a.Z + ((size_t)t * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        // ---- A: consume
        f32x4 wsum[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) wsum[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < a.T) {
            f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base;
            f32x4 v[RG][LPR];
            bool fail = false;
            for (int spins = 0;; ++spins) {
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int k = 0; k < LPR; ++k) v[rg][k] = load_sc1(in + (size_t)rg * PCU * 16 + k * 64);
                drain_vmem();
                bool ok = true;
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int k = 0; k < LPR; ++k) { asm volatile("" : "+v"(v[rg][k])); ok &= frag_ready(v[rg][k]); }
                if (__all(ok)) break;
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int k = 0; k < LPR; ++k) {
                    store_sc1(in + (size_t)rg * PCU * 16 + k * 64, fill);
                    wsum[rg] = (k == 0) ? v[rg][0] : wsum[rg] + v[rg][k];
                }
        }
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) *reinterpret_cast<f32x4*>(&psum[rg][wave][lane][0]) = wsum[rg];
        __syncthreads();
        if (s_fail) return;

        // ---- B: gate gradients (wave rg < RG: lane = 16 i + 4 bb + e); lane group l/16 of a wave holds producers 16w + 4k + l/16
        float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
        if (cellw) {
            if (act) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        dh_rec += psum[wave][w][16 * g4 + 4 * cbb + ce][ci];
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cpv, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                if (!(a.variant & XCD_DEFER_OUTPUTS)) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }
                dcv = cg.dc_out;
            }
            float* f = &dzA[wave][4 * ce + ci][cbb];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
        }
        __syncthreads();
        drain_vmem();                      // resets (and dz stores) landed before anything is published: see k_lstm_bwd_xcd
        if ((a.variant & XCD_DEFER_OUTPUTS) && act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }
        if (act && t > a.t0) {
            const float* gn = a.Z + ((size_t)(t - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
            n_si = gn[0]; n_tj = gn[4]; n_sf = gn[8]; n_so = gn[12];
            n_ct = cpv; n_cp = a.Cs[(size_t)(t - 1) * B * PH + hi];
            n_dh = a.dH[(size_t)(t - 1) * B * PH + hi];
        }

        // ---- C: produce the partials of dh_{t-1}: 256 destination units per wave = 4 column groups of 64
        if (t > 0) {
            f32x4 av[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) av[rg] = *reinterpret_cast<const f32x4*>(&dzA[rg][lane][0]);
            f32x4 acc[RG][4];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc[rg][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#define PAIR_BWD_B(B_)                                                                                      \
            _Pragma("unroll") for (int rg = 0; rg < RG; ++rg)                                               \
                _Pragma("unroll") for (int cg = 0; cg < 4; ++cg)                                            \
                    acc[rg][cg] = mfma44<B_>(av[rg][v], W[16 * cg + 4 * v + (B_ >> 2)][B_ & 3], acc[rg][cg]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                PAIR_BWD_B(0) PAIR_BWD_B(1) PAIR_BWD_B(2) PAIR_BWD_B(3) PAIR_BWD_B(4) PAIR_BWD_B(5) PAIR_BWD_B(6) PAIR_BWD_B(7)
                PAIR_BWD_B(8) PAIR_BWD_B(9) PAIR_BWD_B(10) PAIR_BWD_B(11) PAIR_BWD_B(12) PAIR_BWD_B(13) PAIR_BWD_B(14) PAIR_BWD_B(15)
            }
#undef PAIR_BWD_B
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            f32x4* out = inbox + (size_t)(t & 1) * slot_w;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) store_sc1(out + out_ofs[cg] + (size_t)rg * PCU * 16, acc[rg][cg]);
        }
    }
    if (act) a.dc[hi] = dcv;
}

// backward, RG >= 2 chains.  Buffers exactly as k_lstm_bwd_pair.  Phase (c, t):
//   A  the 4 inbox words per lane of row group c [issued during the previous phase] -> resets -> per-wave sums to LDS -> barrier
//   B  wave c: gate gradients of its rows, dz slice in A-register order to LDS -> barrier
//   C  poll of the next phase, then (wave c) the row-major dz stores and the inputs of its next update; 256 MFMAs; the resets of
//      A are awaited by COUNT (they are the oldest operations in the queue), then the partials are published.
template <int RG>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_pair_chains(const LstmBwdXcdArgs a) {
    static_assert(RG >= 2 && RG <= 4, "chains");
    __shared__ __attribute__((aligned(16))) float psum[4][64][4];
    __shared__ __attribute__((aligned(16))) float dzA[64][4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int grp = role.xcd / PNX, cu = (role.xcd % PNX) * NCU + role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + PGRP - 1) / PGRP, row0 = grp * rpx;
    if (row0 >= B) return;

    f32x4 W[PNW];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * PNW) * 64 + lane;
#pragma unroll
        for (int i = 0; i < PNW; ++i) W[i] = wp[i * 64];
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const unsigned long long actmask = uniform64(__builtin_amdgcn_ballot_w64(act));
    const int rowc = row < B ? row : B - 1;
    const size_t hic = (size_t)rowc * PH + unit;
    float dcv = act ? a.dc[hic] : 0.0f;
    float n_ct = (cellw && a.t1 > a.t0) ? a.Cs[(size_t)a.t1 * B * PH + hic] : 0.0f;
    const size_t slot_w = (size_t)PGRP * PCU * RG * PCU * 16;                // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    // consumer: (dest = cu, row group c): 64 producers x 16 units; wave w takes producers 16w .. 16w+15, 4 words per lane
    const size_t in_base = (((size_t)grp * PCU + cu) * RG) * PCU * 16 + (size_t)(16 * wave) * 16 + lane;      // + c * PCU * 16
    size_t out_ofs[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
        out_ofs[cg] = (((size_t)grp * PCU + 16 * wave + 4 * cg + (lane >> 4)) * RG) * PCU * 16 + (size_t)cu * 16 + (lane & 15);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
    const float* gload = a.Z + (size_t)rowc * PG4 + 64 * cu + 16 * cbb + ce;              // + t * B * PG4
    float* gstore = a.Z + (size_t)rowc * PG4 + 64 * cu + 16 * cbb + ce;

    // inputs of every cell wave's first update (step t1 - 1): gates, c_{t} (c_prev), dH; c_{t+1} is carried in n_ct
    if (cellw && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        chain_issue_dw4(gload + (size_t)t * B * PG4);
        chain_issue_dw_a(a.Cs + (size_t)t * B * PH + hic);
        chain_issue_dw_b(a.dH + (size_t)t * B * PH + hic);
    }
    vm_wait<0>();
#pragma unroll
    for (int i = 0; i < PNW; ++i) asm volatile("" : "+v"(W[i]));
    asm volatile("" : "+v"(dcv)); asm volatile("" : "+v"(n_ct));
    int cons = RG + 1;

    auto phase = [&](auto Cc, const int t) __attribute__((always_inline)) -> bool {
        constexpr int C = decltype(Cc)::value, CN = (C + 1) % RG, WP = (C + RG - 1) % RG;       // WP: the cell wave of the phase before
        const int tn = C + 1 < RG ? t : t - 1;                               // time index of the next phase
        const bool pending = !(C == 0 && t == a.t1 - 1);
        const bool next_exists = !(C == RG - 1 && t == a.t0);
        const bool has_in = t + 1 < a.T;                                     // dh partials arrive for this step
        const bool produce = t > 0;
        const bool issue_next = next_exists && (tn + 1 < a.T);
        if (!has_in || !produce || !pending || !issue_next) cons = RG + 1;
        // ---- A: consume.  Behind the poll's loads this wave issued (previous phase, part C) 10 operations if it was that phase's
        // cell wave, and 4 publish stores
        f32x4 wsum = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_in) {
            f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base + (size_t)C * PCU * 16;
            f32x4 v[4];
            bool ready = false;
            if (pending) {
                if (cons > 0) vm_wait<0>();
                else if (wave == WP) vm_wait<14>();
                else vm_wait<4>();
                chain_adopt_fragments<C>(v);
                ready = chain_fragments_ready(v);
                if (a.prof != nullptr && lane == 0) {
                    atomicAdd(a.prof + wave, 1ull);
                    if (!ready) atomicAdd(a.prof + 4 + wave, 1ull);
                }
            }
            if (!ready) {
                bool fail = false;
                for (int spins = 0;; ++spins) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = load_sc1(in + k * 64);
                    drain_vmem();
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                    if (__all(ok)) break;
                    if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                    if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
                }
                if (fail && lane == 0) {
                    __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_fail = 1;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                store_sc1(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        }
        *reinterpret_cast<f32x4*>(&psum[wave][lane][0]) = wsum;
        __syncthreads();
        if (s_fail) return false;
        // ---- B: gate gradients of chain C by wave C.  Its inputs were issued in part C of its previous phase; behind them: that
        // phase's publish (4), the RG - 1 phases in between (4 resets + 4 poll loads + 4 publish stores each), this phase's resets (4)
        float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
        if (wave == C) {
            if (cons > 0) vm_wait<0>(); else vm_wait<12 * RG - 4>();
            float nq[4], n_cp, n_dh;
            chain_adopt_dw4(nq);
            chain_adopt_dw2(n_cp, n_dh);
            if (act) {
                float dh_rec = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        dh_rec += psum[w][16 * g4 + 4 * cbb + ce][ci];
                const CellGrad cg = cell_backward(nq[0], nq[1], nq[2], nq[3], n_ct, n_cp, dcv, n_dh + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                dcv = cg.dc_out;
            }
            n_ct = n_cp;                                                      // c_t of the next update (step t-1) is this step's c_{t-1}
            float* f = &dzA[4 * ce + ci][cbb];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
        }
        __syncthreads();
        // ---- C: next poll, slow traffic of wave C, MFMAs, publish
        if (issue_next) {
            f32x4* inn = inbox + (size_t)((tn + 1) & 1) * slot_w + in_base + (size_t)CN * PCU * 16;
            chain_issue_fragments<CN>(inn, inn + 64, inn + 128, inn + 192);
        }
        if (wave == C) {
            store_dw4_stride16_masked(actmask, gstore + (size_t)t * B * PG4, di, dj, df, dg);        // row-major dz for the weight-gradient GEMMs
            const int tq = t > a.t0 ? t - 1 : t;
            chain_issue_dw4(gload + (size_t)tq * B * PG4);
            chain_issue_dw_a(a.Cs + (size_t)tq * B * PH + hic);
            chain_issue_dw_b(a.dH + (size_t)tq * B * PH + hic);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (produce) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&dzA[lane][0]);
            f32x4 acc[4];
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#define CHAIN_BWD_B(B_)                                                                                     \
            _Pragma("unroll") for (int cg = 0; cg < 4; ++cg)                                                \
                acc[cg] = mfma44<B_>(av[v], W[16 * cg + 4 * v + (B_ >> 2)][B_ & 3], acc[cg]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                CHAIN_BWD_B(0) CHAIN_BWD_B(1) CHAIN_BWD_B(2) CHAIN_BWD_B(3) CHAIN_BWD_B(4) CHAIN_BWD_B(5) CHAIN_BWD_B(6) CHAIN_BWD_B(7)
                CHAIN_BWD_B(8) CHAIN_BWD_B(9) CHAIN_BWD_B(10) CHAIN_BWD_B(11) CHAIN_BWD_B(12) CHAIN_BWD_B(13) CHAIN_BWD_B(14) CHAIN_BWD_B(15)
            }
#undef CHAIN_BWD_B
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // MFMA result -> VMEM store data: wait states by hand (inline-asm stores)
            // the resets of part A must have landed before this block publishes anything (two-slot argument of k_lstm_bwd_rs);
            // behind them: the poll (4) and, for wave C, 4 dz stores + 6 input loads
            if (cons > 0 || !issue_next) vm_wait<0>();
            else if (wave == C) vm_wait<14>();
            else vm_wait<4>();
            f32x4* out = inbox + (size_t)(t & 1) * slot_w + (size_t)C * PCU * 16;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) store_sc1(out + out_ofs[cg], acc[cg]);
        }
        if (cons > 0) --cons;
        return true;
    };

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        if (!phase(std::integral_constant<int, 0>{}, t)) return;
        if (!phase(std::integral_constant<int, 1>{}, t)) return;
        if constexpr (RG > 2) { if (!phase(std::integral_constant<int, 2 % RG>{}, t)) return; }
        if constexpr (RG > 3) { if (!phase(std::integral_constant<int, 3 % RG>{}, t)) return; }
    }
    if (act) a.dc[hic] = dcv;
}

// Kh [1024][4096] (packed gate columns) -> the register images of the pair kernels (same convention as k_repack_kh_xcd):
//   fwd word i (= 16q + b, q < 4), component e of (cu, w), lane l:  Kh[256w + 64q + 16(b/4) + 4(b%4) + e][64cu + l]
//   bwd word i, component e' of (cu, w), lane l, r = 4i + e' = 64cg + k (cg < 4):  Kh[256w + 64cg + l][64cu + k]
__device__ __forceinline__ void repack_kh_pair_body(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int b, int nb) {
    const int total = PCU * 4 * PNW * 64;           // f32x4 words per copy
    for (int idx = b * blockDim.x + threadIdx.x; idx < total; idx += nb * blockDim.x) {
        const int l = idx & 63, i = (idx >> 6) & 63, w = (idx >> 12) & 3, cu = idx >> 14;
        {
            const int q = i >> 4, b = i & 15;
            const int k0 = PKW * w + 64 * q + 16 * (b >> 2) + 4 * (b & 3);
            const float* src = Kh + (size_t)k0 * PG4 + 64 * cu + l;
            float4 v;
            v.x = src[0]; v.y = src[PG4]; v.z = src[2 * (size_t)PG4]; v.w = src[3 * (size_t)PG4];
            reinterpret_cast<float4*>(fwd)[idx] = v;
        }
        {
            const int r = 4 * i, cg = r >> 6, k = r & 63;
            reinterpret_cast<float4*>(bwd)[idx] =
                *reinterpret_cast<const float4*>(Kh + (size_t)(PKW * w + 64 * cg + l) * PG4 + 64 * cu + k);
        }
    }
}
__global__ void k_repack_kh_pair(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    repack_kh_pair_body(Kh, fwd, bwd, blockIdx.x, gridDim.x);
}

#include "lstm_pair16.h"        // hidden 1024 on the bf16 matrix pipe (k_lstm_*_pair16): textually part of this file

// every layer's K_h into every layout the recurrent kernels read, ONE launch (blockIdx.y = 2 * layer + {0: the column-split
// kernels' fragment order, 1: the XCD / XCD-pair register image}): the four repacks of a two-layer model were 56 us of a
// 2.85 ms cfg-C step as separate launches
__global__ void k_repack_kh_all(const RepackAllArgs a, const StepIncArgs inc) {
    if (inc.step != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) step_increment_body(inc);
    const int l = blockIdx.y >> 1, kind = blockIdx.y & 1;
    if (kind == 0) { if (a.mode != 1) repack_kh_chunked(a.Kh[l], a.cf[l], a.cb[l], a.Hp, blockIdx.x, gridDim.x); return; }
    if (a.xf[l] == nullptr || a.mode == 2) return;
    if (a.Hp == SH) repack_kh_slice_body(a.Kh[l], a.xf[l], a.xb[l], blockIdx.x, gridDim.x);
    else if (a.Hp == PH && a.bx3) repack_kh_pair16_body(a.Kh[l], a.xf[l], a.xb[l], blockIdx.x, gridDim.x);
    else if (a.Hp == PH) repack_kh_pair_body(a.Kh[l], a.xf[l], a.xb[l], blockIdx.x, gridDim.x);
    else if (a.bx3) repack_kh_xcd16_body(a.Kh[l], a.xf[l], a.xb[l], blockIdx.x, gridDim.x);
    else repack_kh_xcd_body(a.Kh[l], a.xf[l], a.xb[l], blockIdx.x, gridDim.x);
}

}  // namespace

hipError_t launch_repack_kh_all(hipStream_t s, const RepackAllArgs& a, const StepIncArgs* inc) {
    if (a.n <= 0) return hipSuccess;
    if (a.n > REPACK_MAX_LAYERS) return hipErrorInvalidValue;
    for (int l = 0; l < a.n; ++l) if (a.xf[l] != nullptr && !(a.Hp == XH || a.Hp == PH || a.Hp == SH)) return hipErrorInvalidValue;
    const long long total = (long long)a.Hp * 4 * a.Hp / 4;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 1024);
    if (a.bx3 && a.Hp != XH && a.Hp != PH) return hipErrorInvalidValue;
    StepIncArgs none{};
    hipLaunchKernelGGL(k_repack_kh_all, dim3(blocks, 2 * a.n), dim3(256), 0, s, a, inc ? *inc : none);
    return hipGetLastError();
}

// row groups of 4 rows per weight copy: hidden 512 -> 8 copies (one per XCD), 3 is padded to 4 (the lane mapping of
// k_lstm_bwd_xcd needs a divisor of 4); hidden 1024 -> 4 copies (one per XCD pair), 1 .. 4
// row groups of a launch whose rows are packed `rpx` per XCD (0: spread over all eight): the bf16-split kernels take up to 16
// rows per XCD at the same MFMA cost, so a batch may sit on fewer XCDs than the 8-way split would use
static int xcd_row_groups(int B, int Hp = 512);
static int xcd_row_groups_packed(int B, int rpx, int Hp = 512) {
    const int rg = xcd_row_groups(B, Hp);
    if (rpx <= 0) return rg;
    const int need = (rpx + 3) / 4;
    if (Hp == PH) return std::max(rg, need);      // (the pair kernels take three row groups as they are)
    return std::max(rg, need == 3 ? 4 : need);
}
static int xcd_row_groups(int B, int Hp) {
    if (Hp == PH) return ((B + PGRP - 1) / PGRP + 3) / 4;
    if (Hp == SH) return ((B + NSL - 1) / NSL + 3) / 4;         // sixteen slices; 1 or 2 row groups up to 128 rows
    const int rpx = (B + NXCD - 1) / NXCD;
    const int rg = (rpx + 3) / 4;
    return rg == 3 ? 4 : rg;
}

bool lstm_xcd_supported(int B, int Hp) {
    if (Hp == SH) return B >= 1 && xcd_row_groups(B, Hp) <= 2;
    return (Hp == XH || Hp == PH) && B >= 1 && xcd_row_groups(B, Hp) <= 4;
}
int lstm_xcd_max_rows(int Hp) { return Hp == PH ? 16 * PGRP : (Hp == XH ? 16 * NXCD : (Hp == SH ? 8 * NSL : 0)); }

long long lstm_xcd_hx_floats(int B, int T, int Hp, bool bx3, int rpx) {
    if (Hp == PH && bx3) return (long long)(T + 1) * (4 * xcd_row_groups_packed(B, rpx, PH)) * PGRP * HXW16P * 4;       // rows x pairs x 6 KiB, see lstm_pair16.h
    if (Hp == PH) return (long long)(T + 1) * PGRP * 4 * xcd_row_groups(B, Hp) * PNQ * 64 * 4;
    if (Hp == SH) return (long long)(T + 1) * NSL * 4 * xcd_row_groups(B, Hp) * 64 * 4;
    if (bx3) return (long long)(T + 1) * (4 * xcd_row_groups_packed(B, rpx)) * NXCD * HXW16 * 4;      // rows x XCDs x 3 KiB, see k_lstm_fwd_xcd16
    return (long long)(T + 1) * NXCD * 4 * xcd_row_groups_packed(B, rpx) * 2 * 64 * 4;
}
long long lstm_xcd_inbox_floats(int B, int Hp, int rpx) {
    if (Hp == PH) return 2LL * PGRP * PCU * xcd_row_groups_packed(B, rpx, PH) * PCU * 16 * 4;
    if (Hp == SH) return 2LL * NSL * SCU * SCU * xcd_row_groups(B, Hp) * 16 * 4;
    return 2LL * NXCD * NCU * NCU * xcd_row_groups_packed(B, rpx) * 16 * 4;
}
// the bf16-split kernels take up to 16 rows per XCD at the same MFMA cost: the fewest XCDs that hold B rows, rows spread evenly
int lstm_xcd16_packed_rows(int B, int Hp) {      // (hidden 1024: per XCD PAIR)
    const int nx = (B + 15) / 16;
    return nx >= 1 && nx <= (Hp == PH ? PGRP : NXCD) ? (B + nx - 1) / nx : 0;
}
long long lstm_xcd_weight_floats(int Hp, bool bx3) { return (bx3 && (Hp == XH || Hp == PH)) ? (long long)Hp * 4 * Hp * 3 / 2 : (long long)Hp * 4 * Hp; }   // three bf16 planes
// The bf16-split kernels pay 1536 MFMA cycles per step for any row count, the fp32 ones 1024 per row group, and the bf16 hand-off
// is 1.5x the bytes in 3x the load instructions: measured (profiles/r03_xcd16_probe3.log, us per step forward / backward)
// B = 20: 2.68 / 2.17 against 1.81 / 1.64, B = 45: 2.57 / 2.35 against 2.17 / 2.38, B = 100: 2.98 / 3.22 against 3.86 / 4.35.
// Hidden 1024 (k_lstm_*_pair16, profiles/r06_pair16_*.log, us per step forward / backward against the fp32 pair kernels' best variants):
// B = 45: 5.65 / 5.85 against 6.1-6.3 / 7.3, B = 64: 5.6 / 7.2 against 7.5 / 9.65, but B = 20-25: 5.2 / 5.6 against 4.2 / 5.1-5.2 --
// the same rule: from three row groups per weight copy on.
bool lstm_xcd_bx3_pays(int B, int Hp) { return (Hp == XH && xcd_row_groups(B) >= 3) || (Hp == PH && xcd_row_groups(B, PH) >= 3); }
// Rows per XCD that fill the row groups the 8-way split already pays for: the batch then sits on the first
// ceil(B / rows) XCDs and the others are free for another stream's GEMMs (B = 45: 8 rows on 6 XCDs instead of 6 on 8).
int lstm_xcd_packed_rows(int B) { return 4 * xcd_row_groups(B); }
hipError_t launch_repack_kh_xcd(hipStream_t s, const float* Kh, float* fwd, float* bwd, int Hp, bool bx3) {
    if (Hp == SH) hipLaunchKernelGGL(k_repack_kh_slice, dim3(256), dim3(256), 0, s, Kh, fwd, bwd);
    else if (Hp == PH && bx3) hipLaunchKernelGGL(k_repack_kh_pair16, dim3(2048), dim3(256), 0, s, Kh, fwd, bwd);
    else if (Hp == PH) hipLaunchKernelGGL(k_repack_kh_pair, dim3(2048), dim3(256), 0, s, Kh, fwd, bwd);
    else if (bx3) hipLaunchKernelGGL(k_repack_kh_xcd16, dim3(512), dim3(256), 0, s, Kh, fwd, bwd);
    else hipLaunchKernelGGL(k_repack_kh_xcd, dim3(1024), dim3(256), 0, s, Kh, fwd, bwd);
    return hipGetLastError();
}

// Variants chosen per shape (tools/xcd_chain_bench, profiles/r03_xcd_probe6*.log; B = 45: forward 2.21 -> 2.14 us per step with
// the outputs deferred, backward 2.40 -> 2.34 without the sleep; B = 100: deferring costs 4 %, no sleep is neutral)
int lstm_xcd_default_variant(int B, bool forward, int Hp, int rpx, bool bx3) {
    // hidden 1024 on the bf16 pipe (lstm_pair16.h): forward probe + streamed fetch (6.35 -> 5.82 -> 5.37 us per step at B = 45), outputs
    // deferred behind the fetch; backward same-XCD partials through the L2 (6.65 -> 5.64).  NOT XCD_LATE_DRAIN: alone on the chip the
    // reset wait behind the first tile group is neutral, in the step -- where the next update's gates / cell states / dH come out of HBM
    // and the next poll queues behind them -- requesting them 800 ticks later costs 0.25 us per step (cfg-C 405 -> 409, same box)
    if (Hp == PH && bx3) return forward ? (XCD_NO_POLL_SLEEP | XCD_PROBE | XCD_STREAM | XCD_DEFER_OUTPUTS | 4 * XCD_PROBE_DELAY) : (XCD_NO_POLL_SLEEP | XCD_LOCAL_PLAIN);
    // hidden 1024 (profiles/r03_pair_probe3..5.log, us per step without / with chains): backward 9.0 -> 7.2 (three row groups),
    // 6.75 -> 5.1 (two), 12.2 -> 9.4 (four); forward 4.65 -> 4.2 with two row groups, but 6.1 -> 6.35 / 7.4 -> 8.4 with three / four
    // (its early polls are ready 95 % of the time: the hand-off IS hidden, the per-phase instruction overhead is what is left).
    // Also measured and dropped for the forward kernel without chains: poll rounds that re-fetch only the incomplete fragments
    // (6.06 -> 6.32, profiles/r03_pair_probe6.log)
    if (Hp == PH) {
        const int rg = xcd_row_groups(B, PH);
        if (rg >= 2 && (!forward || rg == 2)) return XCD_CHAINS | XCD_NO_POLL_SLEEP;
        return XCD_NO_POLL_SLEEP;
    }
    // backward on the bf16 pipe at hidden 512: the partials of four destination tiles leave while the other four multiply (B = 45 packed 15 per XCD:
    // 3.01 -> 2.78 us per step, profiles/r06_xcd16_probe_variants.log; the forward kernel's probe / streamed fetch: 3.06 -> 3.03, not taken)
    if (!forward) return (bx3 && Hp == XH) ? (XCD_NO_POLL_SLEEP | XCD_GROUP_STORES) : XCD_NO_POLL_SLEEP;
    if (Hp == SH) return XCD_DEFER_OUTPUTS | XCD_NO_POLL_SLEEP;
    return xcd_row_groups_packed(B, rpx) <= 2 ? (XCD_DEFER_OUTPUTS | XCD_NO_POLL_SLEEP) : XCD_NO_POLL_SLEEP;
}

hipError_t launch_lstm_fwd_xcd(hipStream_t s, const LstmFwdXcdArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    const dim3 grid(NXCD * NCU), block(256);
    if (a.Hp == SH) {
        if (a.prof || a.bx3 || a.rpx || a.progress) return hipErrorInvalidValue;
        switch (xcd_row_groups(a.B, SH)) {
            case 1: hipLaunchKernelGGL((k_lstm_fwd_slice<1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_fwd_slice<2>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.Hp == PH && a.bx3) {
        if (a.prof) {
            if (xcd_row_groups_packed(a.B, a.rpx, PH) != 3 || a.progress) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_lstm_fwd_pair16<3, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        switch (xcd_row_groups_packed(a.B, a.rpx, PH)) {
            case 1: hipLaunchKernelGGL((k_lstm_fwd_pair16<1, false>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_fwd_pair16<2, false>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((k_lstm_fwd_pair16<3, false>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_fwd_pair16<4, false>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.Hp == PH) {
        const int rg = xcd_row_groups(a.B, PH);
        if (a.prof && !((a.variant & XCD_CHAINS) && rg >= 2)) return hipErrorInvalidValue;
        if ((a.variant & XCD_CHAINS) && rg >= 2) {
            switch (rg) {
                case 2: hipLaunchKernelGGL((k_lstm_fwd_pair_chains<2>), grid, block, 0, s, a); break;
                case 3: hipLaunchKernelGGL((k_lstm_fwd_pair_chains<3>), grid, block, 0, s, a); break;
                default: hipLaunchKernelGGL((k_lstm_fwd_pair_chains<4>), grid, block, 0, s, a); break;
            }
            return hipGetLastError();
        }
        switch (rg) {
            case 1: hipLaunchKernelGGL((k_lstm_fwd_pair<1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_fwd_pair<2>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((k_lstm_fwd_pair<3>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_fwd_pair<4>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.bx3) {
        if (a.prof) {
            if (xcd_row_groups(a.B) != 2) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_lstm_fwd_xcd16<2, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        switch (xcd_row_groups_packed(a.B, a.rpx)) {     // = the row count lstm_xcd_hx_floats sized (and the caller filled) the buffer for
            case 1: hipLaunchKernelGGL((k_lstm_fwd_xcd16<1, false>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_fwd_xcd16<2, false>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_fwd_xcd16<4, false>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.prof) {
        if (xcd_row_groups(a.B) != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_lstm_fwd_xcd<2, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    switch (xcd_row_groups(a.B)) {
        case 1: hipLaunchKernelGGL((k_lstm_fwd_xcd<1, false>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lstm_fwd_xcd<2, false>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_fwd_xcd<4, false>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lstm_bwd_xcd(hipStream_t s, const LstmBwdXcdArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    const dim3 grid(NXCD * NCU), block(256);
    if (a.Hp == SH) {
        if (a.prof || a.bx3 || a.rpx) return hipErrorInvalidValue;
        switch (xcd_row_groups(a.B, SH)) {
            case 1: hipLaunchKernelGGL((k_lstm_bwd_slice<1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_bwd_slice<2>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.Hp == PH && a.bx3) {
        if (a.prof) {
            if (xcd_row_groups_packed(a.B, a.rpx, PH) != 3) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_lstm_bwd_pair16<3, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        switch (xcd_row_groups_packed(a.B, a.rpx, PH)) {
            case 1: hipLaunchKernelGGL((k_lstm_bwd_pair16<1, false>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_bwd_pair16<2, false>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((k_lstm_bwd_pair16<3, false>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_bwd_pair16<4, false>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.Hp == PH) {
        const int rg = xcd_row_groups(a.B, PH);
        if (a.prof && !((a.variant & XCD_CHAINS) && rg >= 2)) return hipErrorInvalidValue;
        if ((a.variant & XCD_CHAINS) && rg >= 2) {
            switch (rg) {
                case 2: hipLaunchKernelGGL((k_lstm_bwd_pair_chains<2>), grid, block, 0, s, a); break;
                case 3: hipLaunchKernelGGL((k_lstm_bwd_pair_chains<3>), grid, block, 0, s, a); break;
                default: hipLaunchKernelGGL((k_lstm_bwd_pair_chains<4>), grid, block, 0, s, a); break;
            }
            return hipGetLastError();
        }
        switch (rg) {
            case 1: hipLaunchKernelGGL((k_lstm_bwd_pair<1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_bwd_pair<2>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((k_lstm_bwd_pair<3>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_bwd_pair<4>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.bx3) {
        if (a.prof) {
            if (xcd_row_groups(a.B) != 2) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_lstm_bwd_xcd16<2, true>), grid, block, 0, s, a);
            return hipGetLastError();
        }
        switch (xcd_row_groups_packed(a.B, a.rpx)) {
            case 1: hipLaunchKernelGGL((k_lstm_bwd_xcd16<1, false>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((k_lstm_bwd_xcd16<2, false>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((k_lstm_bwd_xcd16<4, false>), grid, block, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (a.prof) {
        if (xcd_row_groups(a.B) != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_lstm_bwd_xcd<2, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    switch (xcd_row_groups(a.B)) {
        case 1: hipLaunchKernelGGL((k_lstm_bwd_xcd<1, false>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lstm_bwd_xcd<2, false>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_bwd_xcd<4, false>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fsmg
