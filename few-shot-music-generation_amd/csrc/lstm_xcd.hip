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
// Placement is discovered, not assumed: a block reads its XCC id and takes a ticket from that XCD's counter; the
// (xcd, ticket) pair is its role.  HIP promises nothing about block -> XCD placement, so a ticket >= 32 (an XCD that
// received more than its share) raises the time-out flag like any other failed wait and the caller falls back to the
// column-split kernels: placement decides speed, never results.
#include "fsmg_kernels.h"
#include "lstm_cell.h"
#include <type_traits>

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
            const bool fail = !wait_all_fragments<2 * RG>(hx_in + (size_t)t * hx_step, av, a.spin_limit, a.err_flag, (a.dbg & 32) != 0);
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        if ((a.dbg & 16) && o_have && act) {              // outputs of the step before: behind a successful poll, under the MFMAs
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
            if (a.dbg & 16) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * XH + unit] = cp;
                a.Hs[((size_t)(t + 1) * B + row) * XH + unit] = hn;
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;       // activated gates kept for BPTT
            }
        }
    }
    if ((a.dbg & 16) && o_have && act) {
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
                if (!(a.dbg & 32)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
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
                if (!(a.dbg & 16)) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }       // row-major dz for the weight-gradient GEMMs
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
        if ((a.dbg & 16) && act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }   // variant: dz stores behind the drain, under the MFMAs
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

// ================================================================ round-3 variants of the two kernels above (`pipe` = 1)
// Same chain, same buffers, same MFMA mapping; two changes, both aimed at what the round-2 stamps showed on the critical path
// of a step (hand-off store -> first successful poll: ~1650 ticks in-kernel against 646 between idle CUs; cell update ~800
// ticks on 64 RG of the block's 256 threads):
//  (1) The cell update runs on ALL 256 threads: wave w owns row w of every row group, lane l owns packed gate column l of the
//      CU's 64 (unit 4(l/16) + l%4, gate (l/4)%4) = ONE gate pre-activation per row group.  A lane adds the four waves'
//      partials and the x-part (one coalesced dword), applies its gate's nonlinearity; the gate-0 lane of a unit collects the
//      other three with DPP row shifts and finishes c and h.  All x-part loads and gate / dz stores are 256 contiguous bytes
//      per wave, the row groups of a lane are independent instruction streams (ILP), and the dependent chain behind the
//      barrier is one activation + one tanh instead of five.
//  (2) Nothing slow sits between a hand-off store and the poll that follows it.  A CU returns vector-memory operations in
//      order and vmcnt counts loads and stores alike, so the round-2 kernels' six output stores (c, h, four gates), issued
//      right behind the hand-off store, were what every poll's `vmcnt(0)` waited for.  Here the outputs of step t and the
//      inputs of step t+1 are issued right AFTER the poll of step t+1 has succeeded, i.e. at the head of the MFMA phase that
//      hides them; the values wait in registers / fixed landing registers meanwhile.
// Measured and dropped on the way (profiles/r03_xcd_probe1..4.log): running the row groups of an XCD as TWO chains that the
// same four waves advance alternately, with the poll of one chain issued during the other chain's phase (hand-off latency
// fully hidden).  Correct, but 2.48 / 2.90 us per step against 2.21 / 2.40: every fixed cost of a step (fragment hand-over
// and readiness check, traffic issue, LDS exchange + barrier, the cell update's dependent chain, which loses its ILP) is
// paid per phase, i.e. twice -- with readiness waits AND all slow traffic compiled out that kernel still took 2.32 / 2.71.
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// Landing slots of the asynchronous loads.  A load that is issued in one step and consumed in the next must not live in a
// register the compiler manages: hipcc knows nothing about the pending write-back of an inline-asm load and freely copies /
// re-uses the destination (it did: v_mov of the destination right behind the load).  So these loads target FIXED
// accumulation registers far above anything the compiler allocates here (it uses a[0:31] for the MFMA accumulators), named
// as explicit-register outputs whose value is never used; after the wait that covers them an empty "adopt" asm with the same
// register as its output hands the data to the compiler.  tools/check_xcd_asm.py verifies on the final ISA that nothing else
// touches a[160:175] and that the compiler put no vmcnt wait of its own into the time loop.
#define XCD3_LOAD_DW(REG, PTR) do { float d_; asm volatile("global_load_dword %0, %1, off" : "={" REG "}"(d_) : "v"(PTR) : "memory"); } while (0)
#define XCD3_ADOPT(REG, VAR) asm volatile("" : "={" REG "}"(VAR))

// slot (C, J, K) = a[160 + 8 C + 3 J + K]; row group rg uses C = rg / 2, J = rg % 2, K = which of its (up to three) inputs
template <int C, int J, int K>
__device__ __forceinline__ void issue_dw(const float* p) {
    static_assert(C >= 0 && C < 2 && J >= 0 && J < 2 && K >= 0 && K < 3, "landing slot");
    if constexpr (C == 0 && J == 0 && K == 0) XCD3_LOAD_DW("a160", p);
    if constexpr (C == 0 && J == 0 && K == 1) XCD3_LOAD_DW("a161", p);
    if constexpr (C == 0 && J == 0 && K == 2) XCD3_LOAD_DW("a162", p);
    if constexpr (C == 0 && J == 1 && K == 0) XCD3_LOAD_DW("a163", p);
    if constexpr (C == 0 && J == 1 && K == 1) XCD3_LOAD_DW("a164", p);
    if constexpr (C == 0 && J == 1 && K == 2) XCD3_LOAD_DW("a165", p);
    if constexpr (C == 1 && J == 0 && K == 0) XCD3_LOAD_DW("a168", p);
    if constexpr (C == 1 && J == 0 && K == 1) XCD3_LOAD_DW("a169", p);
    if constexpr (C == 1 && J == 0 && K == 2) XCD3_LOAD_DW("a170", p);
    if constexpr (C == 1 && J == 1 && K == 0) XCD3_LOAD_DW("a171", p);
    if constexpr (C == 1 && J == 1 && K == 1) XCD3_LOAD_DW("a172", p);
    if constexpr (C == 1 && J == 1 && K == 2) XCD3_LOAD_DW("a173", p);
}
template <int C, int J, int K>
__device__ __forceinline__ void adopt_dw(float& v) {
    if constexpr (C == 0 && J == 0 && K == 0) XCD3_ADOPT("a160", v);
    if constexpr (C == 0 && J == 0 && K == 1) XCD3_ADOPT("a161", v);
    if constexpr (C == 0 && J == 0 && K == 2) XCD3_ADOPT("a162", v);
    if constexpr (C == 0 && J == 1 && K == 0) XCD3_ADOPT("a163", v);
    if constexpr (C == 0 && J == 1 && K == 1) XCD3_ADOPT("a164", v);
    if constexpr (C == 0 && J == 1 && K == 2) XCD3_ADOPT("a165", v);
    if constexpr (C == 1 && J == 0 && K == 0) XCD3_ADOPT("a168", v);
    if constexpr (C == 1 && J == 0 && K == 1) XCD3_ADOPT("a169", v);
    if constexpr (C == 1 && J == 0 && K == 2) XCD3_ADOPT("a170", v);
    if constexpr (C == 1 && J == 1 && K == 0) XCD3_ADOPT("a171", v);
    if constexpr (C == 1 && J == 1 && K == 1) XCD3_ADOPT("a172", v);
    if constexpr (C == 1 && J == 1 && K == 2) XCD3_ADOPT("a173", v);
}
// a wave-uniform 64-bit value as an SGPR pair (the "s" constraint of an inline asm does not move a VGPR-resident value itself)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// exec-masked dword stores, issued whatever the mask (an instruction with exec = 0 still takes its slot in the vmcnt order)
__device__ __forceinline__ void store_dw_masked(unsigned long long mask, float* p, float v) {
    mask = uniform64(mask);
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dword %2, %3, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(p), "v"(v) : "memory");
}
// 16-byte hand-off store by the lanes of `mask` (plain scope: stays in the XCD's L2), always issued
__device__ __forceinline__ void store_l2_masked(unsigned long long mask, f32x4* p, f32x4 v) {
    mask = uniform64(mask);
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dwordx4 %2, %3, off\n\t"
                 "s_nop 1\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(p), "v"(v) : "memory");
}
// value of the lane 4*K further up in the same row of 16 lanes (row_shl: lane i reads lane i + 4K; lanes shifted in read 0)
template <int K>
__device__ __forceinline__ float row_up4(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + 4 * K, 0xF, 0xF, true));
}
// value of the lane 4*K further down in the same row (row_shr: lane i reads lane i - 4K)
template <int K>
__device__ __forceinline__ float row_down4(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + 4 * K, 0xF, 0xF, true));
}

// One gate per lane: the nonlinearity of packed gate column `gate` (0 i: sigmoid, 1 j: tanh, 2 f: sigmoid(x + 1), 3 o: sigmoid)
// with exactly the operations of sigmoidf_ / tanhf_ (lstm_cell.h), so a lane produces the bits cell_forward would.
__device__ __forceinline__ float gate_activation(float z, int gate) {
#pragma clang fp contract(off)
    const float x = gate == 2 ? z + 1.0f : z;            // forget_bias = 1 added at run time
    const float arg = gate == 1 ? 2.0f * x : -x;
    const float r = __frcp_rn(1.0f + __expf(arg));
    const float x2 = x * x;
    const float poly = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.05396825f + x2 * 0.02186949f))));
    const float big = 1.0f - 2.0f * r;
    const float th = fabsf(x) < 0.25f ? poly : big;
    return gate == 1 ? th : r;
}

#define XCD3_STAMP(i) if (PROF) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; }
// landing slot of (row group rg, kind k): the slots are named (C, J, K) = (rg / 2, rg % 2, k)
#define XCD3_FOR_RG(...) do { { constexpr int rg = 0; __VA_ARGS__ } if constexpr (RG > 1) { constexpr int rg = 1 % RG; __VA_ARGS__ } \
                              if constexpr (RG > 2) { constexpr int rg = 2 % RG; __VA_ARGS__ } if constexpr (RG > 3) { constexpr int rg = 3 % RG; __VA_ARGS__ } } while (0)

// ---------------------------------------------------------------- forward
// Buffers exactly as k_lstm_fwd_xcd.  Wave w owns row w of every row group, lane l packed column l.
// PROF sums per (block, wave): [0] wait for h_t, [1] output stores + x-part loads + MFMAs, [2] partials -> LDS + barrier,
// [3] cell update up to the hand-off store, [4] loop overhead.
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_xcd3(const LstmFwdXcdArgs a) {
    constexpr int NF = 2 * RG;
    __shared__ __attribute__((aligned(16))) float red[2][RG][4][4][64];     // [step parity][row group][wave][row][column]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;

    f32x4 W[32];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
    }
    const int q = lane >> 4, gate = (lane >> 2) & 3, e = lane & 3;
    const int unit = 16 * cu + 4 * q + e;
    const unsigned long long g0lanes = 0x000F000F000F000Full;       // lanes 16 q + e: the gate-0 lane of every unit
    const unsigned long long wordlanes = 0x0001000100010001ull;     // lanes 16 q: one 16-byte hand-off word per unit quad
    bool actv[RG];
    unsigned long long mall[RG], mg0[RG];
    const float* zload[RG]; float* zstore[RG]; size_t hofs[RG]; f32x4* hx_out[RG];
    float cp[RG], o_a[RG], o_h[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        const int lrow = 4 * rg + wave, row = row0 + lrow;
        const bool act = lrow < rpx && row < B;                     // wave-uniform
        const int rowc = row < B ? row : B - 1;                     // inactive rows load a valid row and discard it
        actv[rg] = act;
        mall[rg] = uniform64(act ? ~0ull : 0ull);
        mg0[rg] = uniform64(act ? g0lanes : 0ull);
        zload[rg] = a.Z + (size_t)rowc * XG4 + 64 * cu + lane;
        zstore[rg] = a.Z + (size_t)rowc * XG4 + 64 * cu + lane;
        hofs[rg] = (size_t)rowc * XH + unit;
        hx_out[rg] = reinterpret_cast<f32x4*>(a.HX) + ((((size_t)xcd * 4 + (cu >> 3)) * RG + rg) * 2 + ((cu >> 2) & 1)) * 64 +
                     16 * (cu & 3) + 4 * q + wave;
        cp[rg] = (act && gate == 0) ? a.Cs[(size_t)a.t0 * B * XH + hofs[rg]] : 0.0f;
        o_a[rg] = 0.f; o_h[rg] = 0.f;
    }
    const size_t hx_step = (size_t)NXCD * 4 * RG * 2 * 64;
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)xcd * 4 + wave) * RG) * 2 * 64 + lane;
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    bool o_have = false;
    // x-part of the first step
    XCD3_FOR_RG(issue_dw<rg / 2, rg % 2, 0>(zload[rg] + (size_t)a.t0 * B * XG4););
    vm_wait<0>();
    // the compiler's own loads (weights, cell state) are waited for HERE: its waitcnt bookkeeping must be empty inside the loop
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(W[i]));
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) asm volatile("" : "+v"(cp[rg]));

    for (int t = a.t0; t < a.t1; ++t) {
        XCD3_STAMP(4)
        f32x4 av[NF];
        {
            // the poll's vmcnt(0) finds nothing slow in front of it: the hand-off stores of the step before, and the output
            // stores / x-part loads that were issued a whole MFMA phase ago
            const bool fail = !(a.dbg & 2) && !wait_all_fragments<NF>(hx_in + (size_t)t * hx_step, av, a.spin_limit, a.err_flag);
            if (a.dbg & 2) {
#pragma unroll
                for (int j = 0; j < NF; ++j) av[j] = load_sc1(hx_in + (size_t)t * hx_step + j * 64);
                drain_vmem();
#pragma unroll
                for (int j = 0; j < NF; ++j) asm volatile("" : "+v"(av[j]));
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
        }
        XCD3_STAMP(0)
        // x-part of this step (issued during the MFMA phase of the step before; the poll's drain covered it) ...
        float zin[RG];
        if (a.dbg & 1) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) zin[rg] = 0.1f;
        } else {
            XCD3_FOR_RG(adopt_dw<rg / 2, rg % 2, 0>(zin[rg]););
            // ... and now, behind a successful poll and ahead of the MFMAs that hide them: the outputs of the step before, the
            // x-part of the next step
            const size_t so = (size_t)t * B * XH, sz = (size_t)(t - 1) * B * XG4;       // outputs of step t-1: c_t, h_t, gates of t-1
            const int tz = t + 1 < a.t1 ? t + 1 : t;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const unsigned long long ma = o_have ? mall[rg] : 0ull, mg = o_have ? mg0[rg] : 0ull;
                store_dw_masked(ma, zstore[rg] + sz, o_a[rg]);                // activated gate, kept for BPTT
                store_dw_masked(mg, a.Cs + so + hofs[rg], cp[rg]);
                store_dw_masked(mg, a.Hs + so + hofs[rg], o_h[rg]);
            }
            XCD3_FOR_RG(issue_dw<rg / 2, rg % 2, 0>(zload[rg] + (size_t)tz * B * XG4););
        }
        // MFMAs: two accumulators per row group (K halves of the wave's slice), all of them issued alternately
        f32x4 acc[RG][2];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) { acc[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define XCD3_FWD_B(B_)                                                                              \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                            \
            _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) {                                     \
                acc[rg][0] = mfma44<B_>(av[2 * rg][e_], W[B_][e_], acc[rg][0]);                     \
                acc[rg][1] = mfma44<B_>(av[2 * rg + 1][e_], W[16 + B_][e_], acc[rg][1]);            \
            }
        XCD3_FWD_B(0) XCD3_FWD_B(1) XCD3_FWD_B(2) XCD3_FWD_B(3) XCD3_FWD_B(4) XCD3_FWD_B(5) XCD3_FWD_B(6) XCD3_FWD_B(7)
        XCD3_FWD_B(8) XCD3_FWD_B(9) XCD3_FWD_B(10) XCD3_FWD_B(11) XCD3_FWD_B(12) XCD3_FWD_B(13) XCD3_FWD_B(14) XCD3_FWD_B(15)
#undef XCD3_FWD_B
        XCD3_STAMP(1)
        // K-split partials meet in LDS: register i = row i, lane = packed column
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const f32x4 s = acc[rg][0] + acc[rg][1];
#pragma unroll
            for (int i = 0; i < 4; ++i) red[t & 1][rg][wave][i][lane] = s[i];
        }
        __syncthreads();
        if (s_fail) return;
        XCD3_STAMP(2)
        // cell update, one gate per lane and row group
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
#pragma clang fp contract(off)
            float zs = 0.0f;
            zs += red[t & 1][rg][0][wave][lane]; zs += red[t & 1][rg][1][wave][lane];
            zs += red[t & 1][rg][2][wave][lane]; zs += red[t & 1][rg][3][wave][lane];
            const float act_g = gate_activation(zin[rg] + zs, gate);
            // gate-0 lane of a unit: sigma(i) is its own value, tanh(j) / sigma(f) / sigma(o) sit 4 / 8 / 12 lanes up
            const float tj = row_up4<1>(act_g), sf = row_up4<2>(act_g), so = row_up4<3>(act_g);
            const float cn = cp[rg] * sf + act_g * tj;
            float hn = tanhf_(cn) * so;
            if (!actv[rg]) hn = 0.0f;                    // pad rows publish zeros so that every word of the buffer is written
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            store_l2_masked(wordlanes, hx_out[rg] + (size_t)(t + 1) * hx_step, hv);
            cp[rg] = cn; o_a[rg] = act_g; o_h[rg] = hn;
        }
        o_have = true;
        XCD3_STAMP(3)
    }
    // outputs of the last step
    if (o_have && !(a.dbg & 1)) {
        const size_t so = (size_t)a.t1 * B * XH, sz = (size_t)(a.t1 - 1) * B * XG4;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            store_dw_masked(mall[rg], zstore[rg] + sz, o_a[rg]);
            store_dw_masked(mg0[rg], a.Cs + so + hofs[rg], cp[rg]);
            store_dw_masked(mg0[rg], a.Hs + so + hofs[rg], o_h[rg]);
        }
    }
    if (PROF && lane == 0 && a.prof) {
        XCD3_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}

// ---------------------------------------------------------------- backward (reduce-scatter inside the XCD)
// Buffers exactly as k_lstm_bwd_xcd (same inbox layout).  Iteration t:
//   A  poll the inbox -> resets -> per-wave sums to LDS -> barrier
//   B  gate gradients on all four waves (wave w: row w; the gate-0 lane of a unit runs cell_backward, the four results go
//      back to the unit's four gate lanes): dz slice in A-register order to LDS -> barrier
//   C  row-major dz stores + the inputs of step t-1 (hidden by the MFMAs); MFMAs; the resets of A are awaited by COUNT
//      (they are the oldest operations in the queue: 4 RG operations were issued behind them), then the partials are published.
// PROF sums: [0] wait for the inbox, [1] resets + sums + barrier, [2] gate gradients + barrier, [3] stores / loads + MFMAs,
// [4] publish + rest.
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_xcd3(const LstmBwdXcdArgs a) {
    constexpr int NG = 4 / RG;                    // lane groups of a wave that read different producers of one row group
    constexpr int LPW = 2 * RG;                   // inbox words per lane: 8 producers x RG x 16 units / 64 lanes
    __shared__ __attribute__((aligned(16))) float psum[4 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float dzA[RG][64][4];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_fail = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    if (row0 >= B) return;

    f32x4 W[32];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
    }
    const int q = lane >> 4, gate = (lane >> 2) & 3, e = lane & 3;
    const int ul = 4 * q + e, unit = 16 * cu + ul;
    const unsigned long long g0lanes = 0x000F000F000F000Full;
    bool actv[RG];
    unsigned long long mall[RG];
    const float* gload[RG]; float* gstore[RG]; size_t hic[RG];
    float dcv[RG], n_ct[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        const int lrow = 4 * rg + wave, row = row0 + lrow;
        const bool act = lrow < rpx && row < B;
        const int rowc = row < B ? row : B - 1;
        actv[rg] = act;
        mall[rg] = uniform64(act ? ~0ull : 0ull);
        gload[rg] = a.Z + (size_t)rowc * XG4 + 64 * cu + lane;
        gstore[rg] = a.Z + (size_t)rowc * XG4 + 64 * cu + lane;
        hic[rg] = (size_t)rowc * XH + unit;
        dcv[rg] = (act && gate == 0) ? a.dc[hic[rg]] : 0.0f;
        n_ct[rg] = (a.t1 > a.t0) ? a.Cs[(size_t)a.t1 * B * XH + hic[rg]] : 0.0f;
    }
    const size_t slot_w = (size_t)NXCD * NCU * NCU * RG * 16;               // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    const size_t in_base = (((size_t)xcd * NCU + cu) * NCU + 8 * wave) * RG * 16 + lane;
    size_t out_ofs[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
        out_ofs[cg] = ((((size_t)xcd * NCU + 8 * wave + 4 * cg + (lane >> 4)) * NCU + cu) * RG) * 16 + (lane & 15);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;

    // inputs of the first update (step t1 - 1): slot 0 = this lane's activated gate, 1 = c_{t} (c_prev), 2 = dH
    if (a.t1 > a.t0) {
        const int t = a.t1 - 1;
        XCD3_FOR_RG(issue_dw<rg / 2, rg % 2, 0>(gload[rg] + (size_t)t * B * XG4);
                    issue_dw<rg / 2, rg % 2, 1>(a.Cs + (size_t)t * B * XH + hic[rg]);
                    issue_dw<rg / 2, rg % 2, 2>(a.dH + (size_t)t * B * XH + hic[rg]););
    }
    vm_wait<0>();
    // the compiler's own loads are waited for HERE: its waitcnt bookkeeping must be empty inside the loop
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(W[i]));
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) { asm volatile("" : "+v"(dcv[rg])); asm volatile("" : "+v"(n_ct[rg])); }

    for (int t = a.t1 - 1; t >= a.t0; --t) {
        XCD3_STAMP(4)
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
                if (__all(ok) || (a.dbg & 2)) break;
                __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_fail = 1;
            }
            XCD3_STAMP(0)
#pragma unroll
            for (int k = 0; k < LPW; ++k) {
                store_l2(in + k * 64, fill);
                wsum = (k == 0) ? v[0] : wsum + v[k];
            }
        } else {
            vm_wait<0>();                                  // no poll in front of the first step of a pass: its inputs come from the prologue
        }
        *reinterpret_cast<f32x4*>(&psum[(wave * 64 + lane) * 4]) = wsum;
        __syncthreads();
        if (s_fail) return;
        XCD3_STAMP(1)
        // ---- B: gate gradients.  The inputs were issued during the MFMA phase of the iteration before; the poll's drain covered them
        float ag[RG], n_cp[RG], n_dh[RG], dz[RG];
        if (a.dbg & 1) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) { ag[rg] = 0.3f; n_cp[rg] = 0.1f; n_dh[rg] = 0.01f; }
        } else {
            XCD3_FOR_RG(adopt_dw<rg / 2, rg % 2, 0>(ag[rg]); adopt_dw<rg / 2, rg % 2, 1>(n_cp[rg]); adopt_dw<rg / 2, rg % 2, 2>(n_dh[rg]););
        }
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            float dh_rec = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int grp = 0; grp < NG; ++grp)
                    dh_rec += psum[(w * 64 + (grp * RG + rg) * 16 + ul) * 4 + wave];
            // gate-0 lane of a unit: sigma(i) is its own value, tanh(j) / sigma(f) / sigma(o) sit 4 / 8 / 12 lanes up
            const float tj = row_up4<1>(ag[rg]), sf = row_up4<2>(ag[rg]), so = row_up4<3>(ag[rg]);
            const CellGrad cg = cell_backward(ag[rg], tj, sf, so, n_ct[rg], n_cp[rg], dcv[rg], n_dh[rg] + dh_rec);
            dcv[rg] = cg.dc_out;
            n_ct[rg] = n_cp[rg];                                     // c_t of the next update (step t-1) is this step's c_{t-1}
            // the four results travel to the unit's four gate lanes
            const float dj = row_down4<1>(cg.dj), df = row_down4<2>(cg.df), dg = row_down4<3>(cg.dg);
            dz[rg] = gate == 0 ? cg.di : gate == 1 ? dj : gate == 2 ? df : dg;
            // A-register order: local column k = 16 q + (4 gate + e) -> register v = q, block b = 4 gate + e, lane 4 b + row
            dzA[rg][4 * (lane & 15) + wave][q] = dz[rg];
        }
        __syncthreads();
        XCD3_STAMP(2)
        // ---- C: slow traffic at the head of the MFMA phase, MFMAs, publish
        if (!(a.dbg & 1)) {
            const int tq = t > a.t0 ? t - 1 : t;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) store_dw_masked(mall[rg], gstore[rg] + (size_t)t * B * XG4, dz[rg]);   // row-major dz for the weight-gradient GEMMs
            XCD3_FOR_RG(issue_dw<rg / 2, rg % 2, 0>(gload[rg] + (size_t)tq * B * XG4);
                        issue_dw<rg / 2, rg % 2, 1>(a.Cs + (size_t)tq * B * XH + hic[rg]);
                        issue_dw<rg / 2, rg % 2, 2>(a.dH + (size_t)tq * B * XH + hic[rg]););
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) {
            f32x4 av[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) av[rg] = *reinterpret_cast<const f32x4*>(&dzA[rg][lane][0]);
            f32x4 acc[RG][2];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) { acc[rg][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define XCD3_BWD_B(B_)                                                                                  \
            _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) {                                         \
                acc[rg][0] = mfma44<B_>(av[rg][v], W[4 * v + (B_ >> 2)][B_ & 3], acc[rg][0]);            \
                acc[rg][1] = mfma44<B_>(av[rg][v], W[16 + 4 * v + (B_ >> 2)][B_ & 3], acc[rg][1]);       \
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                XCD3_BWD_B(0) XCD3_BWD_B(1) XCD3_BWD_B(2) XCD3_BWD_B(3) XCD3_BWD_B(4) XCD3_BWD_B(5) XCD3_BWD_B(6) XCD3_BWD_B(7)
                XCD3_BWD_B(8) XCD3_BWD_B(9) XCD3_BWD_B(10) XCD3_BWD_B(11) XCD3_BWD_B(12) XCD3_BWD_B(13) XCD3_BWD_B(14) XCD3_BWD_B(15)
            }
#undef XCD3_BWD_B
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // MFMA result -> VMEM store data: wait states by hand (inline-asm stores)
            XCD3_STAMP(3)
            // the resets of part A must have landed before this block publishes anything (two-slot argument of k_lstm_bwd_rs);
            // behind them this wave issued RG dz stores and 3 RG input loads, which are NOT waited for
            if (a.dbg & 1) vm_wait<0>(); else vm_wait<4 * RG>();
            f32x4* out = inbox + (size_t)(t & 1) * slot_w;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                store_l2(out + out_ofs[0] + rg * 16, acc[rg][0]);
                store_l2(out + out_ofs[1] + rg * 16, acc[rg][1]);
            }
        }
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
        if (actv[rg] && gate == 0) a.dc[hic[rg]] = dcv[rg];
    if (PROF && lane == 0 && a.prof) {
        XCD3_STAMP(4)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}
#undef XCD3_STAMP
#undef XCD3_FOR_RG

// Kh [512][2048] (packed gate columns) -> the register images of the two XCD-local kernels:
//   fwd word i (= 16q + b), component e of (cu, w), lane l:  Kh[128w + 64q + 16(b/4) + 4(b%4) + e][64cu + l]
//   bwd word i, component e' of (cu, w), lane l, r = 4i + e' = 64cg + k:  Kh[128w + 64cg + l][64cu + k]
__global__ void k_repack_kh_xcd(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    const int total = NCU * 4 * 32 * 64;            // f32x4 words per copy
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
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

}  // namespace

static int xcd_row_groups(int B) {
    const int rpx = (B + NXCD - 1) / NXCD;
    const int rg = (rpx + 3) / 4;
    return rg == 3 ? 4 : rg;
}

bool lstm_xcd_supported(int B, int Hp) { return Hp == XH && B >= 1 && xcd_row_groups(B) <= 4; }

long long lstm_xcd_hx_floats(int B, int T) { return (long long)(T + 1) * NXCD * 4 * xcd_row_groups(B) * 2 * 64 * 4; }
long long lstm_xcd_inbox_floats(int B) { return 2LL * NXCD * NCU * NCU * xcd_row_groups(B) * 16 * 4; }
long long lstm_xcd_weight_floats() { return (long long)XH * XG4; }
// Rows per XCD that fill the row groups the 8-way split already pays for: the batch then sits on the first
// ceil(B / rows) XCDs and the others are free for another stream's GEMMs (B = 45: 8 rows on 6 XCDs instead of 6 on 8).
int lstm_xcd_packed_rows(int B) { return 4 * xcd_row_groups(B); }

hipError_t launch_repack_kh_xcd(hipStream_t s, const float* Kh, float* fwd, float* bwd) {
    hipLaunchKernelGGL(k_repack_kh_xcd, dim3(1024), dim3(256), 0, s, Kh, fwd, bwd);
    return hipGetLastError();
}

hipError_t launch_lstm_fwd_xcd(hipStream_t s, const LstmFwdXcdArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    const dim3 grid(NXCD * NCU), block(256);
    const int rg = xcd_row_groups(a.B);
    if (a.prof) {
        if (rg != 2) return hipErrorInvalidValue;
        if (a.pipe) hipLaunchKernelGGL((k_lstm_fwd_xcd3<2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_lstm_fwd_xcd<2, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.pipe && rg == 1) { hipLaunchKernelGGL((k_lstm_fwd_xcd3<1, false>), grid, block, 0, s, a); return hipGetLastError(); }
    if (a.pipe && rg == 2) { hipLaunchKernelGGL((k_lstm_fwd_xcd3<2, false>), grid, block, 0, s, a); return hipGetLastError(); }
    if (a.pipe && rg == 4) { hipLaunchKernelGGL((k_lstm_fwd_xcd3<4, false>), grid, block, 0, s, a); return hipGetLastError(); }
    switch (rg) {
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
    const int rg = xcd_row_groups(a.B);
    if (a.prof) {
        if (rg != 2) return hipErrorInvalidValue;
        if (a.pipe) hipLaunchKernelGGL((k_lstm_bwd_xcd3<2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_lstm_bwd_xcd<2, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.pipe && rg == 1) { hipLaunchKernelGGL((k_lstm_bwd_xcd3<1, false>), grid, block, 0, s, a); return hipGetLastError(); }
    if (a.pipe && rg == 2) { hipLaunchKernelGGL((k_lstm_bwd_xcd3<2, false>), grid, block, 0, s, a); return hipGetLastError(); }
    if (a.pipe && rg == 4) { hipLaunchKernelGGL((k_lstm_bwd_xcd3<4, false>), grid, block, 0, s, a); return hipGetLastError(); }
    switch (rg) {
        case 1: hipLaunchKernelGGL((k_lstm_bwd_xcd<1, false>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_lstm_bwd_xcd<2, false>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_lstm_bwd_xcd<4, false>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fsmg
