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
// Placement is discovered, not assumed: a block reads its XCC id and takes a ticket from that XCD's counter; the
// (xcd, ticket) pair is its role.  HIP promises nothing about block -> XCD placement, so a ticket >= 32 (an XCD that
// received more than its share) raises the time-out flag like any other failed wait and the caller falls back to the
// column-split kernels: placement decides speed, never results.
#include "fsmg_kernels.h"
#include "lstm_cell.h"

namespace fsmg {
namespace {

constexpr int XH = 512;            // padded hidden size these kernels are built for
constexpr int XG4 = 4 * XH;
constexpr int NXCD = 8, NCU = 32;  // XCDs per chip, CUs (= blocks) per XCD
constexpr int PLANE = 72;          // floats per gate plane of the LDS reduce buffer (64 + 8: the four gate planes of a writer land on different banks)

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

// ---------------------------------------------------------------- the software pipeline
// The RG row groups of an XCD are independent sequences, so a block runs them as RG interleaved chains:
//   waves 0..3   ("MFMA waves", one per SIMD): for every time step and row group in turn -- wait for the operand,
//                128 MFMAs on the resident weights, hand the result to that row group's cell wave through LDS;
//   wave 4 + rg  ("cell wave" of row group rg, co-resident with an MFMA wave on its SIMD: VALU and MFMA pipes issue side
//                by side): the cell arithmetic of its 64 (row, unit) pairs, the hand-off store, the outputs nobody
//                waits for.
// While one row group's h_t is being finished and crosses the L2, the MFMA pipes work on the other row group.  The
// waves of a block meet through LDS counters (release / acquire at workgroup scope), not s_barrier: a barrier would
// make the MFMA waves wait for a cell wave they have nothing to ask of.  First version (one chain, block barriers,
// cell update on waves 0..RG-1): 2.61 / 2.44 us per step forward / backward at B = 45, of which 2176 ticks MFMA,
// ~870 cell update and ~2400 waiting for the hand-off (tools/xcd_chain_bench.cpp phase profile).
// one count per WAVE: the LDS executes a wave's instructions in order, so lane 0's add follows every lane's stores
__device__ __forceinline__ void lds_signal(int* counter) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// spins until *counter >= target; false on time-out or when another wave of the block has failed
__device__ __forceinline__ bool lds_wait(const int* counter, int target, const int* s_fail, int spin_limit, bool nosleep = false) {
    for (int spins = 0;; ++spins) {
        if (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return true;
        if (__hip_atomic_load(s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || spins >= 64 * spin_limit) return false;
        if (!nosleep) __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void raise_timeout(int* err_flag, int* s_fail) {
    __hip_atomic_store(err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define XCD_STAMP(i) if (PROF) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; }

// ---------------------------------------------------------------- forward
// HX  [T+1][8 xcd][4 w][RG][2 q][64 lanes][4]: h in A-register order.  Lane 4b+i, component e of (w, rg, q) holds
//     h[row 4rg+i of the XCD][unit 128w + 64q + 16(b/4) + 4(b%4) + e]; CU c writes the 16 lanes 16(c%4) .. +15 of
//     (w = c/8, q = (c/4)%2) as 256 contiguous bytes.  Index 0 is the zero state, indices t0+1 .. t1 are pre-filled
//     with the "not written" pattern.
// KhX [32 cu][4 w][32][64 lanes][4]: register image of the weights, see k_repack_kh_xcd.
// PROF (diagnostic build): per (block, wave) sums of s_memtime ticks: MFMA waves [0] wait for h_t, [1] MFMAs + LDS
// hand-over; cell waves [2] wait for the partials, [3] cell update up to the hand-off store, [4] the other stores.
template <int RG, bool PROF>
__global__ __launch_bounds__(256 + 64 * RG) void k_lstm_fwd_xcd(const LstmFwdXcdArgs a) {
    // [time step parity][row group][MFMA wave][cell lane][gate].  Two copies: MFMA wave w needs h_{t+1} from the CUs
    // 8w .. 8w+7 only, so it can be a step ahead of its own block's cell wave -- never two (h_{t+2} needs every CU's
    // step t+1, which needs this block's h_{t+1})
    __shared__ __attribute__((aligned(16))) float red[2][RG][4][64 * 4];
    __shared__ int arrive[RG];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    if (tid < RG) arrive[tid] = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    const size_t hx_step = (size_t)NXCD * 4 * RG * 2 * 64;                  // f32x4 words per time index
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;

    if (wave < 4) {
        // ---------------- MFMA wave: K range 128 wave .. +127
        f32x4 W[32];
        {
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
        }
        const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)xcd * 4 + wave) * RG) * 2 * 64 + lane;
        // D layout of a 4x4 block: lane = 4*block + column, register = row.  Lane l = local packed column: unit block l/16,
        // gate (l/4)%4, unit%4 = l%4; the cell lane of (row i, unit) is 16 i + 4 (l/16) + l%4 and wants its four gates
        // as one 16-byte word
        float* const rbase = &red[0][0][wave][0] + ((lane >> 4) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3);
        for (int t = a.t0; t < a.t1; ++t) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                f32x4 av[2];
                bool got;
                if (a.flags & 4) {                                  // single-phase poll: both fragments every round
                    const f32x4* af = hx_in + (size_t)t * hx_step + rg * 2 * 64;
                    got = false;
                    for (int spins = 0; spins <= a.spin_limit; ++spins) {
                        av[0] = load_sc1(af); av[1] = load_sc1(af + 64);
                        drain_vmem();
                        asm volatile("" : "+v"(av[0]), "+v"(av[1]));
                        if (__all(frag_ready(av[0]) && frag_ready(av[1]))) { got = true; break; }
                        if (!(a.flags & 8)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                    }
                } else {
                    got = wait_fragments<2>(hx_in + (size_t)t * hx_step + rg * 2 * 64, av, a.spin_limit, a.err_flag);
                }
                if (!got) {
                    if (lane == 0) raise_timeout(a.err_flag, &s_fail);
                    return;
                }
                XCD_STAMP(0)
                if (PROF && a.prof && xcd == 0 && cu == 0 && lane == 0 && t >= 64 && t < 72)
                    a.prof[256 * 8 * 8 + (((t - 64) * 8 + wave) * 2 + rg) * 4 + 0] = __builtin_amdgcn_s_memtime();    // operand ready
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // two independent chains (q = 0, 1)
// the two chains alternate instruction by instruction: back-to-back MFMAs on ONE accumulator issue every ~12.4 ticks, on two every 8.5
#define XCD_FWD_B(B_)                                                                    \
                _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                          \
                    acc[0] = mfma44<B_>(av[0][e_], W[B_][e_], acc[0]);                      \
                    acc[1] = mfma44<B_>(av[1][e_], W[16 + B_][e_], acc[1]);                 \
                }
                if (!(a.flags & 32)) {                          // (experiment: bit 5 skips the MFMAs -> exchange-only timing)
                XCD_FWD_B(0) XCD_FWD_B(1) XCD_FWD_B(2) XCD_FWD_B(3) XCD_FWD_B(4) XCD_FWD_B(5) XCD_FWD_B(6) XCD_FWD_B(7)
                XCD_FWD_B(8) XCD_FWD_B(9) XCD_FWD_B(10) XCD_FWD_B(11) XCD_FWD_B(12) XCD_FWD_B(13) XCD_FWD_B(14) XCD_FWD_B(15)
                }
#undef XCD_FWD_B
                const f32x4 z = acc[0] + acc[1];
                float* rp = rbase + ((t & 1) * RG + rg) * (4 * 64 * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[64 * i] = z[i];            // cell lane 16 i + ... : 16 lanes x 4 floats further on
                lds_signal(&arrive[rg]);
                XCD_STAMP(1)
                if (PROF && a.prof && xcd == 0 && cu == 0 && lane == 0 && t >= 64 && t < 72)
                    a.prof[256 * 8 * 8 + (((t - 64) * 8 + wave) * 2 + rg) * 4 + 1] = __builtin_amdgcn_s_memtime();    // partials handed over
            }
        }
    } else {
        // ---------------- cell wave of row group rg: lane = 16 i + 4 bb + e -> row 4rg+i of the XCD, unit 16cu + 4bb + e
        const int rg = wave - 4;
        const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
        const int lrow = 4 * rg + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
        const bool act = lrow < rpx && row < B;
        float cp = act ? a.Cs[((size_t)a.t0 * B + row) * XH + unit] : 0.0f;
        f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) + ((((size_t)xcd * 4 + (cu >> 3)) * RG + rg) * 2 + ((cu >> 2) & 1)) * 64 +
                        16 * (cu & 3) + 4 * cbb + ci;
        for (int t = a.t0; t < a.t1; ++t) {
            float zin[4] = {0.f, 0.f, 0.f, 0.f};
            float* zp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            if (act && !(a.flags & 128)) {                // (experiment: bit 7 skips the x-part loads)
#pragma unroll
                for (int g = 0; g < 4; ++g) zin[g] = zp[4 * g];
            }
            if (!lds_wait(&arrive[rg], 4 * (t - a.t0 + 1), &s_fail, a.spin_limit, (a.flags & 16) != 0)) {
                if (lane == 0) raise_timeout(a.err_flag, &s_fail);
                return;
            }
            XCD_STAMP(2)
            if (PROF && a.prof && xcd == 0 && cu == 0 && lane == 0 && t >= 64 && t < 72)
                a.prof[256 * 8 * 8 + (((t - 64) * 8 + wave) * 2 + 0) * 4 + 0] = __builtin_amdgcn_s_memtime();          // partials seen
            float hn = 0.0f;
            CellOut co{};
            if (act) {
                const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[t & 1][rg][0][0]) + lane;
                const f32x4 r0 = rsrc[0], r1 = rsrc[64], r2 = rsrc[128], r3 = rsrc[192];
                float zg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float zs = 0.0f;
                    zs += r0[g]; zs += r1[g]; zs += r2[g]; zs += r3[g];
                    zg[g] = zin[g] + zs;
                }
                if (a.flags & 64) { co.h = zg[0] * 0.001f; co.c = cp; }      // (experiment: bit 6 skips the cell arithmetic)
                else co = cell_forward(zg, cp);
                hn = co.h; cp = co.c;
            }
            // the four units of a quad form one 16-byte word of the hand-off; pad rows publish zeros so that every word
            // of the buffer is written and the readers' test terminates
            f32x4 hv;
            hv[0] = quad_bcast<0>(hn); hv[1] = quad_bcast<1>(hn); hv[2] = quad_bcast<2>(hn); hv[3] = quad_bcast<3>(hn);
            if (ce == 0) { if (a.flags & 2) store_sc1(hx_out + (size_t)(t + 1) * hx_step, hv); else store_l2(hx_out + (size_t)(t + 1) * hx_step, hv); }
            if (a.flags & 1) drain_vmem();
            XCD_STAMP(3)
            if (PROF && a.prof && xcd == 0 && cu == 0 && lane == 0 && t >= 64 && t < 72)
                a.prof[256 * 8 * 8 + (((t - 64) * 8 + wave) * 2 + 0) * 4 + 1] = __builtin_amdgcn_s_memtime();          // hand-off store issued
            if (act && !(a.flags & 256)) {                // (experiment: bit 8 skips the output stores)
                a.Cs[((size_t)(t + 1) * B + row) * XH + unit] = cp;
                a.Hs[((size_t)(t + 1) * B + row) * XH + unit] = hn;
                zp[0] = co.si; zp[4] = co.tj; zp[8] = co.sf; zp[12] = co.so;       // activated gates kept for BPTT
            }
            XCD_STAMP(4)
        }
    }
    if (PROF && lane == 0 && a.prof)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 8 + wave) * 8 + i] = pacc[i];
}

// ---------------------------------------------------------------- backward (reduce-scatter inside the XCD)
// Block (xcd, cu) keeps the dz of its 16 units to itself; what travels is dh.  Chain of row group rg, iteration t
// (descending):
//   cell wave:  consume the 32 partials of dh_t for its units (inbox words = 4 rows of one unit), put the fill pattern
//               back, sum in a fixed order; dh -> gate gradients -> row-major dz for the GEMMs + the 4 rows x 64
//               columns dz slice in A-register order in LDS; drain the resets; signal;
//   MFMA waves: wave w multiplies the slice with its resident 64 x 128 slice of K_h^T (destination units 128w .. +127)
//               and stores the 4x4-block results straight from the MFMA registers into the destinations' inboxes
//               (plain stores: same XCD).
// Two inbox slots suffice: nobody can overwrite a slot before every reader of its previous content has put the fill
// pattern back (and drained that store), because progress of every block depends on every other block's publish.
// inbox [2 slots][8 xcd][32 dest][RG][32 producer][16 units][4 rows].
// PROF: MFMA waves [0] wait for dz, [1] MFMAs + stores; cell waves [2] wait for the inbox, [3] sums + cell + LDS, [4] rest.
template <int RG, bool PROF>
__global__ __launch_bounds__(256 + 64 * RG) void k_lstm_bwd_xcd(const LstmBwdXcdArgs a) {
    __shared__ __attribute__((aligned(16))) float psum[RG][64 * 4];
    // two copies by time step parity: the cell wave can be a step ahead of the block's MFMA waves other than wave cu/8
    // (the only one whose publish it waits for), never two
    __shared__ __attribute__((aligned(16))) float dzA[2][RG][64][4];
    __shared__ int ready[RG];
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    if (tid < RG) ready[tid] = 0;
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int xcd = role.xcd, cu = role.cu;
    const int B = a.B;
    const int rpx = (B + NXCD - 1) / NXCD, row0 = xcd * rpx;
    const size_t slot_w = (size_t)NXCD * NCU * RG * NCU * 16;               // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;

    if (wave < 4) {
        // ---------------- MFMA wave: destination units 128 wave .. +127 (two column groups of 64)
        f32x4 W[32];          // component e' of word i = weight register 4i + e' = (cg = /64, k = %64): Kh[128w + 64cg + lane][64cu + k]
        {
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * 32) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) W[i] = wp[i * 64];
        }
        // lane l of column group cg -> destination CU 8w + 4cg + l/16, word (dest, rg, producer = cu, l%16)
        size_t out_ofs[2];
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
            out_ofs[cg] = ((((size_t)xcd * NCU + 8 * wave + 4 * cg + (lane >> 4)) * RG) * NCU + cu) * 16 + (lane & 15);
        int n = 0;
        for (int t = a.t1 - 1; t >= a.t0; --t) {
            ++n;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                if (!lds_wait(&ready[rg], n, &s_fail, a.spin_limit)) {
                    if (lane == 0) raise_timeout(a.err_flag, &s_fail);
                    return;
                }
                XCD_STAMP(0)
                if (t == 0) continue;                                  // nothing to hand on below the first step
                const f32x4 av = *reinterpret_cast<const f32x4*>(&dzA[t & 1][rg][lane][0]);
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                // k = 16 v + b: A register av[v], broadcast block b; weight register cg*64 + k = word cg*16 + 4v + b/4, component b%4
#define XCD_BWD_B(B_)                                                                                   \
                acc[0] = mfma44<B_>(av[v], W[4 * v + (B_ >> 2)][B_ & 3], acc[0]);                        \
                acc[1] = mfma44<B_>(av[v], W[16 + 4 * v + (B_ >> 2)][B_ & 3], acc[1]);
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
                f32x4* out = inbox + (size_t)(t & 1) * slot_w + (size_t)rg * NCU * 16;
                store_l2(out + out_ofs[0], acc[0]);
                store_l2(out + out_ofs[1], acc[1]);
                XCD_STAMP(1)
            }
        }
    } else {
        // ---------------- cell wave of row group rg
        const int rg = wave - 4;
        const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
        const int lrow = 4 * rg + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
        const bool act = lrow < rpx && row < B;
        const size_t hi = (size_t)row * XH + unit;
        float dcv = act ? a.dc[hi] : 0.0f;
        // this block's words of row group rg: [32 producers][16 units]; load k covers producers 4k .. 4k+3 (lane/16), unit lane%16
        const size_t in_base = (((size_t)xcd * NCU + cu) * RG + rg) * NCU * 16 + lane;
        const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
        float* const ps = &psum[rg][0];
        for (int t = a.t1 - 1; t >= a.t0; --t) {
            float si = 0.f, tj = 0.f, sf = 0.f, so = 0.f, ct = 0.f, cpv = 0.f, dht = 0.f;
            float* gp = a.Z + ((size_t)t * B + row) * XG4 + 64 * cu + 16 * cbb + ce;
            if (act) {
                si = gp[0]; tj = gp[4]; sf = gp[8]; so = gp[12];
                ct = a.Cs[(size_t)(t + 1) * B * XH + hi]; cpv = a.Cs[(size_t)t * B * XH + hi];
                dht = a.dH[(size_t)t * B * XH + hi];
            }
            float dh_rec = 0.0f;
            if (t + 1 < a.T) {
                f32x4* in = inbox + (size_t)((t + 1) & 1) * slot_w + in_base;
                f32x4 v[8];
                bool fail = false;
                for (int spins = 0;; ++spins) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = load_sc1(in + k * 64);
                    drain_vmem();
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { asm volatile("" : "+v"(v[k])); ok &= frag_ready(v[k]); }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                    if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
                }
                if (fail) {
                    if (lane == 0) raise_timeout(a.err_flag, &s_fail);
                    return;
                }
                XCD_STAMP(2)
                f32x4 wsum = v[0];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    store_l2(in + k * 64, fill);
                    if (k) wsum = wsum + v[k];
                }
                // lane (group = lane/16, unit = lane%16) holds the sum over producers = group mod 4 of rows 0..3; the cell
                // lane (i, bb, e) adds the four groups of unit 4bb+e, row i.  Same wave: LDS order is program order.
                *reinterpret_cast<f32x4*>(&ps[lane * 4]) = wsum;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) dh_rec += ps[(grp * 16 + 4 * cbb + ce) * 4 + ci];
            } else {
                XCD_STAMP(2)
            }
            float di = 0.f, dj = 0.f, df = 0.f, dg = 0.f;
            if (act) {
                const CellGrad cg = cell_backward(si, tj, sf, so, ct, cpv, dcv, dht + dh_rec);
                di = cg.di; dj = cg.dj; df = cg.df; dg = cg.dg;
                dcv = cg.dc_out;
            }
            // A-register order: local column k = 16bb + 4g + e -> register v = bb, block b = 4g + e, lane 4b + i
            float* f = &dzA[t & 1][rg][4 * ce + ci][cbb];
            f[0] = di; f[16 * 4] = dj; f[32 * 4] = df; f[48 * 4] = dg;
            drain_vmem();                                                 // the resets have landed before anything is published
            lds_signal(&ready[rg]);
            XCD_STAMP(3)
            if (act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }   // row-major dz for the weight-gradient GEMMs
            XCD_STAMP(4)
        }
        if (act) a.dc[hi] = dcv;
    }
    if (PROF && lane == 0 && a.prof)
        for (int i = 0; i < 5; ++i) a.prof[((size_t)(xcd * NCU + cu) * 8 + wave) * 8 + i] = pacc[i];
}
#undef XCD_STAMP

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

hipError_t launch_repack_kh_xcd(hipStream_t s, const float* Kh, float* fwd, float* bwd) {
    hipLaunchKernelGGL(k_repack_kh_xcd, dim3(1024), dim3(256), 0, s, Kh, fwd, bwd);
    return hipGetLastError();
}

hipError_t launch_lstm_fwd_xcd(hipStream_t s, const LstmFwdXcdArgs& a) {
    if (a.t1 <= a.t0) return hipSuccess;
    const dim3 grid(NXCD * NCU), block(256 + 64 * xcd_row_groups(a.B));
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
    const dim3 grid(NXCD * NCU), block(256 + 64 * xcd_row_groups(a.B));
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
