// XCD-PAIR-local recurrence for hidden size 1024 on the bf16 matrix pipe (round 6; cfg-C / cfg-E of BASELINE.json).
// Textually part of lstm_xcd.hip (included inside its anonymous namespace, behind the fp32 pair kernels whose roles, inbox
// layout and constants it shares); reference graph: src/models/lstm_baseline.py:44-55 (MultiRNNCell of BasicLSTMCell).
//
// Why: the fp32 pair kernels above are bound by what a pair multiplies per row group (256 `4x4x1` MFMAs per
// wave and row group, forward 0.9 + 1.7 x RG us per step, backward 0.75 + 2.2 x RG) -- at cfg-C's 12 rows per pair three row
// groups, 6144 matrix-pipe cycles per step.  The same product as six bf16 products of the exact three-way split (the arithmetic
// of gemm.hip and of k_lstm_*_xcd16) is 192 `v_mfma_f32_16x16x32_bf16` per wave = 3072 cycles for ANY row count up to 16.
//
// Where the weights live: K_h as three bf16 planes is 24 MiB per copy; an XCD pair has 32 MiB of registers.  Planes 0 and 1
// (five of the six terms read them) take 256 registers per lane -- loaded into the accumulation half; hipcc then spreads them over
// that half and what the arch half has left (456-476 of 512 registers in all, nothing spilled: tools/check_xcd_asm.py); plane 2
// (read by ONE term) sits in LDS, 128 KiB per CU, each wave reading only its own 32 KiB (one conflict-free ds_read_b128 per
// operand, 32 per step and wave, hidden under the MFMAs; backward: read one tile group ahead).  The rest of the registers holds
// the 24 hand-off fragments of a step (96), the accumulators and the cell update.
// What a step costs and which of the variants (XCD_PROBE, XCD_STREAM, XCD_LOCAL_PLAIN, XCD_LATE_DRAIN) bought what: DESIGN.md 11.8.
//
//   forward : CU c of a pair (64 CUs) owns hidden units 16 c ... 16 c + 15 = packed gate columns 64 c ... (four 16-column
//             tiles); wave w the K range 256 w ... (eight k steps of 32).  A operand = rows of h, as in k_lstm_fwd_xcd16.
//   HX16P   : [T+1][4 pairs][4 w][3 planes][8 k steps][4 RG rows][4 k groups] 16-byte words (8 units of one row and plane);
//             hand-off stores are write-through (half of the consumers sit on the other XCD), loads sc1.
//   backward: dz slice (rows x the CU's 64 gate columns, two k steps) split by the cell threads into LDS; wave w holds K_h^T
//             for destination units 256 w ... (sixteen 16-unit tiles = sixteen destination CUs); the D registers of tile nt
//             are the inbox words (dest 16 w + nt, row group l / 16, producer, unit l % 16) of k_lstm_bwd_pair's inbox.
// The K-split partials of a step meet in LDS: two buffers by step parity up to three row groups, one buffer and a second barrier at four
// (two do not fit beside plane 2 there).
constexpr int P16KS = PKW / 32;                 // k steps of a wave's K range (forward)
constexpr int P16NF = 3 * P16KS;                // hand-off fragments per lane and step
constexpr int P16W = 3 * P16KS * 4;             // 16-byte weight words per lane (image), either direction
constexpr int HXW16P = 4 * P16NF * 4;           // 16-byte words per row of one pair and time index

template <int OFS>
__device__ __forceinline__ void store_sc1_ofs(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1\n\ts_nop 1" : : "v"(p), "v"(v), "n"(OFS) : "memory");
}
template <int N>
__device__ __forceinline__ bool frags16_ready_n(const f32x4 (&av)[N]) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = pk_max_u16(m, __float_as_uint(av[j][e]));
    return (m & 0xffffu) != 0xffffu && (m >> 16) != 0xffffu;
}

#define P16_TERMS_REG(DO) DO(1, 1) DO(1, 0) DO(0, 1) DO(0, 0)
// PROF (tools/xcd_chain_bench): per (block, wave) sums of s_memtime ticks, phases as in k_lstm_fwd_xcd16 / k_lstm_bwd_xcd16
#define P16_STAMP(i) if (PROF) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; }
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_pair16(const LstmFwdXcdArgs a) {
    constexpr int HXR = 4 * RG;                  // rows per pair the hand-off buffer is laid out for
    __shared__ __attribute__((aligned(16))) f32x4 w2[4][P16KS * 4][64];          // plane 2 of the weights: [wave][k step, column tile][lane]
    constexpr bool DB = RG <= 3;                 // the K-split partials in two buffers by step parity where they fit beside plane 2 (24 KiB), else one + a barrier
    __shared__ __attribute__((aligned(16))) float red[DB ? 2 : 1][4][HXR * 16 * 4];          // [step parity][wave][row][unit][gate]
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
    const int rgc = min((rpx + 3) >> 2, RG);     // waves with cell threads: wave rg owns rows 4 rg ... 4 rg + 3

    xbf16x8 W[2][P16KS][4];                      // [plane][k step][column tile]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhX) + ((size_t)(cu * 4 + wave) * P16W) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < P16KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) w2[wave][ks * 4 + nt][lane] = wp[((2 * P16KS + ks) * 4 + nt) * 64];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < P16KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 wv = wp[((pl * P16KS + ks) * 4 + nt) * 64];
                    asm volatile("" : "+a"(wv));          // planes 0 and 1 fill the accumulation half of the register file
                    W[pl][ks][nt] = as_bf16x8(wv);
                }
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < rgc;
    const bool act = cellw && lrow < rpx && row < B;
    const bool pub = cellw && lrow < rpx;         // rows past B on the last pair publish zeros: the readers load every row < rpx
    // XCD-partitioned step: a GEMM on the XCDs this launch leaves free draws row tiles as the time steps complete -- progress[t] and
    // the write-through stores of the row-major h exactly as in k_lstm_fwd_xcd16 (every wave drains its memory queue in the probe of a
    // later step, that step's barrier orders the four waves, thread 0 publishes `lag` steps behind)
    const bool wt = a.progress != nullptr;
    const int lag = ((a.variant & XCD_DEFER_OUTPUTS) ? 2 : 1) + (a.progress_lag & 7);
    const int pk = a.progress_every > 0 ? a.progress_every : 1;
    float cp = act ? a.Cs[((size_t)a.t0 * B + row) * PH + unit] : 0.0f;
    // A operand: lane = (row l % 16, k group l / 16).  The lanes of pad rows load row 0 again (same cache lines, no extra traffic; no
    // divergent loads for the compiler to merge): their rows of D are never read
    const int arow = (lane & 15) < rpx ? (lane & 15) : 0, akg = lane >> 4;
    const size_t hx_step = (size_t)HXR * PGRP * HXW16P;                     // 16-byte words per time index
    const f32x4* hx_in = reinterpret_cast<const f32x4*>(a.HX) + (((size_t)grp * 4 + wave) * P16NF) * (HXR * 4) + arow * 4 + akg;
    // unit u = 16 cu + 4 cbb + ce -> w = u / 256, k step = u % 256 / 32, k group = u % 32 / 8, position u % 8
    f32x4* hx_out = reinterpret_cast<f32x4*>(a.HX) +
        ((((size_t)grp * 4 + (cu >> 4)) * P16NF + ((cu & 15) >> 1)) * (HXR * 4) + lrow * 4 + 2 * (cu & 1) + (cbb >> 1));
    const bool stl = pub && ce == 0 && (cbb & 1) == 0;      // the lane that stores the eight units 8 (cbb / 2) ... of its row
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
    f32x4 av[P16NF];                              // [plane][k step]
    const f32x4* const w2p = &w2[wave][0][lane];
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;      // [5] probe rounds, [6] full rounds, [7] ticks in the probe

    for (int t = a.t0; t < a.t1; ++t) {
        P16_STAMP(4)
        const f32x4* af = hx_in + (size_t)t * hx_step;
        // XCD_PROBE: plane 2 (stored last) of k step 2 (row % 4) + (k group % 2) -- the lanes of a wave cover every (producer CU, cell
        // wave) of the wave's K range once, ONE load per lane and round instead of 24 (the polls of the waves that are early no longer
        // fill the CU's memory pipeline in front of the cell waves' stores).  Whatever the probe says, the fragments themselves decide.
        bool stream = false;
        if (a.variant & XCD_PROBE) {
            // A wave without cell threads reaches its probe a whole cell update before anybody can have stored anything, and its probe rounds
            // (4 per step) stood in the way of the cell waves' stores and off the grid on which the data arrives: it sleeps first, XCD_PROBE_DELAY
            // x 512 clocks (bits 13-15 of the variant; 2048 clocks: 5.46 -> 5.00 us per step at B = 45, profiles/r06_pair16_probe6_delay.log).
            // (The same in front of the backward kernel's first inbox poll: 5.60 -> 5.62 ... 6.10; 256 ... 1024 clocks for the cell waves too:
            // 4.98 -> 4.97 / 5.05 / 5.14 / 5.15: not taken.)
            if (!cellw && t > a.t0) { for (int q = (a.variant / XCD_PROBE_DELAY) & 7; q > 0; --q) __builtin_amdgcn_s_sleep(8); }
            const f32x4* sp = af + (2 * P16KS + 2 * (arow & 3) + (akg & 1)) * (HXR * 4);
            for (int spins = 0; spins < a.spin_limit; ++spins) {
                f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
                sv = load_sc1(sp); drain_vmem();
                if (PROF) ++pacc[5];
                asm volatile("" : "+v"(sv));
                const unsigned m = pk_max_u16(pk_max_u16(__float_as_uint(sv[0]), __float_as_uint(sv[1])), pk_max_u16(__float_as_uint(sv[2]), __float_as_uint(sv[3])));
                if (__all((m & 0xffffu) != 0xffffu && (m >> 16) != 0xffffu)) { stream = (a.variant & XCD_STREAM) != 0; break; }
                if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                if ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) break;
            }
            if (PROF) pacc[7] += __builtin_amdgcn_s_memtime() - plast;
        }
        float zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { zin[g] = zq[0][g]; zq[0][g] = zq[1][g]; zq[1][g] = 0.0f; }
        float* zp = a.Z + ((size_t)t * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        f32x4 acc[4];
        for (;;) {
            if (PROF) ++pacc[6];
            if (stream) {
                // XCD_STREAM: behind a successful probe the 24 fragments are ordinary loads -- the compiler counts them (s_waitcnt vmcnt(N) in
                // front of the first MFMA that reads each) so the matrix pipe starts on k step 0 while the rest of the 72 KiB a CU fetches per
                // step is still on its way (a CU takes in 64 bytes per clock: the fetch alone is as long as the MFMAs).  Checked AFTER the
                // products: a fragment that was not there yet shows the fill pattern, and the step is redone behind the sc1 poll.
                // (hipcc clusters loads by base register: without the scheduling fences the first MFMA waits for 17 of the 28 loads in flight)
#define P16_LD(J) av[J] = af[((J) >> 2) * (4 * HXR * 4) + ((J) & 3) * (HXR * 4)];
#define P16_LD3(KS) P16_LD(KS) P16_LD(8 + KS) P16_LD(16 + KS) __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_sched_barrier(0);
                P16_LD3(0) P16_LD3(1) P16_LD3(2) P16_LD3(3) P16_LD3(4) P16_LD3(5) P16_LD3(6) P16_LD3(7)
#undef P16_LD3
#undef P16_LD
            } else {
                bool fail = false;
                for (int spins = 0;; ++spins) {
#define P16_LD(J) av[J] = load_sc1_ofs<((J) & 3) * HXR * 64>(af + ((J) >> 2) * (4 * HXR * 4));      /* (plane, k step) stride: HXR x 64 bytes */
                    P16_LD(0) P16_LD(1) P16_LD(2) P16_LD(3) P16_LD(4) P16_LD(5) P16_LD(6) P16_LD(7) P16_LD(8) P16_LD(9) P16_LD(10) P16_LD(11)
                    P16_LD(12) P16_LD(13) P16_LD(14) P16_LD(15) P16_LD(16) P16_LD(17) P16_LD(18) P16_LD(19) P16_LD(20) P16_LD(21) P16_LD(22) P16_LD(23)
#undef P16_LD
                    drain_vmem();
#pragma unroll
                    for (int j = 0; j < P16NF; ++j) asm volatile("" : "+v"(av[j]));
                    const bool ok = frags16_ready_n<P16NF>(av);
                    if (__all(ok)) break;
                    if (!(a.variant & XCD_NO_POLL_SLEEP)) __builtin_amdgcn_s_sleep(FSMG_POLL_SLEEP);
                    if (spins >= a.spin_limit || ((spins & 255) == 255 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) { fail = true; break; }
                }
                if (fail && lane == 0) {
                    __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_fail = 1;
                }
            }
            P16_STAMP(0)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < P16KS; ++ks) {      // term order of gemm.hip / k_lstm_fwd_xcd16: smallest products first
                xbf16x8 b2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b2[nt] = as_bf16x8(w2p[(ks * 4 + nt) * 64]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[2 * P16KS + ks]), W[0][ks][nt], acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[ks]), b2[nt], acc[nt], 0, 0, 0);
#define P16_FWD(PA, PB)                                                                                         \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                \
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(av[(PA) * P16KS + ks]), W[PB][ks][nt], acc[nt], 0, 0, 0);
                P16_TERMS_REG(P16_FWD)
#undef P16_FWD
            }
            if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {       // outputs of the step before: behind the fetch and the MFMAs' waits for it
                a.Cs[((size_t)t * B + row) * PH + unit] = o_c;
                store_h_row(a.Hs + ((size_t)t * B + row) * PH + unit, o_hh, wt);
                float* zo = a.Z + ((size_t)(t - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
                zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
            }
            o_have = false;
            if (!stream) break;
            const bool ok = frags16_ready_n<P16NF>(av);
            if (__all(ok)) break;
            stream = false;                            // a fragment was not there yet: redo the step behind the sc1 poll
        }
        if (act && t + 2 < a.t1) {                     // x-part of the update after next (behind the fetch in the memory queue)
            const float* zn = zp + 2 * (size_t)B * PG4;
#pragma unroll
            for (int g = 0; g < 4; ++g) zq[1][g] = zn[4 * g];
        }
        if (PROF) { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); acc[0][0] += 0.0f; }
        P16_STAMP(1)
        if (PROF && (a.variant & 8192)) {          // diagnostics: the same 24 fragments again, now surely present: [5] sc1 loads, [7] plain loads (ticks)
            f32x4 bv[P16NF];
            unsigned long long q0 = __builtin_amdgcn_s_memtime();
#define P16_LD(J) bv[J] = load_sc1_ofs<((J) & 3) * HXR * 64>(af + ((J) >> 2) * (4 * HXR * 4));
            P16_LD(0) P16_LD(1) P16_LD(2) P16_LD(3) P16_LD(4) P16_LD(5) P16_LD(6) P16_LD(7) P16_LD(8) P16_LD(9) P16_LD(10) P16_LD(11)
            P16_LD(12) P16_LD(13) P16_LD(14) P16_LD(15) P16_LD(16) P16_LD(17) P16_LD(18) P16_LD(19) P16_LD(20) P16_LD(21) P16_LD(22) P16_LD(23)
#undef P16_LD
            drain_vmem();
#pragma unroll
            for (int j = 0; j < P16NF; ++j) asm volatile("" : : "v"(bv[j]));
            unsigned long long q1 = __builtin_amdgcn_s_memtime();
            pacc[5] += q1 - q0;
#pragma unroll
            for (int j = 0; j < P16NF; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bv[j]) : "v"(af + (j >> 2) * (4 * HXR * 4) + (j & 3) * (HXR * 4)) : "memory");
            drain_vmem();
#pragma unroll
            for (int j = 0; j < P16NF; ++j) asm volatile("" : : "v"(bv[j]));
            pacc[7] += __builtin_amdgcn_s_memtime() - q1;
            plast = __builtin_amdgcn_s_memtime();
        }
        if (akg < rgc) {                           // D: lane = (column 4 g + e of the tile, rows 4 akg ... + 3)
            float* rp = &red[DB ? (t & 1) : 0][wave][0] + ((4 * akg) * 16 + (lane & 3)) * 4 + ((lane >> 2) & 3);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rp[(r * 16 + 4 * nt) * 4] = acc[nt][r];
        }
        __syncthreads();
        if (s_fail) return;
        if (wt && tid == 0 && t - lag >= a.t0 && (t - lag) % pk == pk - 1)
            __hip_atomic_fetch_add(a.progress + (t - lag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        P16_STAMP(2)

        // The K-split partials are read HERE, by every cell thread.  Up to three row groups they sit in two buffers by step parity (as in
        // k_lstm_fwd_xcd16: a wave that is one step ahead writes the other one, and nobody can be two ahead -- the step's barrier needs
        // every wave).  Four row groups leave room for one buffer beside plane 2, and a second barrier hands it back: a wave's poll covers
        // the producers of ITS K range only, so "behind my next poll" would not mean "this CU's cell waves have read their partials" (in
        // practice they read them a microsecond before anybody can have new ones; the barrier makes it a guarantee for ~200 clocks a step).
        f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0;
        if (act) {
            const f32x4* rsrc = reinterpret_cast<const f32x4*>(&red[DB ? (t & 1) : 0][0][0]) + lrow * 16 + 4 * cbb + ce;
            r0 = rsrc[0]; r1 = rsrc[HXR * 16]; r2 = rsrc[2 * HXR * 16]; r3 = rsrc[3 * HXR * 16];
        }
        if (!DB) __syncthreads();
        if (cellw) {
            float hn = 0.0f, g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f;
            if (act) {
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
                const f32x4 w0 = pack8_bf16(hp[0]), w1 = pack8_bf16(hp[1]), w2v = pack8_bf16(hp[2]);
                if (stl) {
                    f32x4* o = hx_out + (size_t)(t + 1) * hx_step;
                    store_sc1_ofs<0>(o, w0); store_sc1_ofs<0>(o + P16KS * HXR * 4, w1); store_sc1_ofs<0>(o + 2 * P16KS * HXR * 4, w2v);
                }
            }
            P16_STAMP(3)
            if (a.variant & XCD_DEFER_OUTPUTS) {
                o_c = cp; o_hh = hn; o_g[0] = g_si; o_g[1] = g_tj; o_g[2] = g_sf; o_g[3] = g_so; o_have = true;
            } else if (act) {
                a.Cs[((size_t)(t + 1) * B + row) * PH + unit] = cp;
                store_h_row(a.Hs + ((size_t)(t + 1) * B + row) * PH + unit, hn, wt);
                zp[0] = g_si; zp[4] = g_tj; zp[8] = g_sf; zp[12] = g_so;
            }
        }
    }
    if ((a.variant & XCD_DEFER_OUTPUTS) && o_have && act) {
        a.Cs[((size_t)a.t1 * B + row) * PH + unit] = o_c;
        store_h_row(a.Hs + ((size_t)a.t1 * B + row) * PH + unit, o_hh, wt);
        float* zo = a.Z + ((size_t)(a.t1 - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        zo[0] = o_g[0]; zo[4] = o_g[1]; zo[8] = o_g[2]; zo[12] = o_g[3];
    }
    if (wt) {                                     // the last `lag` steps: drain, barrier, publish
        drain_vmem();
        __syncthreads();
        if (tid == 0)
            for (int tp = max(a.t0, a.t1 - lag); tp < a.t1; ++tp)
                if (tp % pk == pk - 1 || tp == a.t1 - 1) __hip_atomic_fetch_add(a.progress + tp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PROF && lane == 0 && a.prof) {
        P16_STAMP(4)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)(role.xcd * NCU + role.cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}

// inbox exactly as k_lstm_bwd_pair: [2 slots][4 pairs][64 dest][RG][64 producers][16 units][4 rows]
template <int RG, bool PROF>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_pair16(const LstmBwdXcdArgs a) {
    constexpr int LPR = 4;                        // inbox words per lane and row group: 16 producers x 16 units / 64 lanes
    __shared__ __attribute__((aligned(16))) f32x4 w2[4][2 * 16][64];             // plane 2 of K_h^T: [wave][k step, destination tile][lane]
    __shared__ __attribute__((aligned(16))) float psum[RG][4][64][4];
    __shared__ __attribute__((aligned(16))) unsigned char dzA[3][2][64][16];      // [plane][k step][lane = (k group, row)][8 bf16]
    __shared__ int s_role[2];
    __shared__ int s_fail;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_fail = 0;
    for (int i = tid; i < 3 * 2 * 64 * 4; i += 256) reinterpret_cast<unsigned*>(&dzA[0][0][0][0])[i] = 0u;     // rows >= 4 RG stay zero
    Role role;
    if (!take_role(a.tickets, a.err_flag, s_role, role)) return;
    const int grp = role.xcd / PNX, cu = (role.xcd % PNX) * NCU + role.cu;
    const int B = a.B;
    const int rpx = a.rpx > 0 ? a.rpx : (B + PGRP - 1) / PGRP, row0 = grp * rpx;
    if (row0 >= B) return;

    xbf16x8 W[2][2][16];                          // [plane][k step][destination tile]: Kh[256 w + 16 nt + l % 16][64 cu + 32 ks + 8 (l / 16) + j]
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.KhXb) + ((size_t)(cu * 4 + wave) * P16W) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) w2[wave][ks * 16 + nt][lane] = wp[((2 * 2 + ks) * 16 + nt) * 64];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) {
                    f32x4 wv = wp[((pl * 2 + ks) * 16 + nt) * 64];
                    asm volatile("" : "+a"(wv));
                    W[pl][ks][nt] = as_bf16x8(wv);
                }
    }
    const int ci = lane >> 4, cbb = (lane >> 2) & 3, ce = lane & 3;
    const int lrow = 4 * wave + ci, row = row0 + lrow, unit = 16 * cu + 4 * cbb + ce;
    const bool cellw = wave < RG;
    const bool act = cellw && lrow < rpx && row < B;
    const size_t hi = (size_t)row * PH + unit;
    float dcv = act ? a.dc[hi] : 0.0f;
    const size_t slot_w = (size_t)PGRP * PCU * RG * PCU * 16;                // f32x4 words per slot
    f32x4* const inbox = reinterpret_cast<f32x4*>(a.inbox);
    // consumer: (dest = cu, row group rg): 64 producers x 16 units; wave w takes producers 16 w ... 16 w + 15, 4 words per lane
    const size_t in_base = (((size_t)grp * PCU + cu) * RG) * PCU * 16 + (size_t)(16 * wave) * 16 + lane;
    // producer: tile nt of wave w -> destination 16 w + nt, word (dest, row group l / 16, producer = cu, unit l % 16)
    const size_t out_ofs = (((size_t)grp * PCU + 16 * wave) * RG + (lane >> 4)) * PCU * 16 + (size_t)cu * 16 + (lane & 15);
    const bool outl = (lane >> 4) < RG;
    // wave w consumes the partials of producers 16 w ... and produces for destinations 16 w ...: XCD (w / 2) of the pair either way
    const bool loc = (a.variant & XCD_LOCAL_PLAIN) && __builtin_amdgcn_readfirstlane(wave >> 1) == (cu >> 5);
    const f32x4 fill = f32x4{__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu)};
    // dz of gate g, local column k = 16 cbb + 4 g + ce: k step cbb / 2, k group 2 (cbb % 2) + g / 2, position 4 (g % 2) + ce
    unsigned short* const dz_out = reinterpret_cast<unsigned short*>(&dzA[0][cbb >> 1][(2 * (cbb & 1)) * 16 + lrow][0]) + ce;
    const f32x4* const w2p = &w2[wave][0][lane];

    float n_si = 0.f, n_tj = 0.f, n_sf = 0.f, n_so = 0.f, n_ct = 0.f, n_cp = 0.f, n_dh = 0.f;
    if (act && a.t1 > a.t0) {
        const int t = a.t1 - 1;
        const float* gp = a.Z + ((size_t)t * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
        n_si = gp[0]; n_tj = gp[4]; n_sf = gp[8]; n_so = gp[12];
        n_ct = a.Cs[(size_t)(t + 1) * B * PH + hi]; n_cp = a.Cs[(size_t)t * B * PH + hi];
        n_dh = a.dH[(size_t)t * B * PH + hi];
    }

    xbf16x8 b2g0[8];                              // plane 2 of the first tile group's B operands
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) b2g0[i8] = as_bf16x8(w2p[((i8 >> 2) * 16 + (i8 & 3)) * 64]);
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = PROF ? __builtin_amdgcn_s_memtime() : 0;   // [5] stores of groups 0-2, [6] late drain, [7] phase C before the first group
    for (int t = a.t1 - 1; t >= a.t0; --t) {
        P16_STAMP(4)
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
                    if (loc) store_l2(in + (size_t)rg * PCU * 16 + k * 64, fill);      // this wave's producers sit on this XCD
                    else store_sc1(in + (size_t)rg * PCU * 16 + k * 64, fill);
                    wsum[rg] = (k == 0) ? v[rg][0] : wsum[rg] + v[rg][k];
                }
        }
        P16_STAMP(0)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) *reinterpret_cast<f32x4*>(&psum[rg][wave][lane][0]) = wsum[rg];
        __syncthreads();
        if (s_fail) return;
        P16_STAMP(1)

        // ---- B: gate gradients (wave rg < RG: lane = 16 i + 4 bb + e); lane group l / 16 of a wave holds producers 16 w + 4 k + l / 16
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
        P16_STAMP(2)
        // The resets of this step's inbox words (and the dz stores) must have landed before anything of this step is published: a
        // producer that sees this CU's partials may write the slot this CU has just reset.  k_lstm_bwd_pair waits for them HERE; with
        // XCD_LATE_DRAIN the wait stands in front of the first partial store instead, behind the first tile group's MFMAs -- the
        // acknowledgements of the write-through resets (~0.6 us) arrive under the cell update and those products.
        const bool late = (a.variant & XCD_LATE_DRAIN) != 0;
        if (!late) drain_vmem();
        if ((a.variant & XCD_DEFER_OUTPUTS) && act) { gp[0] = di; gp[4] = dj; gp[8] = df; gp[12] = dg; }
        auto next_inputs = [&]() __attribute__((always_inline)) {       // gates, cell states and dH of the update one step down
            if (act && t > a.t0) {
                const float* gn = a.Z + ((size_t)(t - 1) * B + row) * PG4 + 64 * cu + 16 * cbb + ce;
                n_si = gn[0]; n_tj = gn[4]; n_sf = gn[8]; n_so = gn[12];
                n_ct = cpv; n_cp = a.Cs[(size_t)(t - 1) * B * PH + hi];
                n_dh = a.dH[(size_t)(t - 1) * B * PH + hi];
            }
        };
        if (!late) next_inputs();           // (late: behind the wait for the resets, so that the wait does not wait for THEM)

        // ---- C: produce the partials of dh_{t-1}: 256 destination units per wave = sixteen 16-unit tiles, in four groups of four so
        // that a group's stores leave while the next group multiplies
        if (t > 0) {
            xbf16x8 av[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) av[pl][ks] = *reinterpret_cast<const xbf16x8*>(&dzA[pl][ks][lane][0]);
            f32x4* out = inbox + (size_t)(t & 1) * slot_w + out_ofs;
            // plane 2 of a group's eight B operands comes out of LDS one group AHEAD (the first group's at the top of the phase): a
            // ds_read_b128 next to the MFMA that reads it costs its full latency, 32 times a step (profiles/r06_pair16_probe2_variants.log:
            // 5870 ticks for 3072 of MFMAs)
            xbf16x8 b2[2][8];
            if (PROF) pacc[7] += __builtin_amdgcn_s_memtime() - plast;
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) b2[0][i8] = b2g0[i8];
            // (Measured and dropped, profiles/r06_pair16_probe4_bwd.log: the four stores of a group issued one by one INSIDE the next group,
            // each behind twelve of its MFMAs -- 5.62 -> 5.77 us per step.  The 48 KiB of partials a CU writes per step leave at ~32 bytes
            // per clock whatever the order, and a wave's MFMAs do not issue past a store that waits for its slot.)
#pragma unroll
            for (int grp4 = 0; grp4 < 4; ++grp4) {
                if (grp4 < 3) {
#pragma unroll
                    for (int i8 = 0; i8 < 8; ++i8) b2[(grp4 + 1) & 1][i8] = as_bf16x8(w2p[((i8 >> 2) * 16 + 4 * (grp4 + 1) + (i8 & 3)) * 64]);
                }
                f32x4 acc[4];
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) acc[j4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#define P16_BWD(AOP, BOP)                                                                                       \
                    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4)                                            \
                        acc[j4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AOP, BOP, acc[j4], 0, 0, 0);
                    P16_BWD(av[2][ks], W[0][ks][4 * grp4 + j4])
                    P16_BWD(av[0][ks], b2[grp4 & 1][ks * 4 + j4])
                    P16_BWD(av[1][ks], W[1][ks][4 * grp4 + j4])
                    P16_BWD(av[1][ks], W[0][ks][4 * grp4 + j4])
                    P16_BWD(av[0][ks], W[1][ks][4 * grp4 + j4])
                    P16_BWD(av[0][ks], W[0][ks][4 * grp4 + j4])
#undef P16_BWD
                }
                if (grp4 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);      // the next group's LDS reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);                    // ... then this group's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                unsigned long long q0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                if (grp4 == 0 && late) { drain_vmem(); next_inputs(); }
                if (PROF && grp4 == 0) { const unsigned long long q1 = __builtin_amdgcn_s_memtime(); pacc[6] += q1 - q0; q0 = q1; }
                if (grp4 == 3) { P16_STAMP(3) }
                if (outl) {
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        if (loc) store_l2(out + (size_t)(4 * grp4 + j4) * RG * PCU * 16, acc[j4]);      // this wave's destinations sit on this XCD
                        else store_sc1(out + (size_t)(4 * grp4 + j4) * RG * PCU * 16, acc[j4]);
                    }
                }
                if (PROF && grp4 < 3) pacc[5] += __builtin_amdgcn_s_memtime() - q0;
            }
        }
    }
    if (act) a.dc[hi] = dcv;
    if (PROF && lane == 0 && a.prof) {
        P16_STAMP(4)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)(role.xcd * NCU + role.cu) * 4 + wave) * 8 + i] = pacc[i];
    }
}
#undef P16_TERMS_REG
#undef P16_STAMP

// Kh [1024][4096] (packed gate columns) -> the three-plane bf16 images of the two kernels above, 16-byte words
// [64 cu][4 w][96][64 lanes] (planes 0 and 1 -> registers, plane 2 -> LDS):
//   fwd word (pl * 8 + ks) * 4 + nt,  lane l, position j: plane pl of Kh[256 w + 32 ks + 8 (l / 16) + j][64 cu + 16 nt + l % 16]
//   bwd word (pl * 2 + ks) * 16 + nt, lane l, position j: plane pl of Kh[256 w + 16 nt + l % 16][64 cu + 32 ks + 8 (l / 16) + j]
__device__ __forceinline__ void repack_kh_pair16_body(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd, int b, int nb) {
    const int total = PCU * 4 * 32 * 64;            // (cu, w, 32 operand slots, lane) per image
    uint4* const fo = reinterpret_cast<uint4*>(fwd);
    uint4* const bo = reinterpret_cast<uint4*>(bwd);
    for (int idx = b * blockDim.x + threadIdx.x; idx < total; idx += nb * blockDim.x) {
        const int l = idx & 63, slot = (idx >> 6) & 31, w = (idx >> 11) & 3, cu = idx >> 13;
        float x[8];
        unsigned pk[3][4];
        const size_t base = ((size_t)(cu * 4 + w) * P16W) * 64 + l;
        {
            const int ks = slot >> 2, nt = slot & 3;
            const float* src = Kh + (size_t)(PKW * w + 32 * ks + 8 * (l >> 4)) * PG4 + 64 * cu + 16 * nt + (l & 15);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = src[(size_t)j * PG4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned p0[3], p1[3];
                split3(x[j], p0); split3(x[j + 1], p1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk[pl][j >> 1] = (p0[pl] & 0xffffu) | (p1[pl] << 16);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fo[base + (size_t)((pl * P16KS + ks) * 4 + nt) * 64] = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
        }
        {
            const int ks = slot >> 4, nt = slot & 15;
            const float* src = Kh + (size_t)(PKW * w + 16 * nt + (l & 15)) * PG4 + 64 * cu + 32 * ks + 8 * (l >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = src[j];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned p0[3], p1[3];
                split3(x[j], p0); split3(x[j + 1], p1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pk[pl][j >> 1] = (p0[pl] & 0xffffu) | (p1[pl] << 16);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bo[base + (size_t)((pl * 2 + ks) * 16 + nt) * 64] = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
        }
    }
}
__global__ void k_repack_kh_pair16(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd) {
    repack_kh_pair16_body(Kh, fwd, bwd, blockIdx.x, gridDim.x);
}
