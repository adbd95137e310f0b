// fp32 GEMMs for gfx950: k_gemm_bx3 (default) assembles every fp32 product from bf16 pieces on v_mfma_f32_32x32x16_bf16
// (16x the fp32 MFMA's rate; exact three-way operand split, six products, fp32 accumulation -- see the block comment above
// it), k_gemm / k_gemm_staged / k_gemm_queue run on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).  The notes below
// describe the fp32-MFMA kernels; tiling, epilogue, split-K and tile order are shared.
//
// One kernel template serves every dense contraction of the LSTM-baseline step
// (DESIGN.md "Kernels"): the hoisted input projection with the embedding gather fused
// into the A-operand load, the vocabulary projection, and their backward pairs (dlogits is
// materialised once by the cross-entropy kernel: an earlier version recomputed
// exp(logit - lse) - onehot inside the operand loads of two GEMMs, 4x per element, and was
// slower).
//
// Tiling: 128x128 block tile, BK = 16, 256 threads = 4 wave64 as 2x2, each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).  Two kernels:
//   * k_gemm (the default, "main kernel" below): dense lane-linear LDS tiles, k-contiguous sources stored
//     untransposed and read with ds_read_b128 under a permuted contraction order, branch-free steady loop;
//   * k_gemm_staged (first version; kept for the x-contiguous operand whose K rows are GATHERED, dKx): both
//     operands K-major in LDS ([k][x], row stride 132 floats for 16-byte copies / 129 for k-contiguous sources
//     that are transposed on the way in with 4 ds_write_b32), predicated loads.
// Both: global loads for tile t+1 are issued before the MFMAs of tile t (register prefetch) and written to the
// other LDS buffer afterwards, one barrier per K tile; blocks are numbered so that the 8 XCDs (block b -> XCD
// b % 8 as observed on MI355X) each walk a contiguous range of tiles and share operand panels in their private L2.
#include "fsmg_kernels.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

namespace fsmg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifndef FSMG_GEMM_BK
#define FSMG_GEMM_BK 16
#endif
// BK = 16: 34 KiB LDS and <=104 VGPRs per block -> 4 resident blocks (16 waves) per CU, which covers the
// per-tile barrier and the prologue/epilogue bubbles better than 2 blocks of BK = 32 (measured +5..15 % per GEMM)
#ifndef FSMG_GEMM_BM
#define FSMG_GEMM_BM 128
#endif
constexpr int BM = FSMG_GEMM_BM, BN = 128, BK = FSMG_GEMM_BK;   // BM in {128, 256}, BK in {16, 32}
constexpr int NTHREADS = 2 * BM;                       // one wave64 per 64x64 sub-tile: (BM/64) x 2 waves
constexpr int NWAVES = NTHREADS / 64;
constexpr int KQ = BK / 4;                             // float4 slots along k in a KC tile row
// 16-byte loads per thread for an operand tile of XW rows/columns
template <int XW> struct Nld { static constexpr int v = XW * BK / 4 / NTHREADS; };
// LDS row strides: KC tiles are scattered with ds_write_b32 (stride = 1 mod 32 -> conflict free), XC tiles are
// copied with ds_write_b128 (stride a multiple of 4 floats)
template <int MODE, int XW> struct TileLd { static constexpr int v = (MODE == OP_KC) ? XW + 1 : XW + 4; };

// ---- KC source: tile [128 x][32 k], k contiguous.  thread -> rows x = tid/8 + 32*i, k quad kq = tid%8
template <int XW>
struct KcLoader {
    static constexpr int NLD = Nld<XW>::v, LD = XW + 1;
    const float* rowp[NLD];
    int kq;
    __device__ __forceinline__ void init(const float* src, int ld, int X, int x0, const int* gather, int tid) {
        kq = tid % KQ;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int x = x0 + (tid / KQ) + (NTHREADS / KQ) * i;
            bool ok = x < X;
            long long row = ok ? (gather ? (long long)gather[x] : (long long)x) : 0;
            rowp[i] = ok ? src + row * ld : nullptr;
        }
    }
    __device__ __forceinline__ void load(float4 (&r)[NLD], int k0, int kend) const {
        int k = k0 + 4 * kq;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowp[i] != nullptr && k < kend) v = *reinterpret_cast<const float4*>(rowp[i] + k);
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, const float4 (&r)[NLD], int tid) const {
        float* base = lds + (4 * kq) * LD + (tid / KQ);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            base[0 * LD + (NTHREADS / KQ) * i] = r[i].x;
            base[1 * LD + (NTHREADS / KQ) * i] = r[i].y;
            base[2 * LD + (NTHREADS / KQ) * i] = r[i].z;
            base[3 * LD + (NTHREADS / KQ) * i] = r[i].w;
        }
    }
};

// ---- XC source: tile [32 k][128 x], x contiguous.  thread -> k rows tid/32 + 8*i, x quad xq = tid%32
template <int XW>
struct XcLoader {
    static constexpr int NLD = Nld<XW>::v, LD = XW + 4, XQ = XW / 4, KSTEP = NTHREADS / XQ;
    const float* colp;   // src + x  (nullptr if x beyond X)
    const int* gather;
    int ld, x;
    __device__ __forceinline__ void init(const float* src, int ld_, int X, int x0, const int* gather_, int tid) {
        ld = ld_;
        gather = gather_;
        x = x0 + 4 * (tid % XQ);
        colp = (x < X) ? src + x : nullptr;
    }
    __device__ __forceinline__ void load(float4 (&r)[NLD], int k0, int kend, int tid) const {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int k = k0 + (tid / XQ) + KSTEP * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (colp != nullptr && k < kend) {
                long long row = gather ? (long long)gather[k] : (long long)k;
                v = *reinterpret_cast<const float4*>(colp + row * ld);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* lds, const float4 (&r)[NLD], int tid) const {
        float* base = lds + (tid / XQ) * LD + 4 * (tid % XQ);
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<float4*>(base + KSTEP * i * LD) = r[i];
    }
};

template <int MODE, int XW> struct Loader;
template <int XW> struct Loader<OP_KC, XW> : KcLoader<XW> {
    __device__ __forceinline__ void fetch(float4 (&r)[Nld<XW>::v], int k0, int kend, int) const { this->load(r, k0, kend); }
};
template <int XW> struct Loader<OP_XC, XW> : XcLoader<XW> {
    __device__ __forceinline__ void fetch(float4 (&r)[Nld<XW>::v], int k0, int kend, int tid) const { this->load(r, k0, kend, tid); }
};

// ---- epilogue shared by both kernels.  MFMA 32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): a
// lane holds single floats of 16 rows.  Each wave transposes its 64x64 sub-tile through its own slice of the (now
// idle) pipeline LDS, 32 rows at a time, and stores 16-byte row segments: 16 store instructions per lane
// instead of 64 (the store tail of a short-K GEMM is issue bound).
// the 64 x 64 sub-tile whose first row / column is mrow0 / ncol0, staged through the wave-private slice `ep` (32 x 68 floats);
// wn = which 64-column half of its 128-column tile tn this is (the forward-only cross entropy's partials are per half)
// sum / maximum over the 16 lanes of a DPP row (every lane gets the result): quad xor 1, quad xor 2, row_half_mirror, row_mirror.
// (__shfl_xor(v, o, 16) is a ds_bpermute -- an LDS-path instruction with its own wait -- per step: four to eight of them in
// each of a tile's 32 store passes cost the fused-softmax projection 12 % of its tile time.)
#define FSMG_ROW16(OP)                                                                                          \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));        \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));        \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));       \
    v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
// lane J of the own DPP row of 16 (row_newbcast:J)
template <int J> __device__ __forceinline__ float cs_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_add_(float a, float b) { return a + b; }
__device__ __forceinline__ float row16_sum(float v) { FSMG_ROW16(row16_add_) return v; }
__device__ __forceinline__ float row16_max(float v) { FSMG_ROW16(fmaxf) return v; }
#undef FSMG_ROW16

__device__ __forceinline__ void store_tile_at(const GemmArgs& g, f32x16 (&acc)[2][2], float* ep, int z, int mrow0, int ncol0,
                                              int tn, int tilesN, int wn, int lane) {
    const int l31 = lane & 31, khalf = lane >> 5;
    float* C = g.C + (long long)z * g.c_slab;
    constexpr int EP_LD = 68;                                  // 64 + 4: rows stay 16-byte aligned
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int tg[8];                                             // forward-only cross entropy: the targets of this lane's eight rows, loaded
        if (g.ce_part != nullptr && !g.ce_store) {             // together ahead of the transpose (one load latency, not one per pass)
#pragma unroll
            for (int p = 0; p < 8; ++p) { const int row = mrow0 + i * 32 + p * 4 + (lane >> 4); tg[p] = row < g.M ? g.ce_tgt[row] : -1; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * khalf) * EP_LD + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): own LDS writes landed (wave-private slice)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rl = p * 4 + (lane >> 4), c4 = (lane & 15) * 4;
            const int row = mrow0 + i * 32 + rl, col = ncol0 + c4;
            float4 v = *reinterpret_cast<const float4*>(ep + rl * EP_LD + c4);
            if (g.ce_part != nullptr && g.ce_store) {
                // train pass, fused softmax (GemmArgs::ce_store): E = exp(x) with NO shift into C (zero in the pad columns) and the sum of
                // the wave's 64 columns as this row's partial -- no row maximum, no target lookup (a dependent global load per pass cost
                // 14 % of the tile): k_ce_finish range-checks the row SUM and takes the target logit as log(E[target])
                if (g.bias != nullptr && col < g.N) {
                    const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                float4 e;
                e.x = (col + 0 < g.ce_nvocab) ? __expf(v.x) : 0.0f; e.y = (col + 1 < g.ce_nvocab) ? __expf(v.y) : 0.0f;
                e.z = (col + 2 < g.ce_nvocab) ? __expf(v.z) : 0.0f; e.w = (col + 3 < g.ce_nvocab) ? __expf(v.w) : 0.0f;
                float sm = (e.x + e.y) + (e.z + e.w);
                sm = row16_sum(sm);
                if (row < g.M) {
                    if ((lane & 15) == 0) g.ce_part[(long long)row * (2 * tilesN) + 2 * tn + wn] = make_float2(0.0f, sm);
                    if (col < g.N) {
                        float* dst = C + (long long)row * g.ldc + col;
                        __builtin_nontemporal_store(e.x, dst); __builtin_nontemporal_store(e.y, dst + 1);
                        __builtin_nontemporal_store(e.z, dst + 2); __builtin_nontemporal_store(e.w, dst + 3);
                    }
                }
                continue;
            }
            if (g.ce_part != nullptr) {
                // forward-only cross entropy: softmax statistics of this row over the wave's 64 columns; the 16
                // lanes of a row (lane>>4 picks the row of this pass) reduce with width-16 shuffles
                if (g.bias != nullptr && col < g.N) {
                    const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                const float NEG = -INFINITY;
                const float x0 = (col + 0 < g.ce_nvocab) ? v.x : NEG, x1 = (col + 1 < g.ce_nvocab) ? v.y : NEG;
                const float x2 = (col + 2 < g.ce_nvocab) ? v.z : NEG, x3 = (col + 3 < g.ce_nvocab) ? v.w : NEG;
                float m = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
                m = row16_max(m);
                float sm = 0.0f;
                if (m > NEG) sm = (expf(x0 - m) + expf(x1 - m)) + (expf(x2 - m) + expf(x3 - m));
                sm = row16_sum(sm);
                if (row < g.M) {
                    const int t = tg[p];
                    if (t >= col && t < col + 4) g.ce_tgt_logit[row] = (t == col) ? v.x : (t == col + 1) ? v.y : (t == col + 2) ? v.z : v.w;
                    if ((lane & 15) == 0) g.ce_part[(long long)row * (2 * tilesN) + 2 * tn + wn] = make_float2(m, sm);
                }
                continue;
            }
            if (row < g.M && col < g.N) {                      // N, ldc are multiples of 4: whole float4 in or out
                if (g.bias != nullptr && z == 0) {
                    const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                float4* dst = reinterpret_cast<float4*>(C + (long long)row * g.ldc + col);
#ifdef FSMG_EXPERIMENTS
                if (g.done != nullptr) {    // a consumer on other CUs reads this tile while the launch is still running (GemmArgs::done): write-through at
                    // agent scope, so that "the wave's stores have been acknowledged" (s_waitcnt vmcnt(0)) means "visible to every XCD" and the
                    // tile needs no L2 write-back of its own (an agent-scope release fence per tile flushes the XCD's whole L2 -- beside a
                    // kernel that streams 460 MB through it: measured, the pair's tail went from 100 to 460 us)
                    typedef float f4_ __attribute__((ext_vector_type(4)));
                    const f4_ q = {v.x, v.y, v.z, v.w};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(dst), "v"(q) : "memory");
                } else
#endif
                if (g.nt_store) {       // write-once streaming output (logits): keep it out of the way of L2-resident data
                    __builtin_nontemporal_store(v.x, &dst->x); __builtin_nontemporal_store(v.y, &dst->y);
                    __builtin_nontemporal_store(v.z, &dst->z); __builtin_nontemporal_store(v.w, &dst->w);
                } else {
                    *dst = v;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // reads done before the slice is overwritten
    }
}

__device__ __forceinline__ void store_tile(const GemmArgs& g, f32x16 (&acc)[2][2], float* smem, int z, int m0, int n0,
                                           int tn, int tilesN, int wave, int lane) {
    const int wm = wave >> 1, wn = wave & 1;
    store_tile_at(g, acc, smem + wave * (32 * 68), z, m0 + wm * 64, n0 + wn * 64, tn, tilesN, wn, lane);   // one 8.5 KiB slice per wave
}

// XCD-aware tile numbering (bijective for any tile count): block b runs on XCD b % 8; each XCD walks a contiguous
// range of tiles, M fastest, so the blocks sharing a B panel sit in one L2
__device__ __forceinline__ int xcd_tile(int bid, int nb) {
    const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// tile id (in the order an XCD walks its range) -> (row tile, column tile).  group_m = 0: row tiles fastest (a column
// panel of B is shared by consecutive blocks, A is streamed once per column panel).  group_m = G > 0: super-rows of G row
// tiles, inside a super-row column tiles slowest... i.e. G consecutive ids share a B panel and the next G the same A
// panels: the blocks resident on an XCD at one time then touch few panels of EITHER operand, so both stay in its 4 MiB L2
// (G = 1: column tiles fastest, for GEMMs whose A is the big operand).  Placement only; results do not change.
__device__ __forceinline__ void tile_coords(int t, int tilesM, int tilesN, int group_m, int& tm, int& tn) {
    if (group_m <= 0) { tm = t % tilesM; tn = t / tilesM; return; }
    const int per = group_m * tilesN;
    const int r = t / per, w = t - r * per;
    const int gm = min(group_m, tilesM - r * group_m);
    tn = w / gm;
    tm = r * group_m + (w - tn * gm);
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(NTHREADS, (BK == 16 ? 1024 : 512) / NTHREADS) void k_gemm_staged(const GemmArgs g) {
    constexpr int LDA = TileLd<AMODE, BM>::v, LDB = TileLd<BMODE, BN>::v;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    constexpr int PIPE = 2 * ASZ + 2 * BSZ, EPI = NWAVES * 32 * 68;
    __shared__ __attribute__((aligned(16))) float smem[PIPE > EPI ? PIPE : EPI];
    float* As = smem;
    float* Bs = smem + 2 * ASZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;

    // ---- XCD-aware tile numbering (bijective for any tile count)
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int nb = tilesM * tilesN;
    const int bid = xcd_tile(blockIdx.x, nb);
    const int tm = bid % tilesM, tn = bid / tilesM;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- K range of this split
    const int z = blockIdx.y;
    int kb = 0, ke = g.K;
    if (g.ksplit > 1) {
        int per = ((g.K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK;
        kb = z * per;
        ke = min(g.K, kb + per);
    }
    const int nk = (ke > kb) ? (ke - kb + BK - 1) / BK : 0;

    Loader<AMODE, BM> la;
    Loader<BMODE, BN> lb;
    la.init(g.A, g.lda, g.M, m0, g.gather, tid);
    lb.init(g.B, g.ldb, g.N, n0, nullptr, tid);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const bool do_colsum = (BMODE == OP_XC) && g.colsum != nullptr && tm == 0 && tid < BN;
    float csum = 0.0f;

    float4 ra[Nld<BM>::v], rb[Nld<BN>::v];
    if (nk > 0) {
        la.fetch(ra, kb, ke, tid);
        lb.fetch(rb, kb, ke, tid);
        la.store(As, ra, tid);
        lb.store(Bs, rb, tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            la.fetch(ra, kb + (kt + 1) * BK, ke, tid);
            lb.fetch(rb, kb + (kt + 1) * BK, ke, tid);
        }
        const float* a_base = As + cur * ASZ + khalf * LDA + wm * 64 + l31;
        const float* b_base = Bs + cur * BSZ + khalf * LDB + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = a_base[kk * LDA], a1 = a_base[kk * LDA + 32];
            const float b0 = b_base[kk * LDB], b1 = b_base[kk * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (do_colsum) {
            const float* bc = Bs + cur * BSZ + tid;
#pragma unroll
            for (int k = 0; k < BK; ++k) csum += bc[k * LDB];
        }
        if (more) {
            la.store(As + (cur ^ 1) * ASZ, ra, tid);
            lb.store(Bs + (cur ^ 1) * BSZ, rb, tid);
        }
        __syncthreads();
    }

    store_tile(g, acc, smem, z, m0, n0, tn, tilesN, wave, lane);
    if (do_colsum && n0 + tid < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + tid] = csum;
}

// ================================================================ main kernel
// LDS images (dense, 8 KiB per 128 x 16 operand tile, "lane-linear": wave w owns pieces 2w and 2w+1 of each tile,
// a piece is 1 KiB, and lane l's 16 bytes sit at piece + 16*l, so every ds_write_b128 is conflict free):
//   XC tile: [16 k][128 x] (rows of 512 B).  Piece p = rows 2p, 2p+1; lane -> row 2p + lane/32, x = 4*(lane%32).
//            MFMA operand read = ds_read_b32 at [k][x]: the two 32-lane halves read different rows and never
//            conflict (ds_read_b32 is serviced per half).
//   KC tile: 16-byte slots S = 4x + r; slot (x, r) holds k = 4*kq .. 4*kq+3 of row x with r = (kq + x/4) % 4.
//            Piece p = rows 16p .. 16p+15; lane -> x = 16p + lane/4, r = lane%4: the 4 lanes of a row fetch its 64
//            contiguous bytes (in rotated order) and store them untransposed.  MFMA operand read = one ds_read_b128
//            per 4 MFMAs; the rotation spreads each of that instruction's 16-lane groups over all 64 banks.
// K order inside a tile: a ds_read_b128 hands a lane four consecutive k of one row, so MFMA (jj, j), j = 0..3,
// contracts k = 4*(2*jj + half) + j (half = lane/32) instead of k = 2*step + half; both operands follow the same
// convention in either storage mode, and a sum over k does not care about the order it is taken in.
// Edges: rows/columns past M or N are clamped to the last valid one (their products land in output rows/columns
// that are never stored), so the steady-state loop has no predicated loads; a K range that is not a multiple of 16
// zero-fills its last tile.
// Global loads for tile t+1 are issued before the MFMAs of tile t (register prefetch) and written to the other LDS
// buffer afterwards: one barrier per K tile.  (LDS-DMA, global_load_lds_dwordx4 into the same images, was measured
// 5-10 % slower than this at 4 pieces per wave and tile: DESIGN.md "rejected".)
constexpr int TILE_F = 128 * BK;        // floats per operand tile (both layouts are dense)
constexpr int PIECE_F = 256;            // floats per wave instruction (1 KiB)

template <int MODE>
struct Stager {
    const float* p[2];
    float4 r[2];
    long long step;
    int kofs[2];
    __device__ __forceinline__ void init(const float* src, int ld, int X, int x0, const int* gather, int kb, int wave, int lane) {
        if (MODE == OP_XC) {
            int x = x0 + 4 * (lane & 31);
            x = min(x, X - 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                kofs[i] = 4 * wave + 2 * i + (lane >> 5);
                p[i] = src + (long long)(kb + kofs[i]) * ld + x;
            }
            step = (long long)BK * ld;
        } else {
            const int xi = lane >> 2, r = lane & 3;
            const int kq = (r - (xi >> 2)) & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int x = x0 + 16 * (2 * wave + i) + xi;
                x = min(x, X - 1);
                const long long row = gather ? (long long)gather[x] : (long long)x;
                kofs[i] = 4 * kq;
                p[i] = src + row * ld + kb + 4 * kq;
            }
            step = BK;
        }
    }
    // full tile: unconditional 16-byte loads now ...
    __device__ __forceinline__ void fetch() {
#pragma unroll
        for (int i = 0; i < 2; ++i) { r[i] = *reinterpret_cast<const float4*>(p[i]); p[i] += step; }
    }
    // ... LDS writes after the MFMAs of the current tile
    __device__ __forceinline__ void commit(float* tile, int wave, int lane) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(tile + (2 * wave + i) * PIECE_F + 4 * lane) = r[i];
    }
    // partial tile (k0 + 16 > kend): zeros past kend
    __device__ __forceinline__ void fetch_partial(int k0, int kend) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + kofs[i] < kend) r[i] = *reinterpret_cast<const float4*>(p[i]);
            p[i] += step;
        }
    }
};

// the four operand values of lane (x, half) for MFMAs (jj, 0..3)
template <int MODE>
__device__ __forceinline__ void read_frag(const float* tile, int x, int half, int jj, float (&f)[4]) {
    if (MODE == OP_KC) {
        const int kq = 2 * jj + half;
        const float4 v = *reinterpret_cast<const float4*>(tile + 4 * (4 * x + ((kq + (x >> 2)) & 3)));
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        const float* r = tile + (4 * (2 * jj + half)) * 128 + x;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = r[j * 128];
    }
}

// one 128 x 128 tile of split z: the body shared by the static-grid kernel and the work-queue kernel below
template <int AMODE, int BMODE>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, float* smem, int tm, int tn, int z, int tilesN) {
    constexpr int TILE_F_ = TILE_F;
    float* As = smem;                   // [2][TILE_F]
    float* Bs = smem + 2 * TILE_F_;     // [2][TILE_F]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int m0 = tm * 128, n0 = tn * 128;

    int kb = 0, ke = g.K;
    if (g.ksplit > 1) {
        const int per = ((g.K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK;
        kb = z * per;
        ke = min(g.K, kb + per);
    }
    const int nk = (ke > kb) ? (ke - kb + BK - 1) / BK : 0;
    const int nfull = (ke > kb) ? (ke - kb) / BK : 0;

    Stager<AMODE> sa;
    Stager<BMODE> sb;
    sa.init(g.A, g.lda, g.M, m0, g.gather, kb, wave, lane);
    sb.init(g.B, g.ldb, g.N, n0, nullptr, kb, wave, lane);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const bool do_colsum = (BMODE == OP_XC) && g.colsum != nullptr && tm == 0 && tid < 128;
    float csum = 0.0f;

    if (nk > 0) {
        if (nfull > 0) { sa.fetch(); sb.fetch(); }
        else { sa.fetch_partial(kb, ke); sb.fetch_partial(kb, ke); }
        sa.commit(As, wave, lane);
        sb.commit(Bs, wave, lane);
    }
    __syncthreads();

    const int xa = wm * 64 + l31, xb = wn * 64 + l31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            if (kt + 1 < nfull) { sa.fetch(); sb.fetch(); }
            else { sa.fetch_partial(kb + (kt + 1) * BK, ke); sb.fetch_partial(kb + (kt + 1) * BK, ke); }
        }
        const float* at = As + cur * TILE_F_;
        const float* bt = Bs + cur * TILE_F_;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float a0[4], a1[4], b0[4], b1[4];
            read_frag<AMODE>(at, xa, khalf, jj, a0);
            read_frag<AMODE>(at, xa + 32, khalf, jj, a1);
            read_frag<BMODE>(bt, xb, khalf, jj, b0);
            read_frag<BMODE>(bt, xb + 32, khalf, jj, b1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (do_colsum) {
            const float* bc = bt + tid;
#pragma unroll
            for (int k = 0; k < BK; ++k) csum += bc[k * 128];
        }
        if (more) {
            sa.commit(As + (cur ^ 1) * TILE_F_, wave, lane);
            sb.commit(Bs + (cur ^ 1) * TILE_F_, wave, lane);
        }
        __syncthreads();
    }

    store_tile(g, acc, smem, z, m0, n0, tn, tilesN, wave, lane);
    if (do_colsum && n0 + tid < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + tid] = csum;
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, 4) void k_gemm(const GemmArgs g) {
    static_assert(BK == 16 && BM == 128, "the direct-to-LDS kernel is written for 128x128x16 tiles");
    constexpr int PIPE = 4 * TILE_F, EPI = 4 * 32 * 68;
    __shared__ __attribute__((aligned(1024))) float smem[PIPE > EPI ? PIPE : EPI];
    const int tilesM = (g.M + 127) / 128, tilesN = (g.N + 127) / 128;
    const int bid = xcd_tile(blockIdx.x, tilesM * tilesN);
    gemm_tile<AMODE, BMODE>(g, smem, bid % tilesM, bid / tilesM, blockIdx.y, tilesN);
}

// Work-queue variant for the XCD-partitioned schedule (GemmArgs::xcd_first): every block draws ONE (split, row tile,
// column tile) item, row tiles slowest, in the order blocks start -- so XCDs that are busy with something else simply
// draw fewer.  A restricted launch
// (xcd_first > 0) only lets blocks on XCDs >= xcd_first draw, and only while *stop == 0 and below work_limit items; the
// clean-up launch (xcd_first < 0) drains the queue chip-wide.  Which block computes a tile never changes the tile.
template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, 4) void k_gemm_queue(const GemmArgs g) {
    constexpr int PIPE = 4 * TILE_F, EPI = 4 * 32 * 68;
    __shared__ __attribute__((aligned(1024))) float smem[PIPE > EPI ? PIPE : EPI];
    __shared__ int s_item;
    const int tilesM = (g.M + 127) / 128, tilesN = (g.N + 127) / 128;
    const int total = tilesM * tilesN * (g.ksplit > 1 ? g.ksplit : 1);
    const int limit = g.xcd_first > 0 ? min(g.work_limit, total) : total;
    if (g.xcd_first > 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 7) < g.xcd_first) return;
    }
    if (threadIdx.x == 0) {
        // one atomicAdd per block (a CAS loop on the counter serialises the whole chip at ~2 us per item); the per-item
        // claim word settles restricted launch against clean-up launch
        int item = -1;
        if (g.xcd_first < 0) {
            const int j = atomicAdd(g.work + 1, 1);
            if (j < total && atomicCAS(g.claim + j, 0, 1) == 0) item = j;
        } else if (__hip_atomic_load(g.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
            const int j = atomicAdd(g.work, 1);
            if (j < limit && atomicCAS(g.claim + j, 0, 1) == 0) item = j;
        }
        s_item = item;
    }
    __syncthreads();
    const int item = s_item;
    if (item < 0) return;
    gemm_tile<AMODE, BMODE>(g, smem, (item / tilesN) % tilesM, item % tilesN, item / (tilesM * tilesN), tilesN);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix pipe ("bx3"): v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so an
// fp32 product is assembled from bf16 pieces.  Every operand value is split EXACTLY into three bf16 numbers
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// (round-to-nearest: |a2| <= 2^-8 |a|, |a3| <= 2^-16 |a|, and the last residual fits 8 significant bits; the two
// subtractions are exact in fp32), and the six partial products that matter are accumulated in fp32:
//     a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2),   dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|
// i.e. one half-ulp of the fp32 product, unbiased -- the accumulation (fp32, as in the fp32 MFMA) dominates the error
// either way.  6 MFMAs of 32x32x16 (32 cycles each) replace 8 of 32x32x2 (64 cycles each) per 16 k: 2.67x the rate.
// (Finite inputs, that is -- values above bf16's largest, 3.39e38, included: measured finite.  An Inf operand makes the
// second piece NaN (bf16(Inf - Inf)) and the products it feeds NaN where the fp32 MFMA produces Inf; pinned by
// tests/test_gpu_parity.py::test_non_finite_weights_give_nan_...; NaN inputs give NaN in both.)
// The split runs on the VALU between the global load and the LDS write (5.5 instructions per element), the LDS holds
// the three planes as [plane][k half][128 x][8 bf16] so that a fragment is one conflict-free ds_read_b128 per lane.
// Same tiling, epilogue, split-K and XCD-aware tile order as k_gemm; same A / B conventions.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// two fp32 values -> their three bf16 planes, packed (x in the low half)
__device__ __forceinline__ void split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(sx, sy);
}

// the same for eight values, the four independent chains advanced stage by stage: the instructions of one chain depend on
// each other (a wave alone needs 62 cycles per pair that way, tools/split_probe), and the compiler keeps inline asm in
// source order, so the interleaving is written out
__device__ __forceinline__ float sub_f32(float a, unsigned b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split8(const float (&v)[8], unsigned (&w)[3][4]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[0][j] = cvt_pk_bf16(v[2 * j], v[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[2 * j] = sub_f32(v[2 * j], w[0][j] << 16); r[2 * j + 1] = sub_f32(v[2 * j + 1], w[0][j] & 0xffff0000u); }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[1][j] = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[2 * j] = sub_f32(r[2 * j], w[1][j] << 16); r[2 * j + 1] = sub_f32(r[2 * j + 1], w[1][j] & 0xffff0000u); }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[2][j] = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
}

constexpr int BX_PLANE = 2 * 128 * 16;          // bytes: [2 k halves][128 x][8 bf16]
constexpr int BX_OPER = 3 * BX_PLANE;           // one operand tile, three planes
constexpr int BX_STAGE = 2 * BX_OPER;           // A + B

// Global -> registers -> LDS for one operand tile (128 x, 16 k).  Eight values per thread either way:
//   XC (x contiguous): thread = (x = tid % 128, k half = tid / 128), eight 4-byte loads down k (a wave reads 256
//       contiguous bytes per load); one 16-byte LDS write per plane;
//   KC (k contiguous): thread = (rows tid / 4 and tid / 4 + 64, k quad = tid % 4), two 16-byte loads (four lanes cover
//       the 64 bytes a row contributes to the tile); two 8-byte LDS writes per plane.
//   XC with gathered K rows (dKx: row k of the operand is row gather[k] of the embedding): the eight row ids of the NEXT
//       tile are fetched one tile ahead, so a tile's loads do not wait for an index load first.
template <int MODE, int XT = 128, int NT = 256, bool BUF = false>   // XT: x extent of the LDS tile written into; NT threads cover NT / 2 of them
struct BxStager {
    static constexpr int KH = XT * 16, PLANE = 2 * KH, XW = NT / 2;
    const float* p[2];
    const float* src_;
    float v[8];
    long long step, ld_;
    int lds_ofs[2];
    const int* gp;          // XC + gather: &gather[k] of this thread's first row of the next tile
    int gk, gK;             // ... that k, and the K bound of the gather array
    int gi[8];
    int kstep;              // k advance per fetch: 16, or 32 when two stagers take alternate tiles (k_gemm_bx3w)
    // Buffer loads (SGPR base + 32-bit lane offset + SGPR k offset) instead of 64-bit lane addresses wherever the operand
    // fits 4 GiB and has no gathered K rows: the eight 4-byte loads of an x-contiguous tile then need no per-load address
    // arithmetic on the VALU (a loader wave of k_gemm_bx3w issues them in 590 cycles instead of 1650 with all loads hitting
    // the L1, profiles/r03_gemm_buf1.log; dW 0.435 -> 0.397 ms, dKh 0.094 -> 0.085, the projection 0.335 -> 0.302)
    // (BUF is the launcher's decision: launch_t checks the sizes and the gather)
    __amdgpu_buffer_rsrc_t rs;
    unsigned vo[2], so, sstep;
    static constexpr bool use_buf = BUF;
    bool coherent = false;      // KC buffer path: agent-scope (sc1) loads -- the rows are being written by a kernel on another XCD (gated queue launch)
    __device__ __forceinline__ void load_ids() {
#pragma unroll
        for (int i = 0; i < 8; ++i) gi[i] = gp[min(i, gK - 1 - gk)];      // clamped: the id of a row past K is never used
    }
    __device__ __forceinline__ void init(const float* src, int ld, int X, int x0, const int* gather, int kb, int tid, int K = 0, int kstep_ = 16, int xl0 = 0) {
        ld_ = ld; gp = nullptr; kstep = kstep_;
        if (MODE == OP_XC) {
            const int x = min(x0 + (tid % XW), X - 1), kh = tid / XW;
            p[0] = src + (long long)(kb + 8 * kh) * ld + x; p[1] = nullptr;
            step = (long long)kstep * ld;
            lds_ofs[0] = kh * KH + (xl0 + (tid % XW)) * 16; lds_ofs[1] = 0;
            if (gather != nullptr) {
                p[0] = src + x;
                gK = K; gk = min(kb + 8 * kh, K - 1); gp = gather + gk;
                load_ids();
            }
        } else {
            const int kq = tid & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int xl = (tid >> 2) + (NT / 4) * i, x = min(x0 + xl, X - 1);
                const long long row = gather ? (long long)gather[x] : (long long)x;
                p[i] = src + row * ld + kb + 4 * kq;
                lds_ofs[i] = (kq >> 1) * KH + (xl0 + xl) * 16 + (kq & 1) * 8;
            }
            step = kstep;
        }
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0xfffffffcu, 0x00020000);
        so = 0; sstep = (unsigned)(step * 4);
        vo[0] = (unsigned)((p[0] - src) * 4);
        vo[1] = (MODE == OP_XC) ? 0u : (unsigned)((p[1] - src) * 4);
        src_ = src;            // (buffer-load path: the lane pointers p[] are not kept, the K tail rebuilds them from vo[])
    }
    __device__ __forceinline__ void advance_ids() {          // ids of the tile after the one just requested
        const int nk = min(gk + kstep, gK - 1);
        gp += nk - gk; gk = nk;
        load_ids();
    }
    __device__ __forceinline__ void fetch() {
        if (MODE == OP_XC) {
            if (gp != nullptr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = p[0][(long long)gi[i] * ld_];
                advance_ids();
                return;
            }
            if (use_buf) {
                const unsigned s0 = __builtin_amdgcn_readfirstlane(so), ldb = (unsigned)ld_ * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[0], s0 + i * ldb, 0));
                so += sstep;                // (p stays at the K range's start: fetch_partial adds `so`)
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p[0][i * ld_];
            p[0] += step;
        } else {
            if (use_buf) {
                typedef unsigned u4_ __attribute__((ext_vector_type(4)));
                const unsigned s0 = __builtin_amdgcn_readfirstlane(so);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u4_ q = coherent ? __builtin_amdgcn_raw_buffer_load_b128(rs, vo[i], s0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rs, vo[i], s0, 0);
                    v[4 * i] = __uint_as_float(q.x); v[4 * i + 1] = __uint_as_float(q.y); v[4 * i + 2] = __uint_as_float(q.z); v[4 * i + 3] = __uint_as_float(q.w);
                }
                so += sstep;
                return;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 q = *reinterpret_cast<const float4*>(p[i]);
                v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
                p[i] += step;
            }
        }
    }
    // last tile of a K range: k0 = first k of the tile, zeros from kend on
    __device__ __forceinline__ void fetch_partial(int k0, int kend, int tid) {
        const long long adv = use_buf ? (long long)(so >> 2) : 0;      // buffer-load path: the pointers never moved
        so += sstep;
        if (MODE == OP_XC) {
            const int kh = tid / XW;
            if (gp != nullptr) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (k0 + 8 * kh + i < kend) ? p[0][(long long)gi[i] * ld_] : 0.0f;
                advance_ids();
                return;
            }
            const float* q = use_buf ? src_ + (vo[0] >> 2) + adv : p[0];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (k0 + 8 * kh + i < kend) ? q[i * ld_] : 0.0f;
            if (!use_buf) p[0] += step;
        } else {
            const int kq = tid & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* q = use_buf ? src_ + (vo[i] >> 2) + adv : p[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * i + e] = (k0 + 4 * kq + e < kend) ? q[e] : 0.0f;
                if (!use_buf) p[i] += step;
            }
        }
    }
    __device__ __forceinline__ void load() {}                 // (the values are in registers once the loads have landed)
    __device__ __forceinline__ float sum8() const { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }
    __device__ __forceinline__ void commit(unsigned char* tile) const {
        unsigned w[3][4];
        split8(v, w);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            if (MODE == OP_XC) {
                *reinterpret_cast<uint4*>(tile + pl * PLANE + lds_ofs[0]) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
            } else {
                *reinterpret_cast<uint2*>(tile + pl * PLANE + lds_ofs[0]) = make_uint2(w[pl][0], w[pl][1]);
                *reinterpret_cast<uint2*>(tile + pl * PLANE + lds_ofs[1]) = make_uint2(w[pl][2], w[pl][3]);
            }
        }
    }
};

// An x-contiguous operand tile through LDS-DMA (k_gemm_bx3h, 512 threads, XT = 256): the register path above needs eight
// 4-byte loads per thread down k, and a CU takes one wave-level load per 16-26 cycles whatever its width -- 64 of the 80
// loads of a k tile carry 256 bytes each.  Here wave w owns columns 32 w ... 32 w + 31 of the tile: two
// `buffer_load_dwordx4 ... lds` per lane (lane = (k row % 8, x quad); eight k rows of 128 contiguous bytes per instruction)
// drop the raw fp32 block [16 k][32 x] into a 2 KiB LDS region of the wave's own, and the commit reads it back transposed
// (lane = (x, k half): eight ds_read_b32 down k, conflict-free), splits and writes the planes as before.  No other wave
// touches the region, so the only synchronisation is the wave's own vmcnt / lgkmcnt; the prefetched tile costs no registers.
// Needs ld % 4 == 0, X % 4 == 0, a 16-byte aligned base and the operand below 4 GiB (launch_t checks).
typedef __attribute__((address_space(3))) void* bx_lds_ptr_t;
template <int XT, bool GATHER = false>
struct BxDmaXC {
    static constexpr int KH = XT * 16, PLANE = 2 * KH;
    __amdgpu_buffer_rsrc_t rs;
    unsigned vo[2], so, sstep, ldb;
    unsigned char* raw;        // the wave's 2 KiB region (wave-uniform)
    int rd_ofs, lds_ofs, kvalid;
    float v[8];
    // GATHER (the two-part A of GemmArgs::m_split): row k of the operand is row gidx[k] of the table (gidx == nullptr: row k).  The row
    // ids of the NEXT tile are requested right behind this tile's DMA loads and are in by the time the following fetch() needs them
    // (load()'s vmcnt(0) sits in between); rows past the K range are clamped to its last row here and zeroed in load() as always.
    const int* gidx; int gk[2], gend; unsigned xofs, gnext[2];
    __device__ __forceinline__ unsigned grow(int i) { const int k = min(gk[i], gend); gk[i] += 16; return gidx ? (unsigned)gidx[k] : (unsigned)k; }
    __device__ __forceinline__ void init(const float* src, int ld, int X, int x0, const int* gather, int kb, int tid, int = 0, unsigned char* raw_ = nullptr, int kend = 0) {
        const int lane = tid & 63, wave = tid >> 6;
        raw = raw_;
        const int xcol = min(x0 + 32 * wave + 4 * (lane & 7), X - 4), krow = lane >> 3;
        ldb = (unsigned)ld * 4u;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0xfffffffcu, 0x00020000);
        if constexpr (GATHER) {
            gidx = gather; gend = kend - 1; xofs = (unsigned)xcol * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i) gk[i] = kb + 8 * i + krow;
#pragma unroll
            for (int i = 0; i < 2; ++i) vo[i] = grow(i) * ldb + xofs;
#pragma unroll
            for (int i = 0; i < 2; ++i) gnext[i] = grow(i);
            so = 0; sstep = 0;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) vo[i] = (unsigned)(kb + 8 * i + krow) * ldb + (unsigned)xcol * 4u;
            so = 0; sstep = 16u * ldb;
        }
        rd_ofs = ((lane >> 5) * 256 + (lane & 31)) * 4;
        lds_ofs = (lane >> 5) * KH + (32 * wave + (lane & 31)) * 16;
        kvalid = 16;
    }
    __device__ __forceinline__ void fetch() {
        const unsigned s0 = __builtin_amdgcn_readfirstlane(so);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bx_lds_ptr_t)raw, 16, vo[0], s0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bx_lds_ptr_t)(raw + 1024), 16, vo[1], s0, 0, 0);
        if constexpr (GATHER) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { vo[i] = gnext[i] * ldb + xofs; gnext[i] = grow(i); }
        } else {
            so += sstep;
        }
    }
    // last tile of a K range: rows from kend on are read from row kend - 1 and zeroed in load()
    __device__ __forceinline__ void fetch_partial(int k0, int kend, int tid) {
        if constexpr (GATHER) {
            fetch();                        // (clamped to row kend - 1 when the row ids were requested)
            kvalid = kend - k0;
            return;
        }
        const int krow = (tid & 63) >> 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {         // vo[i] + so addresses row k0 + 8 i + krow: step back to row kend - 1 from beyond it
            const int over = max(k0 + 8 * i + krow - (kend - 1), 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bx_lds_ptr_t)(raw + 1024 * i), 16, vo[i] + so - (unsigned)over * ldb, 0, 0, 0);
        }
        so += sstep;
        kvalid = kend - k0;
    }
    __device__ __forceinline__ void load() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* r = reinterpret_cast<const float*>(raw + rd_ofs);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = r[j * 32];
        if (kvalid < 16) {
            const int khalf8 = (rd_ofs >> 10) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (khalf8 + j >= kvalid) v[j] = 0.0f;
        }
    }
    __device__ __forceinline__ float sum8() const { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }
    __device__ __forceinline__ void commit(unsigned char* tile) const {
        unsigned w[3][4];
        split8(v, w);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(tile + pl * PLANE + lds_ofs) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    }
};
// An operand that arrives PRE-SPLIT (GemmArgs::Apl / Bpl, k_gemm_bx3h): the plane image [plane][k / 8][x][8 bf16] holds, per plane and
// k group, the 16-byte words of consecutive x one after the other -- which is the LDS image of a tile -- so a [256 x][16 k] tile is
// six runs of 4 KiB: 24 LDS-DMA instructions of 1 KiB, three per wave, straight into the stage buffer every wave reads its fragments
// from.  No registers, no VALU, no LDS instruction; what is left of the operand in the k loop is three vector-memory issues per wave.
// The stage buffer is SHARED (the raw regions of BxDmaXC are wave-private): a DMA may only be issued once every wave is past the
// barrier behind the stage's last reads, and must have landed (counted vmcnt) before the barrier that publishes the stage -- hence
// three stages for such an operand: tile kt + 2 is requested while tile kt is being multiplied.
template <int XT>
struct BxPlanes {
    static constexpr int KH = XT * 16, PLANE = 2 * KH;
    __amdgpu_buffer_rsrc_t rs;
    unsigned vo[3], so, sstep;
    int lofs[3];            // wave-uniform byte offsets of the wave's three 1 KiB runs inside an operand tile
    __device__ __forceinline__ void init(const void* planes, int K, int X, int x0, int kb, int tid) {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned k8 = 2u * (unsigned)((K + 15) / 16);
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(planes), 0, (int)0xfffffffcu, 0x00020000);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = wave * 3 + j, pl = idx >> 3, kh = (idx >> 2) & 1, q = idx & 3;
            const unsigned x = (unsigned)min(x0 + q * 64 + lane, X - 1);       // past X: the last column again (lands in columns that are never stored)
            vo[j] = ((pl * k8 + (unsigned)(kb >> 3) + kh) * (unsigned)X + x) * 16u;
            lofs[j] = pl * PLANE + kh * KH + q * 1024;
        }
        so = 0; sstep = 2u * (unsigned)X * 16u;
    }
    __device__ __forceinline__ void fetch(unsigned char* stage) {
        const unsigned s0 = __builtin_amdgcn_readfirstlane(so);
#pragma unroll
        for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (bx_lds_ptr_t)(stage + lofs[j]), 16, vo[j], s0, 0, 0);
        so += sstep;
    }
};
template <bool DMA, int MODE, int XT, int NT, bool BUF, bool GATHER = false> struct BxStagerSel { typedef BxStager<MODE, XT, NT, BUF> type; };
template <int XT, int NT, bool BUF, bool GATHER> struct BxStagerSel<true, OP_XC, XT, NT, BUF, GATHER> { typedef BxDmaXC<XT, GATHER> type; };

// PROF (tools/gemm_bench PROF=1): per (block, wave) s_memtime stamps -> g.prof[(block * 4 + wave) * 8 + ...]:
//   [0] entry, [1] first loop iteration, [2] sum over k tiles of (fragment reads + MFMA issue), [3] sum of (wait for the
//   next tile's loads + split + LDS writes), [4] sum of (LDS drain + barrier), [5] loop exit, [6] kernel exit,
//   [7] XCC_ID << 32 | HW_ID
#define BX_STAMP(i) if (PROF) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                                __builtin_amdgcn_sched_barrier(0); pacc[i] += n_ - plast; plast = n_; }
// BUFM: which operands are fetched with buffer loads (0 none, 1 B only, 2 both; BxStager)
template <int AMODE, int BMODE, bool PROF = false, int BUFM = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_bx3(const GemmArgs g) {
    constexpr int EPI = 4 * 32 * 68 * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(2 * BX_STAGE > EPI) ? 2 * BX_STAGE : EPI];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int tilesM = (g.M + 127) / 128, tilesN = (g.N + 127) / 128;
    const int bid = xcd_tile(blockIdx.x, tilesM * tilesN);
    int tm, tn; tile_coords(bid, tilesM, tilesN, g.group_m, tm, tn);
    const int z = blockIdx.y;
    const int m0 = tm * 128, n0 = tn * 128;
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = 0, p_entry = 0, p_loop = 0;
    if (PROF) { p_entry = plast = __builtin_amdgcn_s_memtime(); }

    int kb = 0, ke = g.K;
    if (g.ksplit > 1) {
        const int per = ((g.K + g.ksplit - 1) / g.ksplit + 15) / 16 * 16;
        kb = z * per;
        ke = min(g.K, kb + per);
    }
    const int nk = (ke > kb) ? (ke - kb + 15) / 16 : 0;
    const int nfull = (ke > kb) ? (ke - kb) / 16 : 0;

    BxStager<AMODE, 128, 256, (BUFM >= 2)> sa;
    BxStager<BMODE, 128, 256, (BUFM >= 1)> sb;
    sa.init(g.A, g.lda, g.M, m0, g.gather, kb, tid, g.K);
    sb.init(g.B, g.ldb, g.N, n0, nullptr, kb, tid, g.K);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const bool do_colsum = (BMODE == OP_XC) && g.colsum != nullptr && tm == 0;
    float csum = 0.0f;

    if (nk > 0) {
        if (nfull > 0) { sa.fetch(); sb.fetch(); }
        else { sa.fetch_partial(kb, ke, tid); sb.fetch_partial(kb, ke, tid); }
        if (do_colsum) csum += sb.sum8();
        sa.commit(smem);
        sb.commit(smem + BX_OPER);
    }
    __syncthreads();

    const int fa = khalf * 2048 + (wm * 64 + l31) * 16, fb = khalf * 2048 + (wn * 64 + l31) * 16;
    if (PROF) { p_loop = plast = __builtin_amdgcn_s_memtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            if (kt + 1 < nfull) { sa.fetch(); sb.fetch(); }
            else { sa.fetch_partial(kb + (kt + 1) * 16, ke, tid); sb.fetch_partial(kb + (kt + 1) * 16, ke, tid); }
        }
        const unsigned char* at = smem + cur * BX_STAGE;
        const unsigned char* bt = at + BX_OPER;
        bf16x8_t a[3][2], b[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[pl][i] = *reinterpret_cast<const bf16x8_t*>(at + pl * BX_PLANE + fa + i * 512);
                b[pl][i] = *reinterpret_cast<const bf16x8_t*>(bt + pl * BX_PLANE + fb + i * 512);
            }
        // smallest terms first; four independent accumulators per term keep the pipe full
#define BX_TERM(PA, PB)                                                                                        \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[0][0], 0, 0, 0);            \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[0][1], 0, 0, 0);            \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[1][0], 0, 0, 0);            \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[1][1], 0, 0, 0);
        BX_TERM(2, 0) BX_TERM(0, 2) BX_TERM(1, 1) BX_TERM(1, 0) BX_TERM(0, 1) BX_TERM(0, 0)
#undef BX_TERM
        BX_STAMP(2)
        if (more) {
            if (do_colsum) csum += sb.sum8();
            sa.commit(smem + (cur ^ 1) * BX_STAGE);
            sb.commit(smem + (cur ^ 1) * BX_STAGE + BX_OPER);
        }
        BX_STAMP(3)
        __syncthreads();
        BX_STAMP(4)
    }
    const unsigned long long p_exit_loop = PROF ? __builtin_amdgcn_s_memtime() : 0;

    float* smem_f = reinterpret_cast<float*>(smem);
    if (do_colsum) {                    // the two k halves of a column live in threads tid and tid + 128
        __shared__ float s_cs[128];
        if (tid >= 128) s_cs[tid - 128] = csum;
        __syncthreads();
        if (tid < 128 && n0 + tid < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + tid] = csum + s_cs[tid];
    }
    store_tile(g, acc, smem_f, z, m0, n0, tn, tilesN, wave, lane);
    if (PROF && g.prof != nullptr && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long p_end = __builtin_amdgcn_s_memtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        o[0] = p_entry; o[1] = p_loop; o[2] = pacc[2]; o[3] = pacc[3]; o[4] = pacc[4]; o[5] = p_exit_loop; o[6] = p_end;
        o[7] = ((unsigned long long)xcc << 32) | hw;
    }
}

// Wave-specialised variants (GemmArgs::bx3 == 2: MT = 1, 128 x 128 block tile; bx3 == 3: MT = 2, 256 x 128): the same LDS
// image per 128 rows, k order and term order as k_gemm_bx3 -- hence the same bits for the same K split -- but the block has
// 4 MT "multiplier" waves (64 x 64 each) that only read fragments and issue MFMAs, and 4 "loader" waves that only load, split
// and write the next tile (a workgroup's waves are dealt to the SIMDs cyclically, so every SIMD gets MT multipliers and one
// loader per block).  Why (tools/gemm_bench PROF=1, profiles/r03_gemm_prof*.log): in k_gemm_bx3 a lone 4-wave block needs
// 2470 cycles per k tile for 768 cycles of MFMAs -- reads, MFMAs, ~100 VALU instructions of the split, LDS writes and the
// barrier are one serial chain per wave, and three co-resident blocks only fill the matrix pipe to 70 %.  With the roles
// split the multipliers' chain is reads + MFMAs (1060-1300 cycles), and the split issues in the MFMA shadow of other waves;
// it then is the VALU that binds (31 cycles per value: 16 values per thread and k tile = 500 of the 768 cycles per multiplier
// wave at MT = 1), which is what MT = 2 relieves: a 256 x 128 tile splits 0.75 x the values per MFMA.
// The loaders fetch two tiles ahead (two register sets taking alternate tiles).
__device__ __forceinline__ void bx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int AMODE, int BMODE, int MT, bool PROF = false, int BUFM = 0>
__global__ __launch_bounds__(256 * (MT + 1), MT == 1 ? 2 : 1) void k_gemm_bx3w(const GemmArgs g) {
    constexpr int XA = 128 * MT;                       // rows of the block tile
    constexpr int A_PLANE = 2 * XA * 16, A_OPER = 3 * A_PLANE, STAGE = A_OPER + BX_OPER;
    constexpr int EPI = 4 * MT * 32 * 68 * 4;
    constexpr int NMW = 4 * MT;                        // multiplier waves
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(2 * STAGE > EPI) ? 2 * STAGE : EPI];
    __shared__ float s_cs[128];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + XA - 1) / XA, tilesN = (g.N + 127) / 128;
    const int bid = xcd_tile(blockIdx.x, tilesM * tilesN);
    int tm, tn; tile_coords(bid, tilesM, tilesN, g.group_m, tm, tn);
    const int z = blockIdx.y;
    const int m0 = tm * XA, n0 = tn * 128;
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = 0, p_entry = 0, p_loop = 0, p_exit_loop = 0;
    if (PROF) { p_entry = plast = __builtin_amdgcn_s_memtime(); }

    int kb = 0, ke = g.K;
    if (g.ksplit > 1) {
        const int per = ((g.K + g.ksplit - 1) / g.ksplit + 15) / 16 * 16;
        kb = z * per;
        ke = min(g.K, kb + per);
    }
    const int nk = (ke > kb) ? (ke - kb + 15) / 16 : 0;
    const int nfull = (ke > kb) ? (ke - kb) / 16 : 0;

    if (wave >= NMW) {
        // ---------------------------------------------------------------- loaders: tile kt + 1 is split and written while
        // the others multiply tile kt; its values were requested two iterations earlier
        const int lt = tid - 64 * NMW;
        BxStager<AMODE, XA, 256, (BUFM >= 2)> sa[2][MT];
        BxStager<BMODE, 128, 256, (BUFM >= 1)> sb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < MT; ++i) sa[q][i].init(g.A, g.lda, g.M, m0 + 128 * i, g.gather, kb + 16 * q, lt, g.K, 32, 128 * i);
            sb[q].init(g.B, g.ldb, g.N, n0, nullptr, kb + 16 * q, lt, g.K, 32);
        }
        if (PROF && (g.dbg & 4)) {              // diagnostics: every k tile re-reads the block's FIRST tile (L1 / L2 resident)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < MT; ++i) { sa[q][i].step = 0; sa[q][i].sstep = 0; }
                sb[q].step = 0; sb[q].sstep = 0;
            }
        }
        const bool do_colsum = (BMODE == OP_XC) && g.colsum != nullptr && tm == 0;
        float csum = 0.0f;
#define BXW_FETCH(Q, T)                                                                                        \
        if ((T) < nfull) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) sa[Q][i].fetch();                                   \
            sb[Q].fetch();                                                                                     \
        } else {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) sa[Q][i].fetch_partial(kb + (T) * 16, ke, lt);      \
            sb[Q].fetch_partial(kb + (T) * 16, ke, lt);                                                        \
        }
#define BXW_COMMIT(Q, ST)                                                                                      \
        if (do_colsum) csum += sb[Q].sum8();                                                                   \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) sa[Q][i].commit(smem + (ST) * STAGE);                   \
        sb[Q].commit(smem + (ST) * STAGE + A_OPER);
        // diagnostics (PROF instantiation only; results are wrong): g.dbg & 1 = no global loads inside the loop,
        // g.dbg & 2 = no split / LDS writes inside the loop
        const bool ld_on = !(PROF && (g.dbg & 1)), cm_on = !(PROF && (g.dbg & 2));
        if (nk > 0) {
            BXW_FETCH(0, 0)
            if (nk > 1) { BXW_FETCH(1, 1) }
            BXW_COMMIT(0, 0)
            if (nk > 2) { BXW_FETCH(0, 2) }
        }
        bx_barrier();
        if (PROF) { p_loop = plast = __builtin_amdgcn_s_memtime(); }
        // stamped instantiation: [2] = the wait for the tile about to be split (the loads of the tile after it are younger)
        constexpr int NLD = MT * (AMODE == OP_KC ? 2 : 8) + (BMODE == OP_KC ? 2 : 8);
        for (int kt = 0; kt < nk; kt += 2) {
            if (PROF && kt + 3 < nk && g.gather == nullptr) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory"); BX_STAMP(2) }
            if (kt + 1 < nk) {                      // tile kt + 1 sits in set 1
                if (cm_on) { BXW_COMMIT(1, 1) }
                if (kt + 3 < nk && ld_on) { BXW_FETCH(1, kt + 3) }
            }
            BX_STAMP(3)
            bx_barrier();
            BX_STAMP(4)
            if (kt + 1 >= nk) break;
            if (PROF && kt + 4 < nk && g.gather == nullptr) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory"); BX_STAMP(2) }
            if (kt + 2 < nk) {                      // tile kt + 2 sits in set 0
                if (cm_on) { BXW_COMMIT(0, 0) }
                if (kt + 4 < nk && ld_on) { BXW_FETCH(0, kt + 4) }
            }
            BX_STAMP(3)
            bx_barrier();
            BX_STAMP(4)
        }
#undef BXW_FETCH
#undef BXW_COMMIT
        if (PROF) p_exit_loop = __builtin_amdgcn_s_memtime();
        if (do_colsum && lt >= 128) s_cs[lt - 128] = csum;
        bx_barrier();
        if (do_colsum && lt < 128 && n0 + lt < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + lt] = csum + s_cs[lt];
    } else {
        // ---------------------------------------------------------------- multipliers
        const int wm = wave >> 1, wn = wave & 1;
        const int l31 = lane & 31, khalf = lane >> 5;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        const int fa = khalf * (XA * 16) + (wm * 64 + l31) * 16, fb = khalf * 2048 + (wn * 64 + l31) * 16;
        bx_barrier();
        if (PROF) { p_loop = plast = __builtin_amdgcn_s_memtime(); }
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char* at = smem + (kt & 1) * STAGE;
            const unsigned char* bt = at + A_OPER;
            bf16x8_t a[3][2], b[3][2];
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)            // the planes of the first terms first
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[pl][i] = *reinterpret_cast<const bf16x8_t*>(at + pl * A_PLANE + fa + i * 512);
                    b[2 - pl][i] = *reinterpret_cast<const bf16x8_t*>(bt + (2 - pl) * BX_PLANE + fb + i * 512);
                }
#define BX_TERM(PA, PB)                                                                                        \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][0], acc[0][0], 0, 0, 0);            \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][0], b[PB][1], acc[0][1], 0, 0, 0);            \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][0], acc[1][0], 0, 0, 0);            \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][1], b[PB][1], acc[1][1], 0, 0, 0);
            BX_TERM(2, 0) BX_TERM(0, 2) BX_TERM(1, 1) BX_TERM(1, 0) BX_TERM(0, 1) BX_TERM(0, 0)
#undef BX_TERM
            BX_STAMP(2)
            bx_barrier();
            BX_STAMP(4)
        }
        if (PROF) p_exit_loop = __builtin_amdgcn_s_memtime();
        bx_barrier();
        store_tile(g, acc, reinterpret_cast<float*>(smem), z, m0, n0, tn, tilesN, wave, lane);
    }
    if (PROF && g.prof != nullptr && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long p_end = __builtin_amdgcn_s_memtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (NMW + 4) + wave) * 8;
        o[0] = p_entry; o[1] = p_loop; o[2] = pacc[2]; o[3] = pacc[3]; o[4] = pacc[4]; o[5] = p_exit_loop; o[6] = p_end;
        o[7] = ((unsigned long long)xcc << 32) | hw;
    }
}

// ---- 256 x 256 block tile, eight waves of 128 x 64 (GemmArgs::bx3 == 3) -------------------------------------------------------
// Per MFMA this tile needs half of everything the 128 x 128 kernels need beside the matrix pipe: 32 KiB of operands, 16 values
// to split per thread and 18 fragment reads per 48 MFMAs of a wave (2.6 issued instructions per MFMA instead of 5.0, 10.7
// instead of 21 bytes per cycle and CU at the pipe's full rate).  One block per CU, two waves per SIMD -- waves w and w + 4 --
// and those two run the k tile in OPPOSITE order so that they do not both want the matrix pipe right behind the barrier:
// waves 0-3 read their fragments, multiply, then split and write their share of the next tile; waves 4-7 split and write
// first (their loads were issued a whole iteration earlier), then read and multiply.  Same LDS image per 128 rows / columns,
// k order and term order as k_gemm_bx3: the same bits for the same K split.
// QUEUE (GemmArgs::xcd_first != 0): the tile comes from the work queue of k_gemm_queue instead of blockIdx -- a restricted launch
// (xcd_first > 0) lets only blocks on XCDs >= xcd_first draw, only items below work_limit and only while *stop == 0; the clean-up
// launch (xcd_first < 0) takes what nobody claimed.  Which block computes an item never changes the item.
// AG: two-part op(A) (GemmArgs::m_split; x-contiguous operands through LDS-DMA only)
// PLM: which operands arrive pre-split as plane images (GemmArgs::Apl / Bpl; bit 0: A, bit 1: B) -- BxPlanes, three LDS stages each
template <int AMODE, int BMODE, bool PROF = false, int BUFM = 0, bool QUEUE = false, bool AG = false, int PLM = 0>
__global__ __launch_bounds__(512, 1) void k_gemm_bx3h(const GemmArgs g) {
    constexpr int XT = 256;
    constexpr int PLANE = 2 * XT * 16, OPER = 3 * PLANE;                              // 8 KiB, 24 KiB
    constexpr int EPI = 8 * 32 * 68 * 4;
    constexpr bool DMA = BUFM >= 3;                       // x-contiguous operands through LDS-DMA (BxDmaXC)
    constexpr bool APL = (PLM & 1) != 0, BPL = (PLM & 2) != 0;
    constexpr int NSA = APL ? 3 : 2, NSB = BPL ? 3 : 2;   // LDS stages per operand
    constexpr int A_BYTES = NSA * OPER, PIPE = (NSA + NSB) * OPER;                    // 96 / 120 / 144 KiB
    static_assert(PIPE >= EPI, "the epilogue's transpose slices live in the pipeline's LDS");
    constexpr int RAW = DMA ? ((AMODE == OP_XC && !APL) + (BMODE == OP_XC && !BPL)) * 8 * 2048 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[PIPE + RAW];
    __shared__ float s_cs[256];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 2) & 1, wn = wave & 3;      // wave tile: rows 128 wm ..., columns 64 wn ...
    const bool late = wave >= 4;                          // order of the k tile, see above
    const int l31 = lane & 31, khalf = lane >> 5;
    const int tilesM = (g.M + XT - 1) / XT, tilesN = (g.N + XT - 1) / XT;
    int tm, tn, z;
    if (QUEUE) {
        __shared__ int s_item;
        const int total = tilesM * tilesN * (g.ksplit > 1 ? g.ksplit : 1);
        if (g.xcd_first > 0) {
            int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            // below xcd_first: the XCDs of the kernel this launch runs beside.  A block can only be placed there before that kernel
            // has taken its CUs (leave at once) or after it has left them (gated launches: join the drawing when it is really over)
            if ((xcc & 7) < g.xcd_first) {
#ifdef FSMG_EXPERIMENTS
                if (g.dbg & 64) return;
#endif
                if (g.gate == nullptr) return;
                // ONE thread decides for the block: the counter may be completing while the block reads it, and a block whose threads
                // disagree would go on with half its waves (first version: half-computed tiles, one step in three)
                if (tid == 0) s_item = __hip_atomic_load(g.gate + g.gate_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= g.gate_expect;
                __syncthreads();
                const int join = s_item;
                __syncthreads();
                if (!join) return;
            }
        }
        if (tid == 0) {
            int item = -1;
            if (g.xcd_first < 0) {
                const int j = atomicAdd(g.work + 1, 1);
                if (j < total && atomicCAS(g.claim + j, 0, 1) == 0) item = j;
            } else if (__hip_atomic_load(g.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                const int j = atomicAdd(g.work, 1);
                if (j < min(g.work_limit, total) && atomicCAS(g.claim + j, 0, 1) == 0) item = j;
            }
            if (item >= 0 && g.gate != nullptr) {          // wait for the last time step the tile's rows belong to
                const int tm_ = (item / tilesN) % tilesM;
                int t_need = (min(g.M, (tm_ + 1) * XT) - 1) / g.gate_rows;
                if (g.gate_every > 1) t_need = min(g.gate_last, (t_need / g.gate_every + 1) * g.gate_every - 1);
                const int spin_cap = g.gate_spin;
                for (int spins = 0; __hip_atomic_load(g.gate + t_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.gate_expect; ++spins) {
                    __builtin_amdgcn_s_sleep(32);
                    if (spins >= spin_cap || ((spins & 63) == 63 && __hip_atomic_load(g.gate_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
                        __hip_atomic_store(g.gate_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        item = -1;
                        break;
                    }
                }
            }
            s_item = item;
        }
        __syncthreads();
        const int item = s_item;
        if (item < 0) return;
        // The rows were written (write-through) by another XCD while this launch was running, and are read with ordinary loads:
        // this XCD's L2 was invalidated when the launch started (the acquire of every kernel dispatch -- what makes any producer /
        // consumer pair of kernels on different XCDs work), and a row enters it only through loads issued behind that row's gate.
        // Measured alternatives (profiles/r04_xov_*.log): an agent-scope acquire here empties the L2 for every block of the XCD
        // (920 times in 0.5 ms; 105 instead of 85 us per tile), agent-scope (sc1) loads of the operand bypass the L2 (the same 105 us).
#ifdef FSMG_EXPERIMENTS
        if (g.gate != nullptr && (g.dbg & 128)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (the measured alternative)
#endif
        tm = (item / tilesN) % tilesM; tn = item % tilesN; z = item / (tilesM * tilesN);
    } else {
        const int bid = xcd_tile(blockIdx.x, tilesM * tilesN);
        tile_coords(bid, tilesM, tilesN, g.group_m, tm, tn);
        z = blockIdx.y;
    }
    const int m0 = tm * XT, n0 = tn * XT;
    unsigned long long pacc[5] = {0, 0, 0, 0, 0}, plast = 0, p_entry = 0, p_loop = 0;
    if (PROF && (g.dbg & 16) && blockIdx.x < 256 && blockIdx.y == 0) {      // experiment: first-round blocks start staggered
        const int n = ((blockIdx.x >> 3) & 7) * (g.dbg >> 8);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }
    if (PROF) { p_entry = plast = __builtin_amdgcn_s_memtime(); }

    int kb = 0, ke = g.K;
    if (g.ksplit > 1) {
        const int per = ((g.K + g.ksplit - 1) / g.ksplit + 15) / 16 * 16;
        kb = z * per;
        ke = min(g.K, kb + per);
    }
    const int nk = (ke > kb) ? (ke - kb + 15) / 16 : 0;
    const int nfull = (ke > kb) ? (ke - kb) / 16 : 0;

    typename std::conditional<APL, BxPlanes<XT>, typename BxStagerSel<DMA, AMODE, XT, 512, (BUFM >= 2), AG>::type>::type sa;      // 512 threads cover 256 rows / columns, 8 values each
    typename std::conditional<BPL, BxPlanes<XT>, typename BxStagerSel<DMA, BMODE, XT, 512, (BUFM >= 1)>::type>::type sb;
    static_assert(!(AG && PLM != 0), "two-part A: not with plane images");
    if constexpr (APL) sa.init(g.Apl, g.K, g.M, m0, kb, tid);
    else if constexpr (AG) {                              // which part of op(A) this row tile belongs to
        static_assert(!AG || (DMA && AMODE == OP_XC), "two-part A: LDS-DMA staged x-contiguous operand");
        const bool first = m0 < g.m_split;
        sa.init(first ? g.A : g.A2, first ? g.lda : g.lda2, first ? g.m_split : g.M - g.m_split, first ? m0 : m0 - g.m_split,
                first ? g.gather : nullptr, kb, tid, g.K, smem + PIPE + wave * 2048, ke);
    } else
    if constexpr (DMA && AMODE == OP_XC) sa.init(g.A, g.lda, g.M, m0, nullptr, kb, tid, g.K, smem + PIPE + wave * 2048);
    else sa.init(g.A, g.lda, g.M, m0, g.gather, kb, tid, g.K);
#ifdef FSMG_EXPERIMENTS
    if constexpr (!APL && QUEUE && AMODE == OP_KC && BUFM >= 2) sa.coherent = g.gate != nullptr && (g.dbg & 32) != 0;     // (the measured alternative: agent-scope loads)
#endif
    if constexpr (BPL) sb.init(g.Bpl, g.K, g.N, n0, kb, tid);
    else if constexpr (DMA && BMODE == OP_XC) sb.init(g.B, g.ldb, g.N, n0, nullptr, kb, tid, g.K, smem + PIPE + ((AMODE == OP_XC && !APL) ? 8 * 2048 : 0) + wave * 2048);
    else sb.init(g.B, g.ldb, g.N, n0, nullptr, kb, tid, g.K);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // column sums of op(B): thread = (x = tid % 256, k half = tid / 256) of the XC stager
    const bool do_colsum = !BPL && (BMODE == OP_XC) && g.colsum != nullptr && tm == 0;
    float csum = 0.0f;
    // weighted column sums (GemmArgs::colsum_w): ONE dword load per wave and k tile -- lane l takes the weight of K row 8 * (its k half)
    // + l % 8, requested with the tile's loads a k tile ahead -- and the commit's eight FMAs read it through DPP row broadcasts
    // (row_newbcast:j: lane j of the own row of 16).  What was measured against it on the 400 us dW launch: two float4 loads per lane at
    // the commit +46 us (latency in the open), the same a k tile ahead +130 us (two more 1 KiB vector loads per wave beside the four of
    // the LDS-DMA staging; a CU issues one wave-level load per ~20 cycles), s_load_dwordx16 through the scalar cache +46 us (an SMEM
    // request in flight turns every counted LDS wait of the k loop and its barrier into lgkmcnt(0)).  The request is UNCONDITIONAL in
    // the kernels that can be asked for weights (both operands x-contiguous: the weight-gradient shapes); a launch without weights reads
    // floats of B it never uses.
    constexpr bool WCS = AMODE == OP_XC && BMODE == OP_XC && !BPL;
    const bool cs_weighted = WCS && do_colsum && g.colsum_w != nullptr;
    const float* cs_p = (WCS && g.colsum_w != nullptr ? g.colsum_w : g.B) + kb + 8 * ((DMA && BMODE == OP_XC) ? (lane >> 5) : (tid >> 8)) + (lane & 7);
    float cs_wv = 0.0f;
    unsigned char* const smemB = smem + A_BYTES;
    // vector-memory instructions a full tile's requests issue per wave (plane images first, then the operand that is split in the loop):
    // what may still be in flight -- tile kt + 2 -- when tile kt + 1's LDS-DMA must have landed (the counted wait in front of the barrier)
    constexpr int NLD_A = APL ? 3 : (AMODE == OP_KC ? 2 : (DMA ? 2 : 8)), NLD_B = BPL ? 3 : (BMODE == OP_KC ? 2 : (DMA ? 2 : 8));
    constexpr int NLD = NLD_A + NLD_B + (WCS ? 1 : 0);
    // T: tile index, S3: its stage among three (plane images only).  The DMAs of the plane images go out FIRST: whatever waits for the
    // loads of the other operand (its commit) has then waited for them too (loads return in order).
#define BXH_FETCH(T, S3)                                                                                       \
    if constexpr (APL) sa.fetch(smem + (S3) * OPER);                                                           \
    if constexpr (BPL) sb.fetch(smemB + (S3) * OPER);                                                          \
    if ((T) < nfull) { if constexpr (!APL) sa.fetch(); if constexpr (!BPL) sb.fetch(); }                       \
    else { if constexpr (!APL) sa.fetch_partial(kb + (T) * 16, ke, tid); if constexpr (!BPL) sb.fetch_partial(kb + (T) * 16, ke, tid); } \
    if constexpr (WCS) cs_wv = cs_p[(T) * 16];
#define BXH_COMMIT(ST)                                                                                         \
    if constexpr (!APL) sa.load();                                                                             \
    if constexpr (!BPL) {                                                                                      \
    sb.load();                                                                                                 \
    if (do_colsum) {                                                                                           \
        if (cs_weighted) {               /* weighted per K row: cs_wv came in with this tile's loads (BXH_FETCH) */ \
            csum = fmaf(sb.v[0], cs_bcast<0>(cs_wv), csum); csum = fmaf(sb.v[1], cs_bcast<1>(cs_wv), csum);   \
            csum = fmaf(sb.v[2], cs_bcast<2>(cs_wv), csum); csum = fmaf(sb.v[3], cs_bcast<3>(cs_wv), csum);   \
            csum = fmaf(sb.v[4], cs_bcast<4>(cs_wv), csum); csum = fmaf(sb.v[5], cs_bcast<5>(cs_wv), csum);   \
            csum = fmaf(sb.v[6], cs_bcast<6>(cs_wv), csum); csum = fmaf(sb.v[7], cs_bcast<7>(cs_wv), csum);   \
        } else csum += sb.sum8();                                                                              \
    }                                                                                                          \
    }                                                                                                          \
    if constexpr (!APL) sa.commit(smem + (ST) * OPER);                                                         \
    if constexpr (!BPL) sb.commit(smemB + (ST) * OPER);
    // the wait for tile T1's plane-image DMAs in front of the barrier that publishes it: tile T1 + 1's requests (NLD of them, when it
    // is a full tile: a K tail's loads are predicated and not counted on) may stay in flight
#define BXH_WAIT_PLANES(T1)                                                                                    \
    if constexpr (PLM != 0) {                                                                                  \
        if ((T1) + 1 < nfull) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");                      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    }

    if (nk > 0) {
        BXH_FETCH(0, 0)
        BXH_COMMIT(0)
        if (nk > 1) { BXH_FETCH(1, 1) }                   // every fetch sits right behind a commit (the VALU half of an iteration)
        BXH_WAIT_PLANES(0)
    }
    bx_barrier();
    const int fa = khalf * (XT * 16) + (wm * 128 + l31) * 16, fb = khalf * (XT * 16) + (wn * 64 + l31) * 16;
    if (PROF) { p_loop = plast = __builtin_amdgcn_s_memtime(); }
    int s3 = 0;                                           // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const int s3n = s3 == 0 ? 2 : s3 - 1;             // (kt + 2) % 3: the stage tile kt - 1 was read from
        // every wave starts with the fragments of the FIRST term (a[2], b[0]: 6 of the 18 reads), so that the late waves'
        // first MFMAs do not wait for LDS behind their commit; the other twelve follow the commit and land behind those
        // MFMAs.  (All 18 up front cost 72 live registers during the split: 10 spilled with two x-contiguous operands, and
        // the stagers' load offsets spilled -- reloaded from scratch in front of every load -- with LDS-DMA.)  Two
        // k-contiguous operands have the registers for all 18 and are faster that way (3670 against 3970 cycles per k tile).
        constexpr bool READS_ALL_FIRST = (AMODE == OP_KC && BMODE == OP_KC) || PLM != 0;
        const unsigned char* at = smem + (APL ? s3 : (kt & 1)) * OPER;
        const unsigned char* bt = smemB + (BPL ? s3 : (kt & 1)) * OPER;
        bf16x8_t a[3][4], b[3][2];
        // both operands as plane images: nothing to split, the requests of tile kt + 2 go out at the top of the iteration
        if constexpr (PLM == 3) { if (kt + 2 < nk) { BXH_FETCH(kt + 2, s3n) } }
#define BXH_READ_A(pl) _Pragma("unroll") for (int i = 0; i < 4; ++i) a[pl][i] = *reinterpret_cast<const bf16x8_t*>(at + (pl) * PLANE + fa + i * 512);
#define BXH_READ_B(pl) _Pragma("unroll") for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const bf16x8_t*>(bt + (pl) * PLANE + fb + j * 512);
        BXH_READ_A(2) BXH_READ_B(0)
        if (READS_ALL_FIRST) { BXH_READ_A(0) BXH_READ_B(2) BXH_READ_A(1) BXH_READ_B(1) }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PLM != 3) if (late && more) {
            BXH_COMMIT((kt + 1) & 1)
            if (kt + 2 < nk) { BXH_FETCH(kt + 2, s3n) }
        }
        __builtin_amdgcn_sched_barrier(0);
        BX_STAMP(3)
        if (!READS_ALL_FIRST) { BXH_READ_A(0) BXH_READ_B(2) BXH_READ_A(1) BXH_READ_B(1) }
#undef BXH_READ_A
#undef BXH_READ_B
#define BXH_TERM(PA, PB)                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0);
        BXH_TERM(2, 0) BXH_TERM(0, 2) BXH_TERM(1, 1) BXH_TERM(1, 0) BXH_TERM(0, 1) BXH_TERM(0, 0)
#undef BXH_TERM
        BX_STAMP(2)
        if constexpr (PLM != 3) if (!late && more) {
            BXH_COMMIT((kt + 1) & 1)
            if (kt + 2 < nk) { BXH_FETCH(kt + 2, s3n) }
        }
        BX_STAMP(3)
        if (more) { BXH_WAIT_PLANES(kt + 1) }
        bx_barrier();
        BX_STAMP(4)
        s3 = s3 == 2 ? 0 : s3 + 1;
    }
#undef BXH_FETCH
#undef BXH_COMMIT
#undef BXH_WAIT_PLANES
    const unsigned long long p_exit_loop = PROF ? __builtin_amdgcn_s_memtime() : 0;

    if (do_colsum) {                    // the two k halves of a column live in threads tid and tid + 256 (LDS-DMA: lanes l, l + 32)
        const int col = (DMA && BMODE == OP_XC) ? 32 * wave + (lane & 31) : (tid & 255);
        const bool upper = (DMA && BMODE == OP_XC) ? lane >= 32 : tid >= 256;
        if (upper) s_cs[col] = csum;
        bx_barrier();
        if (!upper && n0 + col < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + col] = csum + s_cs[col];
    }
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    if (PROF && (g.dbg & 8)) {          // experiment: no output (what the store tail costs)
        float sink = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
        if (sink == 1.2345e-30f) g.C[0] = sink;
    } else
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {    // the wave's 128 x 64 tile as two 64 x 64 halves
        // a 256-column tile may reach past the last 128-column tile of N: nothing to store there, and the forward-only cross
        // entropy has no partial slot for it (a write would land in the next row's slots)
        if ((n0 + wn * 64) / 128 >= (g.N + 127) / 128) break;
        f32x16 (&sub)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&acc[2 * h2][0]);
        store_tile_at(g, sub, ep, z, m0 + wm * 128 + h2 * 64, n0 + wn * 64, (n0 + wn * 64) / 128, (g.N + 127) / 128, wn & 1, lane);
    }
#ifdef FSMG_EXPERIMENTS
    if constexpr (QUEUE) {
        if (g.done != nullptr) {            // the tile is complete: every wave's (write-through) stores acknowledged, then count it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bx_barrier();
            if (tid == 0) __hip_atomic_fetch_add(g.done + tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
    if (PROF && g.prof != nullptr && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long p_end = __builtin_amdgcn_s_memtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
        o[0] = p_entry; o[1] = p_loop; o[2] = pacc[2]; o[3] = pacc[3]; o[4] = pacc[4]; o[5] = p_exit_loop; o[6] = p_end;
        o[7] = ((unsigned long long)xcc << 32) | hw;
    }
}
#undef BX_STAMP

template <int AMODE, int BMODE>
hipError_t launch_t(hipStream_t s, const GemmArgs& g, int lds_pad) {
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    // weighted column sums: the 256 x 256-tile kernels' x-contiguous-operand instantiations only; exp(logit) stores: the bf16-split kernels
    if (g.colsum_w != nullptr && !(g.bx3 == 3 && AMODE == OP_XC && BMODE == OP_XC && g.colsum != nullptr)) return hipErrorInvalidValue;
    if (g.ce_store && !(g.bx3 != 0 && g.ce_part != nullptr && g.ksplit <= 1)) return hipErrorInvalidValue;      // (every bf16-split kernel: one epilogue)
    // Pre-split operands (plane images).  MEASURED AND REJECTED as a product path (round 6, profiles/r06_gemm_planes_rejected.txt): the same
    // bits, and no faster -- the k loop is bound by the matrix pipe its two waves per SIMD share plus the barrier / first-fragment latency of
    // every k tile, not by the split (dH 265 -> 272 us with the weights' image, 274 with both images, 285 with the 1.5 x larger image of
    // E' alone; the projection 353 -> 351 / 343).  The instantiations live in the experiment build (make experiments, tools/gemm_bench PL=).
    if (g.Apl != nullptr || g.Bpl != nullptr) {
#ifndef FSMG_EXPERIMENTS
        return hipErrorInvalidValue;
#else
        const long long a_b = 4LL * g.lda * (AMODE == OP_KC ? g.M : g.K), b_b = 4LL * g.ldb * (BMODE == OP_KC ? g.N : g.K);
        const bool a_ok = g.Apl != nullptr ? plane_image_bytes(g.K, g.M) < 0xfffff000LL : (g.gather == nullptr && a_b < 0xfffff000LL && AMODE == OP_KC);
        const bool b_ok = g.Bpl != nullptr ? plane_image_bytes(g.K, g.N) < 0xfffff000LL : (b_b < 0xfffff000LL && BMODE == OP_KC);
        if (g.bx3 != 3 || !a_ok || !b_ok || g.prof != nullptr || g.m_split > 0 || (g.Bpl != nullptr && g.colsum != nullptr) ||
            (((uintptr_t)g.Apl | (uintptr_t)g.Bpl) & 15) != 0) return hipErrorInvalidValue;
        const int plm = (g.Apl != nullptr ? 1 : 0) | (g.Bpl != nullptr ? 2 : 0);
        const int total = ((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.ksplit > 1 ? g.ksplit : 1);
        if (g.xcd_first != 0) {                      // work-queue launch (the gated projection: A split in the loop, B = the weights' image)
            if (g.work == nullptr || g.claim == nullptr || (g.xcd_first > 0 && g.stop == nullptr) || plm != 2) return hipErrorInvalidValue;
            const int lim = g.work_limit < total ? g.work_limit : total;
            const int blocks = g.xcd_first > 0 ? (int)(((long long)lim * 8 + 7 - g.xcd_first) / (8 - g.xcd_first)) + 64 : total;
            if constexpr (AMODE == OP_KC) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2, true, false, 2>), dim3(blocks), dim3(512), 0, s, g); return hipGetLastError(); }
            return hipErrorInvalidValue;
        }
        dim3 grid3(((g.M + 255) / 256) * ((g.N + 255) / 256), g.ksplit > 1 ? g.ksplit : 1);
        if (plm == 3) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2, false, false, 3>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
        if constexpr (AMODE == OP_KC) {
            if (plm == 2) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2, false, false, 2>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
        }
        if constexpr (BMODE == OP_KC) {
            if (plm == 1) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2, false, false, 1>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
        }
        return hipErrorInvalidValue;
#endif
    }
    if (g.xcd_first != 0 && g.bx3 == 3) {      // work-queue launch of the 256 x 256-tile kernel (one block per CU)
        if (g.work == nullptr || g.claim == nullptr || (g.xcd_first > 0 && g.stop == nullptr) || g.gather != nullptr || g.prof != nullptr) return hipErrorInvalidValue;
        const long long a_b = 4LL * g.lda * (AMODE == OP_KC ? g.M : g.K), b_b = 4LL * g.ldb * (BMODE == OP_KC ? g.N : g.K);
        if (a_b >= 0xfffff000LL || b_b >= 0xfffff000LL) return hipErrorInvalidValue;
        const int total = ((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.ksplit > 1 ? g.ksplit : 1);
        const int lim = g.work_limit < total ? g.work_limit : total;
        const int blocks = g.xcd_first > 0 ? (int)(((long long)lim * 8 + 7 - g.xcd_first) / (8 - g.xcd_first)) + 64 : total;
        const bool a_dma = AMODE != OP_XC || (g.lda % 4 == 0 && g.M % 4 == 0 && ((uintptr_t)g.A & 15) == 0);
        const bool b_dma = BMODE != OP_XC || (g.ldb % 4 == 0 && g.N % 4 == 0 && ((uintptr_t)g.B & 15) == 0);
        if constexpr (AMODE == OP_XC && BMODE == OP_XC) {
            if (g.m_split > 0) {       // two-part op(A) from the queue (round 6: dKx + dKh of the layer above beside a BPTT chain)
                const bool ok = g.m_split % 256 == 0 && g.m_split < g.M && g.A2 != nullptr && b_dma && g.lda % 4 == 0 && g.lda2 % 4 == 0 &&
                                (g.M - g.m_split) % 4 == 0 && (((uintptr_t)g.A | (uintptr_t)g.A2) & 15) == 0 && 4LL * g.lda2 * g.K < 0xfffff000LL;
                if (!ok) return hipErrorInvalidValue;
                hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 3, true, true>), dim3(blocks), dim3(512), 0, s, g);
                return hipGetLastError();
            }
            if (a_dma && b_dma) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 3, true>), dim3(blocks), dim3(512), 0, s, g); return hipGetLastError(); }
        }
        if (g.m_split > 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2, true>), dim3(blocks), dim3(512), 0, s, g);
        return hipGetLastError();
    }
    if (g.xcd_first != 0) {             // work-queue launch: one resident set of blocks (4 per CU), see k_gemm_queue
        if (g.work == nullptr || g.claim == nullptr || (g.xcd_first > 0 && g.stop == nullptr) || (AMODE == OP_XC && g.gather != nullptr)) return hipErrorInvalidValue;
        const int total = tilesM * tilesN * (g.ksplit > 1 ? g.ksplit : 1);
        // a restricted launch needs as many blocks on the XCDs that may draw as there are items below its limit
        const int lim = g.work_limit < total ? g.work_limit : total;
        const int blocks = g.xcd_first > 0 ? (int)(((long long)lim * 8 + 7 - g.xcd_first) / (8 - g.xcd_first)) + 64 : total;
        hipLaunchKernelGGL((k_gemm_queue<AMODE, BMODE>), dim3(blocks), dim3(NTHREADS), lds_pad, s, g);
        return hipGetLastError();
    }
    dim3 grid(tilesM * tilesN, g.ksplit > 1 ? g.ksplit : 1);
    // buffer loads (32-bit offsets) for the operands that fit 4 GiB and whose rows are not gathered along K; a gathered KC
    // operand (embedding rows by token id) keeps 64-bit addresses too: its extent is the table's, which this call does not know
    const long long a_bytes = 4LL * g.lda * (AMODE == OP_KC ? g.M : g.K), b_bytes = 4LL * g.ldb * (BMODE == OP_KC ? g.N : g.K);
    const bool a_buf = g.gather == nullptr && a_bytes < 0xfffff000LL, b_buf = b_bytes < 0xfffff000LL;
    static const bool buf_off = std::getenv("FSMG_GEMM_BUF") && std::atoi(std::getenv("FSMG_GEMM_BUF")) == 0;      // A/B runs
    const int bufm = (buf_off && g.prof == nullptr) ? 0 : (a_buf && b_buf) ? 2 : (b_buf ? 1 : 0);
#ifdef FSMG_EXPERIMENTS
    if (g.prof != nullptr && bufm != 2) return hipErrorInvalidValue;     // the stamped instantiations exist for bufm == 2 only
#else
    if (g.prof != nullptr) return hipErrorInvalidValue;                  // stamped instantiations + ablations: experiment builds only (make experiments)
#endif
    if (g.bx3 == 3) {        // 256 x 256 tile
        dim3 grid3(((g.M + 255) / 256) * ((g.N + 255) / 256), g.ksplit > 1 ? g.ksplit : 1);
        if (g.m_split > 0) {    // two-part op(A): [gathered or plain | plain], both parts and B through LDS-DMA
            if constexpr (AMODE == OP_XC && BMODE == OP_XC) {
                const bool ok = g.m_split % 256 == 0 && g.m_split < g.M && g.A2 != nullptr && g.prof == nullptr && b_buf &&
                                g.lda % 4 == 0 && g.lda2 % 4 == 0 && (g.M - g.m_split) % 4 == 0 && g.ldb % 4 == 0 && g.N % 4 == 0 &&
                                (((uintptr_t)g.A | (uintptr_t)g.A2 | (uintptr_t)g.B) & 15) == 0 && 4LL * g.lda2 * g.K < 0xfffff000LL &&
                                (g.gather != nullptr || 4LL * g.lda * g.K < 0xfffff000LL);
                if (!ok) return hipErrorInvalidValue;
                hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 3, false, true>), grid3, dim3(512), lds_pad, s, g);
                return hipGetLastError();
            }
            return hipErrorInvalidValue;
        }
        // x-contiguous operands through LDS-DMA where their shape allows 16-byte row pieces (BxDmaXC)
        const bool dma_off = !gemm_dma_enabled();
        const bool a_dma = AMODE != OP_XC || (g.lda % 4 == 0 && g.M % 4 == 0 && ((uintptr_t)g.A & 15) == 0);
        const bool b_dma = BMODE != OP_XC || (g.ldb % 4 == 0 && g.N % 4 == 0 && ((uintptr_t)g.B & 15) == 0);
        // (measured: 4760 -> 3880 cycles per k tile with two x-contiguous operands, 16 -> 4 loads per thread and tile; with one,
        // 10 -> 4 loads, 3700 -> 3820: not used there)
        const bool dma = !dma_off && bufm == 2 && a_dma && b_dma;
        if constexpr (AMODE == OP_XC && BMODE == OP_XC) {
#ifdef FSMG_EXPERIMENTS
            if (dma && g.prof != nullptr) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, true, 3>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
#endif
            if (dma) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 3>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
        }
#ifdef FSMG_EXPERIMENTS
        if (g.prof != nullptr) { hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, true, 2>), grid3, dim3(512), lds_pad, s, g); return hipGetLastError(); }
#endif
        if (bufm == 2) hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 2>), grid3, dim3(512), lds_pad, s, g);
        else if (bufm == 1) hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE, false, 1>), grid3, dim3(512), lds_pad, s, g);
        else hipLaunchKernelGGL((k_gemm_bx3h<AMODE, BMODE>), grid3, dim3(512), lds_pad, s, g);
        return hipGetLastError();
    }
    if (g.bx3 == 2) {        // wave-specialised variant
#ifdef FSMG_EXPERIMENTS
        if (g.prof != nullptr) { hipLaunchKernelGGL((k_gemm_bx3w<AMODE, BMODE, 1, true, 2>), grid, dim3(512), lds_pad, s, g); return hipGetLastError(); }
#endif
        if (bufm == 2) hipLaunchKernelGGL((k_gemm_bx3w<AMODE, BMODE, 1, false, 2>), grid, dim3(512), lds_pad, s, g);
        else if (bufm == 1) hipLaunchKernelGGL((k_gemm_bx3w<AMODE, BMODE, 1, false, 1>), grid, dim3(512), lds_pad, s, g);
        else hipLaunchKernelGGL((k_gemm_bx3w<AMODE, BMODE, 1>), grid, dim3(512), lds_pad, s, g);
        return hipGetLastError();
    }
    if (g.bx3) {             // (gathered K rows included: BxStager)
#ifdef FSMG_EXPERIMENTS
        if (g.prof != nullptr) { hipLaunchKernelGGL((k_gemm_bx3<AMODE, BMODE, true, 2>), grid, dim3(256), lds_pad, s, g); return hipGetLastError(); }
#endif
        if (bufm == 2) hipLaunchKernelGGL((k_gemm_bx3<AMODE, BMODE, false, 2>), grid, dim3(256), lds_pad, s, g);
        else if (bufm == 1) hipLaunchKernelGGL((k_gemm_bx3<AMODE, BMODE, false, 1>), grid, dim3(256), lds_pad, s, g);
        else hipLaunchKernelGGL((k_gemm_bx3<AMODE, BMODE>), grid, dim3(256), lds_pad, s, g);
        return hipGetLastError();
    }
    // lds_pad: unused dynamic LDS that only lowers the number of co-resident blocks per CU
    // rows of a gathered XC operand change with k: that one (dKx) keeps the register-staged kernel
    if (AMODE == OP_XC && g.gather != nullptr) hipLaunchKernelGGL((k_gemm_staged<OP_XC, OP_XC>), grid, dim3(NTHREADS), lds_pad, s, g);
    else hipLaunchKernelGGL((k_gemm<AMODE, BMODE>), grid, dim3(NTHREADS), lds_pad, s, g);
    return hipGetLastError();
}

// fp32 operand -> its plane image (GemmArgs::Apl / Bpl): planes[pl][kg][x][8 bf16], kg < k8 = 2 * ceil(K / 16), zeros for k >= K.
// The same split8 the k loops run, so the pieces -- hence every product -- are the ones the in-loop split makes.
//   XC source (src[k][x]): thread = (x, kg), eight loads down k (a wave: 256 contiguous bytes each), one 16-byte store per plane
//       (a wave: 1 KiB contiguous);
//   KC source (src[x][k]): thread = (x = t / 8, kg = t % 8): 32 contiguous bytes per thread, eight threads a 256-byte run of a row; the
//       stores of a wave are eight runs of 128 bytes per plane.
template <int MODE>
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, int ld, int K, int X, uint4* __restrict__ planes) {
    const int k8 = 2 * ((K + 15) / 16);
    int x, kg;
    if (MODE == OP_XC) { x = blockIdx.x * 256 + threadIdx.x; kg = blockIdx.y; }
    else { x = blockIdx.x * 32 + (threadIdx.x >> 3); kg = blockIdx.y * 8 + (threadIdx.x & 7); }
    if (x >= X || kg >= k8) return;
    float v[8];
    const int k0 = kg * 8;
    if (MODE == OP_XC) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (k0 + i < K) ? src[(long long)(k0 + i) * ld + x] : 0.0f;
    } else {
        const float* r = src + (long long)x * ld + k0;
        if (k0 + 8 <= K && ((ld | k0) & 3) == 0 && ((uintptr_t)src & 15) == 0) {
            const float4 q0 = *reinterpret_cast<const float4*>(r), q1 = *reinterpret_cast<const float4*>(r + 4);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (k0 + i < K) ? r[i] : 0.0f;
        }
    }
    unsigned w[3][4];
    split8(v, w);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) planes[((long long)pl * k8 + kg) * X + x] = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, long long stride, int nslab,
                               float* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        float s = slabs[i];
        for (int z = 1; z < nslab; ++z) s += slabs[z * stride + i];
        out[i] = s;
    }
}

// blocks [0, nb1): range 1, the rest: range 2 -- the same per-element sum as k_reduce_slabs
__global__ void k_reduce_slabs2(const float* __restrict__ slabs, long long stride, int nslab, float* __restrict__ out, long long n,
                                int nb1, const float* __restrict__ slabs2, long long stride2, float* __restrict__ out2, long long n2) {
    const bool second = (int)blockIdx.x >= nb1;
    const float* sl = second ? slabs2 : slabs;
    float* o = second ? out2 : out;
    const long long st = second ? stride2 : stride, nn = second ? n2 : n;
    const long long b = second ? (long long)blockIdx.x - nb1 : blockIdx.x, nb = second ? (long long)gridDim.x - nb1 : nb1;
    for (long long i = b * blockDim.x + threadIdx.x; i < nn; i += nb * blockDim.x) {
        float v = sl[i];
        for (int z = 1; z < nslab; ++z) v += sl[z * st + i];
        o[i] = v;
    }
}

}  // namespace

int gemm_block_slots() { return 256 * ((BK == 16 ? 4 : 2) * 256 / NTHREADS); }
int gemm_tile_m() { return BM; }
bool gemm_dma_enabled() {
    static const bool off = std::getenv("FSMG_GEMM_DMA") && std::atoi(std::getenv("FSMG_GEMM_DMA")) == 0;
    return !off;
}
// dynamic-LDS padding that caps the resident blocks per CU (160 KiB LDS) when a GEMM runs beside the recurrent-step
// kernels on the auxiliary stream.  The padded block is just too big for blocks_per_cu + 1 of them to fit, NOT
// 1/blocks_per_cu of the LDS: the first version of this cap handed the GEMM blocks all 160 KiB, so a step-kernel
// block (4-9 KiB of LDS) could only be placed on a CU when a GEMM block retired, and a step co-running with the dW
// GEMM took 15.8 us instead of 6.4.  With this sizing the capped blocks leave >= 36 KiB per CU.
int gemm_lds_pad_for(int blocks_per_cu) {
    constexpr int own = 4 * 32 * 68 * 4;                            // static LDS of one block (epilogue > pipeline)
    const int max_blocks = (BK == 16 ? 4 : 2) * 256 / NTHREADS;
    if (blocks_per_cu >= max_blocks) return 0;
    const int block = (160 * 1024) / (blocks_per_cu + 1) + 1024;    // blocks_per_cu + 1 of these exceed 160 KiB
    return block > own ? block - own : 0;
}

hipError_t launch_gemm(hipStream_t s, int amode, int bmode, const GemmArgs& g, int lds_pad) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (amode == OP_KC && bmode == OP_XC) return launch_t<OP_KC, OP_XC>(s, g, lds_pad);
    if (amode == OP_XC && bmode == OP_XC) return launch_t<OP_XC, OP_XC>(s, g, lds_pad);
    if (amode == OP_KC && bmode == OP_KC) return launch_t<OP_KC, OP_KC>(s, g, lds_pad);
    return hipErrorInvalidValue;
}

hipError_t launch_split_planes(hipStream_t s, int mode, const float* src, int ld, int K, int X, void* planes) {
    if (K <= 0 || X <= 0) return hipSuccess;
    if (((uintptr_t)planes & 15) != 0 || (mode != OP_KC && mode != OP_XC)) return hipErrorInvalidValue;
    const int k8 = (int)plane_image_k8(K);
    if (mode == OP_XC) hipLaunchKernelGGL((k_split_planes<OP_XC>), dim3((X + 255) / 256, k8), dim3(256), 0, s, src, ld, K, X, (uint4*)planes);
    else hipLaunchKernelGGL((k_split_planes<OP_KC>), dim3((X + 31) / 32, (k8 + 7) / 8), dim3(256), 0, s, src, ld, K, X, (uint4*)planes);
    return hipGetLastError();
}

hipError_t launch_reduce_slabs2(hipStream_t s, const float* slabs, long long slab_stride, int nslab, float* out, long long n,
                                const float* slabs2, long long slab_stride2, float* out2, long long n2) {
    if (n2 <= 0) return launch_reduce_slabs(s, slabs, slab_stride, nslab, out, n);
    if (n <= 0) return launch_reduce_slabs(s, slabs2, slab_stride2, nslab, out2, n2);
    int nb1 = (int)std::min<long long>((n + 255) / 256, 2048), nb2 = (int)std::min<long long>((n2 + 255) / 256, 64);
    hipLaunchKernelGGL(k_reduce_slabs2, dim3(nb1 + nb2), dim3(256), 0, s, slabs, slab_stride, nslab, out, n, nb1, slabs2, slab_stride2, out2, n2);
    return hipGetLastError();
}

hipError_t launch_reduce_slabs(hipStream_t s, const float* slabs, long long slab_stride, int nslab,
                               float* out, long long n) {
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, s, slabs, slab_stride, nslab, out, n);
    return hipGetLastError();
}

}  // namespace fsmg
