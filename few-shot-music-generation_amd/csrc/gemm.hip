// fp32 GEMM for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak).
//
// One kernel template serves every dense contraction of the LSTM-baseline step
// (DESIGN.md "Kernels"): the hoisted input projection with the embedding gather fused
// into the A-operand load, the vocabulary projection, and their backward pairs (dlogits is
// materialised once by the cross-entropy kernel: an earlier version recomputed
// exp(logit - lse) - onehot inside the operand loads of two GEMMs, 4x per element, and was
// slower).
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 wave64 as 2x2, each wave owns a
// 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).  Both operands are staged
// through LDS K-major ([k][x]) so that the MFMA operand read is one conflict-free
// ds_read_b32 per 32-lane half (lanes 0-31 take k, lanes 32-63 take k+1):
//   * XC sources (x contiguous in HBM) are copied with 16-B loads + ds_write_b128, row
//     stride 132 floats;
//   * KC sources (k contiguous in HBM) are loaded 16 B along k and transposed on the way
//     in with 4 ds_write_b32, row stride 129 floats (129 = 1 mod 32 makes the 32 lanes of
//     a group hit 32 distinct banks).
// Global loads for tile t+1 are issued before the MFMAs of tile t (register prefetch) and
// written to the other LDS buffer afterwards: one barrier per K tile.
// Blocks are numbered so that the 8 XCDs (block b -> XCD b % 8 as observed on MI355X) each
// walk a contiguous range of tiles and share operand panels in their private L2.
#include "fsmg_kernels.h"

namespace fsmg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifndef FSMG_GEMM_BK
#define FSMG_GEMM_BK 16
#endif
// BK = 16: 34 KiB LDS and <=116 VGPRs per block -> 4 resident blocks (16 waves) per CU, which covers the
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

template <int AMODE, int BMODE>
__global__ __launch_bounds__(NTHREADS, (BK == 16 ? 1024 : 512) / NTHREADS) void k_gemm(const GemmArgs g) {
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
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
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
#ifndef FSMG_DBG_NOFETCH
        if (more) {
            la.fetch(ra, kb + (kt + 1) * BK, ke, tid);
            lb.fetch(rb, kb + (kt + 1) * BK, ke, tid);
        }
#endif
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
#ifndef FSMG_DBG_NOSTORE
        if (more) {
            la.store(As + (cur ^ 1) * ASZ, ra, tid);
            lb.store(Bs + (cur ^ 1) * BSZ, rb, tid);
        }
#endif
#ifndef FSMG_DBG_NOBARRIER
        __syncthreads();
#else
        asm volatile("" ::: "memory");
#endif
    }

    // ---- epilogue.  MFMA 32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): a lane holds
    // single floats of 16 rows.  Each wave transposes its 64x64 sub-tile through its own slice of the (now
    // idle) pipeline LDS, 32 rows at a time, and stores 16-byte row segments: 16 store instructions per lane
    // instead of 64 (the store tail of a short-K GEMM is issue bound).
    float* C = g.C + (long long)z * g.c_slab;
    constexpr int EP_LD = 68;                                  // 64 + 4: rows stay 16-byte aligned
    float* ep = smem + wave * (32 * EP_LD);                    // 4 waves x 8.5 KiB <= the 66 KiB pipeline buffers
    const int ncol0 = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * khalf) * EP_LD + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): own LDS writes landed (wave-private slice)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rl = p * 4 + (lane >> 4), c4 = (lane & 15) * 4;
            const int row = m0 + wm * 64 + i * 32 + rl, col = ncol0 + c4;
            float4 v = *reinterpret_cast<const float4*>(ep + rl * EP_LD + c4);
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
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
                float sm = 0.0f;
                if (m > NEG) sm = (expf(x0 - m) + expf(x1 - m)) + (expf(x2 - m) + expf(x3 - m));
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
                if (row < g.M) {
                    const int t = g.ce_tgt[row];
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
    if (do_colsum && n0 + tid < g.N) g.colsum[(long long)z * g.colsum_slab + n0 + tid] = csum;
}

template <int AMODE, int BMODE>
hipError_t launch_t(hipStream_t s, const GemmArgs& g, int lds_pad) {
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, g.ksplit > 1 ? g.ksplit : 1);
    // lds_pad: unused dynamic LDS that only lowers the number of co-resident blocks per CU
    hipLaunchKernelGGL((k_gemm<AMODE, BMODE>), grid, dim3(NTHREADS), lds_pad, s, g);
    return hipGetLastError();
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

}  // namespace

int gemm_block_slots() { return 256 * ((BK == 16 ? 4 : 2) * 256 / NTHREADS); }
int gemm_tile_m() { return BM; }
// dynamic-LDS padding that caps the resident blocks per CU (160 KiB LDS): leaves room for the recurrent-step
// kernels' waves and registers when a GEMM runs beside them on the auxiliary stream
int gemm_lds_pad_for(int blocks_per_cu) {
    constexpr int own = (BK == 16) ? NWAVES * 32 * 68 * 4 : 67584;  // static LDS of one block
    const int max_blocks = (BK == 16 ? 4 : 2) * 256 / NTHREADS;
    if (blocks_per_cu >= max_blocks) return 0;
    const int budget = (160 * 1024) / blocks_per_cu;                // LDS share that admits exactly this many
    return budget - own - 1024 > 0 ? budget - own - 1024 : 0;
}

hipError_t launch_gemm(hipStream_t s, int amode, int bmode, const GemmArgs& g, int lds_pad) {
    if (g.M <= 0 || g.N <= 0) return hipSuccess;
    if (amode == OP_KC && bmode == OP_XC) return launch_t<OP_KC, OP_XC>(s, g, lds_pad);
    if (amode == OP_XC && bmode == OP_XC) return launch_t<OP_XC, OP_XC>(s, g, lds_pad);
    if (amode == OP_KC && bmode == OP_KC) return launch_t<OP_KC, OP_KC>(s, g, lds_pad);
    return hipErrorInvalidValue;
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
