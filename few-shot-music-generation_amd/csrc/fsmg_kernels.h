// Internal launcher interface between the C-ABI orchestration (api_*.hip behind fsmg_model.h) and the
// gfx950 kernels.  Nothing here is exported; include/fsmg.h is the public surface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsmg {

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N], fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
// Operand storage modes (what is contiguous in HBM):
//   KC: the K index is contiguous  (A stored [M][K] / B stored [N][K])
//   XC: the non-K index is contiguous (A stored [K][M] / B stored [K][N])
enum { OP_KC = 0, OP_XC = 1 };
struct GemmArgs {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int M, N, K;
    const float* bias;      // optional [N], added in the epilogue
    const int* gather;      // optional row gather for A: KC -> M-rows, XC -> K-rows index into A
    float* colsum;          // optional [N]: column sums of op(B) over K (XC B only), written by M-tile 0
    int ksplit;             // >= 1; slab z covers a K range, C/colsum slab stride below
    long long c_slab;       // elements between consecutive K-split slabs of C
    long long colsum_slab;
    // forward-only cross entropy: when ce_part != nullptr C is NOT stored; instead every wave writes, per row of its
    // 64-column slice, (max, sum exp(x - max)) over the columns < ce_nvocab into ce_part[row][2*tile_n + half] and
    // the logit of the row's target column into ce_tgt_logit[row]
    float2* ce_part; const int* ce_tgt; float* ce_tgt_logit; int ce_nvocab;
    // ce_store != 0 (train passes, "fused softmax"): besides the partials the epilogue stores E = exp(x) (x = logit incl. bias; 0 in the
    // pad columns) into C -- the un-normalised softmax with NO shift, which launch_ce_finish turns into the loss gradient by patching
    // one element per row and handing a per-row scale to the two GEMMs that consume it.  The partial of a slice is (0, sum of its E):
    // no row maximum, no target lookup (launch_ce_finish takes the target logit as log(E[target])).  Valid while every row's sum and
    // target element stay normal fp32 numbers (CE_SUM_MIN / CE_SUM_MAX / CE_TGT_MIN: launch_ce_finish checks, the caller falls back
    // to launch_ce_rows otherwise).
    int ce_store;
    // column sums of op(B) weighted per K row (XC B, 256 x 256-tile kernel only): colsum[n] = sum_k colsum_w[k] * B[k][n]
    // (dd = sum_r c_r E'[r][v] of the fused softmax); colsum_w must be readable up to K rounded up to 16
    const float* colsum_w;
    // host-side plumbing (api_schedule.hip gemm()): C[m][:] is multiplied by row_scale[m] once it is complete -- in the deferred
    // slab sum where there is one, by launch_scale_rows otherwise.  The kernels ignore it.
    const float* row_scale;
    int nt_store;           // 1: C is written with non-temporal stores (streaming, read back much later)
    // Work-queue launches for the XCD-partitioned schedule (k_gemm_queue; not with `gather` on an XC operand).  Every block
    // draws one (split, row tile, column tile) item, row tiles slowest, in the order blocks start.  work[0] / work[1]: draw
    // counters of the restricted / clean-up launch, claim[items]: taken marks; all zeroed by the caller.
    //   xcd_first > 0: restricted launch -- only blocks on XCDs >= xcd_first draw, only items < work_limit, and only while
    //                  *stop == 0 (the caller raises it when the kernel that owned the other XCDs is done);
    //   xcd_first < 0: clean-up launch -- one block per item, computes the items nobody took.  Order it after everything
    //                  the remaining items read.
    // Placement decides speed, never results: an item is computed once, by the same code either way.
    int xcd_first; int* work; int* claim; int work_limit; const int* stop;
    // Gate of a work-queue launch whose A rows are still being produced by a kernel on the other XCDs (the overlapped step: the
    // projection beside the forward recurrence).  gate[t] reaches gate_expect when the rows of time step t (gate_rows rows each,
    // time-major) are in memory; a block that has drawn a row tile waits for the tile's last step, then takes an agent-scope
    // acquire.  Bounded: after gate_spin polls, or when *gate_err == 2 (the producer gave up), the block raises *gate_err = 2 and
    // leaves its tile uncomputed -- the step is skipped like any timed-out step.  Blocks of a restricted launch that land below
    // xcd_first join the drawing once gate[gate_last] is complete (the producer has left those XCDs).
    const int* gate; int gate_expect; int gate_rows; int gate_last; int* gate_err; int gate_spin;
    int gate_every;         // the producer publishes every gate_every-th step only (LstmFwdXcdArgs::progress_every): wait for the group's last
    int bx3;                // 1: k_gemm_bx3 -- fp32 product from three bf16 planes per operand on the bf16 matrix pipe (gemm.hip);
                            // 2: k_gemm_bx3w, its wave-specialised variant (same bits; two 512-thread blocks per CU)
    int group_m;                // bf16-split kernels: tile order, see tile_coords (gemm.hip); 0 = row tiles fastest
    unsigned long long* prof;   // diagnostics (tools/gemm_bench PROF=1, bx3 only): [blocks][waves][8] stamps, see k_gemm_bx3
    int dbg;                    // diagnostics, stamped instantiation only (k_gemm_bx3w): ablation bits, results are wrong
    // Two-part op(A) of an XC x XC product on the 256 x 256-tile kernel (the merged weight gradient [x | h_prev]^T dZ of a layer: one
    // launch, one K split, one set of slabs for what were two GEMMs over the same dZ).  m_split > 0: rows [0, m_split) of op(A) are
    // the columns of A -- K rows gathered through `gather` when it is set (the caller guarantees the gathered table < 4 GiB) --,
    // rows [m_split, M) the columns of A2 (never gathered).  m_split is a multiple of 256.
    const float* A2; int lda2; int m_split;
    // Work-queue launches of the 256 x 256-tile kernel: done != nullptr -> the tile is stored write-through (sc1), and a block whose stores
    // have all been acknowledged adds 1 to done[row tile] (zeroed by the caller): what a consumer on other CUs gates the rows of that row tile on
    // (launch_ce_rows_gated: the cross entropy under the forward pair's tail)
    int* done;              // (experiment builds: measured and rejected, profiles/r05_ce_under_tail_*)
    // Operands that arrive PRE-SPLIT (256 x 256-tile kernel only; round 6): the "plane image" launch_split_planes writes -- the three
    // bf16 planes of op(A) / op(B) as [plane][k / 8][x][8 bf16], k padded with zeros to a multiple of 16, x = the M (N) index -- i.e.
    // the kernel's LDS image per k group, so a tile is 24 LDS-DMA instructions of 1 KiB and no split work in the k loop.  Same six
    // products in the same order as the in-loop split: the same bits.  A / B (fp32) are then not read (B's column sums: not with Bpl).
    const void* Apl; const void* Bpl;
};
// Plane images (GemmArgs::Apl / Bpl).  src is op(X) stored k-contiguous (mode OP_KC: src[x][k], ld >= K) or x-contiguous (OP_XC:
// src[k][x], ld >= X); planes holds plane_image_bytes(K, X) bytes.  Exact: piece1 + piece2 + piece3 == value for every finite fp32.
inline long long plane_image_k8(int K) { return 2LL * ((K + 15) / 16); }
inline long long plane_image_bytes(int K, int X) { return 3LL * plane_image_k8(K) * X * 16; }
hipError_t launch_split_planes(hipStream_t s, int mode, const float* src, int ld, int K, int X, void* planes);
// amode/bmode in {OP_KC, OP_XC}. Supported combinations: (KC,XC) (XC,XC) (KC,KC)
hipError_t launch_gemm(hipStream_t s, int amode, int bmode, const GemmArgs& g, int lds_pad = 0);
// dynamic-LDS padding that caps a GEMM at `blocks_per_cu` resident blocks per CU
int gemm_lds_pad_for(int blocks_per_cu);
// resident 128x128 blocks the chip holds at once (256 CUs x blocks per CU): the split-K policy's slot count
int gemm_block_slots();
int gemm_tile_m();   // rows of a block tile (128 or 256)
bool gemm_dma_enabled();   // the 256-tile kernel stages x-contiguous operands through LDS-DMA (FSMG_GEMM_DMA=0: registers; A/B runs)
// out[i] = sum_z slabs[z][i]  (fixed order -> deterministic split-K)
hipError_t launch_reduce_slabs(hipStream_t s, const float* slabs, long long slab_stride, int nslab,
                               float* out, long long n);
// two such sums in one launch (a split-K GEMM's C and its column sums); the second may be empty (n2 == 0)
hipError_t launch_reduce_slabs2(hipStream_t s, const float* slabs, long long slab_stride, int nslab, float* out, long long n,
                                const float* slabs2, long long slab_stride2, float* out2, long long n2);

// ---------------------------------------------------------------- recurrent steps (lstm_step.hip)
struct LstmFwdArgs {
    const float* KhF;     // fragment-ordered recurrent weights (forward copy, see k_repack_kh)
    const float* hF_prev; // fragment-ordered h_{t-1}: [ceil(B/16)][Hp/16][64 lanes][4]
    float* hF_next;       // fragment-ordered h_t (written for step t+1)
    float* z;            // [B][4Hp] in: x-part pre-activations (+bias); out: activated gates i,j,f,o
    const float* c_prev; // [B][Hp]
    float* c_next;       // [B][Hp]
    float* h_next;       // [B][Hp]
    int B, Hp;
};
// prof != nullptr selects the instrumented build: per wave 8 s_memtime stamps (entry, loads landed,
// partials in LDS, past the barrier, done)
hipError_t launch_lstm_fwd_step(hipStream_t s, const LstmFwdArgs& a, unsigned long long* prof = nullptr);

// Persistent variant of the forward steps: ONE launch runs the time steps [t0, t1) of a layer.  Block (column tile,
// row tile) keeps its slice of the recurrent weights in registers for the whole launch; the blocks of a row tile
// hand h_t to each other through the fragment-ordered copy HF, whose time indices t0+1 .. t1 must be pre-filled
// with 0xFF bytes ("not written yet", see lstm_step.hip).  All blocks must be co-resident: check
// lstm_fwd_chain_supported() first.  A spin that does not complete sets *err_flag = 2 and every block leaves.
struct LstmFwdChainArgs {
    const float* KhF;     // forward fragment-ordered recurrent weights of the layer
    float* HF;            // [T+1][ceil(B/16)*16][Hp] fragment-ordered h, time index 0 = initial state
    float* Z;             // [T][B][4Hp] in: x-part pre-activations (+bias); out: activated gates
    float* Cs;            // [T+1][B][Hp]
    float* Hs;            // [T+1][B][Hp]
    int* err_flag;
    int B, Hp, T, t0, t1;
    int spin_limit;       // polls before a wave gives up (each ~0.5 us); 0 forces the timeout path (tests)
};
bool lstm_fwd_chain_supported(int B, int Hp);
hipError_t launch_lstm_fwd_chain(hipStream_t s, const LstmFwdChainArgs& a);
// Variant for shapes whose (column tile x row tile) grid does not fit the chip (H = 1024; 100-row episodes): a block
// owns a column tile for ALL row tiles (2..8) and walks them inside a step with the same resident weights.
bool lstm_fwd_chain_rt_supported(int B, int Hp);
hipError_t launch_lstm_fwd_chain_rt(hipStream_t s, const LstmFwdChainArgs& a);

struct LstmBwdArgs {
    const float* KhF;      // fragment-ordered recurrent weights (backward copy)
    const float* dzF_next; // fragment-ordered dz of step t+1: [ceil(B/16)][4Hp/16][64][4] (nullptr at the last step)
    float* dzF_cur;        // fragment-ordered dz of step t (written for step t-1)
    float* gates;         // [B][4Hp] in: activated gates of step t; out: dz of step t
    const float* c_t;     // [B][Hp]
    const float* c_prev;  // [B][Hp]
    float* dc;            // [B][Hp] carried cell gradient (in/out)
    const float* dh_top;  // [B][Hp] gradient arriving from above at step t
    int B, Hp;
};
hipError_t launch_lstm_bwd_step(hipStream_t s, const LstmBwdArgs& a, unsigned long long* prof = nullptr);

// Persistent variant of the backward steps: ONE launch runs BPTT for the time steps t1-1 down to t0 of a layer.  Same
// hand-off protocol as LstmFwdChainArgs; dz_t is handed over through dzF_all[t], whose indices t0 .. t1-1 must be
// pre-filled with 0xFF bytes.  Step t reads dzF_all[t+1] (skipped when t+1 == T: no recurrent gradient arrives).
struct LstmBwdChainArgs {
    const float* KhF;      // backward fragment-ordered recurrent weights of the layer
    float* dzF_all;        // [T][ceil(B/16)*16][4Hp] fragment-ordered dz per time step
    float* Z;              // [T][B][4Hp] in: activated gates; out: dz (row-major, for the weight-gradient GEMMs)
    const float* Cs;       // [T+1][B][Hp]
    float* dc;             // [B][Hp] carried cell gradient: read at t1-1, written back after t0
    const float* dH;       // [T][B][Hp] gradient arriving from above
    int* err_flag;
    int B, Hp, T, t0, t1;
    int spin_limit;
};
bool lstm_bwd_chain_supported(int B, int Hp);
hipError_t launch_lstm_bwd_chain(hipStream_t s, const LstmBwdChainArgs& a);

// Reduce-scatter form of the persistent backward steps.  Block j of a row tile owns 16 hidden units = 64 packed gate
// columns and keeps dz_t for those columns to itself; what travels is dh: after step t+1 it multiplies its dz slice
// with its COLUMN slice of Kh (all Hp units x 64 columns, resident in registers) and sends every block i the 16x16
// partial of dh_t that belongs to i's units (1 KiB), so a block receives Hp/16 KiB per step instead of the
// 16 x 4Hp x 4 B of dz the all-gather form reads.  inbox: [2 slots][row tiles][consumer][producer][64 lanes][4],
// every word 0xFFFFFFFF before the first launch of a pass (readers put the pattern back after reading).
struct LstmBwdRsArgs {
    const float* KhF;      // backward fragment-ordered recurrent weights of the layer
    float* inbox;
    float* Z;              // [T][B][4Hp] in: activated gates; out: dz (row-major, for the weight-gradient GEMMs)
    const float* Cs;       // [T+1][B][Hp]
    float* dc;             // [B][Hp] carried cell gradient: read at t1-1, written back after t0
    const float* dH;       // [T][B][Hp] gradient arriving from above
    int* err_flag;
    int B, Hp, T, t0, t1;
    int spin_limit;
};
bool lstm_bwd_rs_supported(int B, int Hp);
long long lstm_bwd_rs_inbox_floats(int B, int Hp);
hipError_t launch_lstm_bwd_rs(hipStream_t s, const LstmBwdRsArgs& a);
// recurrent weights [Hp][4Hp] -> the forward and backward fragment-ordered copies (Hp*4Hp floats each)
hipError_t launch_repack_kh(hipStream_t s, const float* Kh, float* fwd, float* bwd, int Hp);

// ---------------------------------------------------------------- XCD-local recurrence (lstm_xcd.hip), hidden size 512
// Rows are split over the 8 XCDs (ceil(B/8) each); every XCD holds a full register-resident copy of K_h and hands h_t /
// the dh partials between its own 32 CUs through its L2.  grid = 256 blocks x 256 threads; roles come from the XCC id
// plus a per-XCD ticket (tickets: 8 ints, ZERO before every launch).  Same bounded-spin / err_flag = 2 contract as the
// column-split persistent kernels.
// variants of the XCD-local kernels (same results, different schedules of the cell threads' memory traffic)
enum { XCD_DEFER_OUTPUTS = 16,      // forward: c / h / gate stores of step t are issued behind the poll of step t+1; backward: dz stores behind the drain
       XCD_NO_POLL_SLEEP = 32,      // no s_sleep between two polls of a hand-off
       XCD_CHAINS = 64,             // hidden 1024: the row groups of an XCD pair as independent chains (k_lstm_*_pair_chains)
       XCD_PROBE = 128,             // k_lstm_*_pair16: a wave polls ONE word per lane (chosen so that the wave's lanes cover every producer of its
                                    // fragments) and fetches the rest only when none of those shows the fill pattern
       XCD_LOCAL_PLAIN = 256,       // k_lstm_bwd_pair16: partials / resets between CUs of the SAME XCD stay in its L2 (plain stores)
       XCD_LATE_DRAIN = 1024,       // k_lstm_bwd_pair16: the wait for the inbox resets stands in front of the first partial store, not the MFMAs
       XCD_GROUP_STORES = 2048,     // k_lstm_bwd_xcd16: the partials of four destination tiles leave while the other four multiply
       XCD_PROBE_DELAY = 8192,      // k_lstm_fwd_pair16: x (1 ... 7) = that many times 512 clocks of sleep in front of the first probe of a wave without cell threads
       XCD_STREAM = 512 };          // k_lstm_fwd_pair16 (with XCD_PROBE): behind a successful probe the fragments are ordinary (compiler-counted) loads
                                    // consumed k step by k step under the MFMAs, checked afterwards; a miss redoes the step behind the sc1 poll
int lstm_xcd_default_variant(int B, bool forward, int Hp = 512, int rpx = 0, bool bx3 = false);
struct LstmFwdXcdArgs {
    const float* KhX;     // forward register image of K_h (launch_repack_kh_xcd)
    float* HX;            // [T+1][8][4][RG][2][64][4] hand-off buffer; index 0 = zero state, t0+1 .. t1 = 0xFF fill
    float* Z;             // [T][B][4Hp] in: x-part pre-activations (+bias); out: activated gates
    float* Cs;            // [T+1][B][Hp]
    float* Hs;            // [T+1][B][Hp]
    int* tickets;
    int* err_flag;
    int B, T, t0, t1;
    int spin_limit;
    unsigned long long* prof;   // != nullptr: instrumented build, [256 blocks][4 waves][8] tick sums per phase (RG = 2 only)
    int rpx;                    // rows per XCD; 0 = ceil(B / 8).  lstm_xcd_packed_rows(B) packs the batch on the first XCDs
    int variant;                // XCD_* bits; lstm_xcd_default_variant(B, forward) has the measured choice
    int Hp;                     // 512 (0 = 512): one copy of K_h per XCD; 1024: one copy per XCD PAIR (k_lstm_fwd_pair)
    int bx3;                    // hidden 512: KhX / HX are in the bf16-split format, run k_lstm_fwd_xcd16
    int* progress;              // bf16-split kernels only, nullptr = off: [T] counters, ZERO before the launch; progress[t] reaches
                                // lstm_xcd_active_blocks(B, rpx) when the row-major h of step t (Hs index t + 1) is in memory
                                // (write-through stores) -- what a GEMM on the other XCDs gates its row tiles on
    int progress_lag;           // diagnostics: publish this many steps later than the stores' completion requires
    int progress_every;         // publish every this many steps (0 = 1): only progress[t] with t % every == every - 1, and progress[T - 1],
                                // are counted up -- a publish is a memory operation in front of the next poll of the publishing wave
};
struct LstmBwdXcdArgs {
    const float* KhXb;    // backward register image of K_h
    float* inbox;         // [2][8][32][32][RG][16][4], every word 0xFFFFFFFF before the first launch of a pass
    float* Z;             // [T][B][4Hp] in: activated gates; out: dz
    const float* Cs;      // [T+1][B][Hp]
    float* dc;            // [B][Hp]
    const float* dH;      // [T][B][Hp]
    int* tickets;
    int* err_flag;
    int B, T, t0, t1;
    int spin_limit;
    unsigned long long* prof;
    int rpx;              // as LstmFwdXcdArgs
    int variant;          // as LstmFwdXcdArgs
    int Hp;               // as LstmFwdXcdArgs
    int bx3;              // as LstmFwdXcdArgs (KhXb in the bf16-split format, k_lstm_bwd_xcd16)
};
bool lstm_xcd_supported(int B, int Hp);        // Hp 512: up to 128 rows; Hp 1024 (one copy of K_h per XCD pair): up to 64 rows
int lstm_xcd_max_rows(int Hp);                 // 128 / 64 / 0
long long lstm_xcd_hx_floats(int B, int T, int Hp = 512, bool bx3 = false, int rpx = 0);   // bx3: for the bf16-split kernels (hidden 512); rpx: rows packed per XCD
long long lstm_xcd_inbox_floats(int B, int Hp = 512, int rpx = 0);
int lstm_xcd16_packed_rows(int B, int Hp = 512);      // bf16-split kernels: rows per XCD (hidden 1024: per XCD pair) that put B rows on the fewest (up to 16 each)
inline int lstm_xcd_active_blocks(int B, int rpx, int Hp = 512) { return (Hp == 1024 ? 64 : 32) * ((B + rpx - 1) / rpx); }     // blocks of a packed launch that own rows
long long lstm_xcd_weight_floats(int Hp = 512, bool bx3 = false);   // floats per register image
bool lstm_xcd_bx3_pays(int B, int Hp);         // the bf16-split kernels are the faster ones at this row count
int lstm_xcd_packed_rows(int B);               // rows per XCD that leave whole XCDs free without adding row groups (hidden 512)
hipError_t launch_repack_kh_xcd(hipStream_t s, const float* Kh, float* fwd, float* bwd, int Hp = 512, bool bx3 = false);
// all layers, both layouts, one launch: Kh[l] -> cf / cb (launch_repack_kh) and, where xf[l] != nullptr, xf / xb (launch_repack_kh_xcd)
constexpr int REPACK_MAX_LAYERS = 4;
struct StepIncArgs;
struct RepackAllArgs { int n; int Hp; const float* Kh[REPACK_MAX_LAYERS]; float* cf[REPACK_MAX_LAYERS]; float* cb[REPACK_MAX_LAYERS]; float* xf[REPACK_MAX_LAYERS]; float* xb[REPACK_MAX_LAYERS];
                       int bx3; /* hidden 512: the XCD images as three bf16 planes (k_lstm_*_xcd16) */
                       int mode; /* 0: both layouts, 1: the XCD / XCD-pair images only, 2: the column-split fragment copies only */ };
// inc != nullptr: thread 0 of block (0, 0) also closes the train step (step_increment_body: the repack is the last kernel of a step)
hipError_t launch_repack_kh_all(hipStream_t s, const RepackAllArgs& a, const StepIncArgs* inc = nullptr);
hipError_t launch_lstm_fwd_xcd(hipStream_t s, const LstmFwdXcdArgs& a);
hipError_t launch_lstm_bwd_xcd(hipStream_t s, const LstmBwdXcdArgs& a);

// ---------------------------------------------------------------- everything else (elementwise.hip)
// up to MULTI_MAX_OPS small memory passes in ONE launch (every pointer 16-byte aligned):
//   MULTI_FILL    dst[0 .. n) = word (32-bit pattern); cond != nullptr: only when *cond != 0 (read on the device when the launch runs)
//   MULTI_REDUCE  dst[i] = sum_z src[z * stride + i] for i < n, slabs added in order z = 0 .. nslab - 1 (deterministic split-K);
//                 sq != nullptr: also the squared-norm partials of dst, one double per SQ_CHUNK (= sqnorm_blocks) elements, bit-equal
//                 to launch_sqnorm_partials(dst, n, sq)
//   MULTI_MEAN    *dst = sum(src[0 .. n)) / (n + 1e-12), one block, double accumulation in the order of launch_loss_reduce with one
//                 group (the mean loss of a train pass: rides in a launch that has work for the rest of the chip)
enum { MULTI_FILL = 0, MULTI_REDUCE = 1, MULTI_MEAN = 2 };
constexpr int MULTI_MAX_OPS = 16;
struct MultiOp { int kind; void* dst; const void* src; long long n; long long stride; int nslab; uint32_t word; const int* cond; double* sq;
                 const float* row_scale; int row_len; /* REDUCE without sq: dst[i] = row_scale[i / row_len] * sum (row_len % 4 == 0) */ };
struct MultiOps { MultiOp op[MULTI_MAX_OPS]; int count; };
hipError_t launch_multi_op(hipStream_t s, const MultiOps& r);
// out[r][0..T) = table[idx[r]][0..T) for r < n_rows (device-resident split table -> the token staging buffer)
hipError_t launch_gather_rows(hipStream_t s, const int* table, const int* idx, int n_rows, int T, int n_songs, int* out, int* err_flag);
// p[0 .. n_words) = word (p 16-byte aligned); the step uses this instead of hipMemsetAsync so that its hipGraph holds kernel nodes only
hipError_t launch_fill32(hipStream_t s, void* p, uint32_t word, long long n_words);
// dst[0 .. n_words) = src[0 .. n_words) (both 16-byte aligned, n_words a multiple of 4): a kernel instead of hipMemcpyAsync D2D, whose
// engine (shader blit or SDMA) the runtime picks per call -- 31 us for the 78 MB of cfg-E's parameters, on the stream, every time
hipError_t launch_copy_words(hipStream_t s, void* dst, const void* src, long long n_words);
// the same, skipped on the device when *cond == 0
hipError_t launch_fill32_if(hipStream_t s, const int* cond, void* p, uint32_t word, long long n_words);
// tokens [nseq][T] (support rows then query rows) -> time-major input ids X[t][b] (start word at t=0)
// and targets Y[t][b]; sets *err_flag if any id is outside [0, vocab).
// tok_first / tok_count != nullptr (train passes): also the occurrence table of the input ids for launch_embed_grad -- first
// position (atomicMin) and count per token; both arrays [vocab + 1], (INT_MAX, 0) before the launch
hipError_t launch_token_prep(hipStream_t s, const int* support, int n_support, const int* query, int n_query,
                             int T, int vocab, int start_word, int* X, int* Y, int* err_flag, int* tok_first = nullptr, int* tok_count = nullptr);
// per logits row: lse and cross entropy against the target; dlogits != nullptr also materialises
// (softmax - onehot) * inv_n (pad columns zero) for the backward projection GEMMs
hipError_t launch_ce_rows(hipStream_t s, const float* logits, int ld, int rows, int n_vocab, const int* tgt,
                          float* lse, float* ce, float* dlogits, float inv_n);
// the same rows (register-resident kernels only: ld <= 12288) by a persistent grid of `blocks` blocks, each row behind the completion
// counter of its row tile: done[row / tile_rows] >= done_expect (GemmArgs::done of the work-queue projection that is still running)
hipError_t launch_ce_rows_gated(hipStream_t s, const float* logits, int ld, int rows, int n_vocab, const int* tgt, float* lse, float* ce,
                                float* dlogits, float inv_n, const int* done, int done_expect, int tile_rows, int* err_flag, int spin_cap, int blocks,
                                int* next_row /* one int, zero before the launch: the rows are drawn from it */);
// what the shift-free softmax of GemmArgs::ce_store is used for: e^-60 <= S = sum_v exp(x_v) <= 1e30 (the largest logit of a row within
// about [-60, 69 - log(vocabulary)]) and exp(target logit) >= 1e-30 (its log is taken) -- E, S and c = 1 / (n S) stay normal fp32 numbers
constexpr float CE_SUM_MIN = 8.0e-27f, CE_SUM_MAX = 1.0e30f, CE_TGT_MIN = 1.0e-30f;
// Fused softmax of a train pass, second half (first: GemmArgs::ce_store): per row, from the partials -- lse, ce = lse - target logit,
// S = sum_v exp(x_v), c = inv_n / S -> crow[row]; E[row][tgt] -= S, so that (softmax - onehot) * inv_n == c * E' for the WHOLE row
// (dlogits is never written: dH = diag(c) (E' W^T), dW = (diag(c) Hout)^T E', dd = sum_r c_r E'[r]); hs_scaled[row][:] = c * hs[row][:].
// A row outside CE_SUM_MIN / CE_SUM_MAX / CE_TGT_MIN (or NaN) raises *err_flag = 2 and counts itself in *range_counter (host-mapped):
// the step is skipped and repeated on launch_ce_rows.
hipError_t launch_ce_finish(hipStream_t s, const float2* part, int nparts, const int* tgt, int rows, float inv_n,
                            float* E, int ld, float* lse, float* ce, float* crow, const float* hs, float* hs_scaled, int hp,
                            int* err_flag, long long* range_counter);
// C[m][0 .. N) *= row_scale[m] (rows of N floats, N % 4 == 0)
hipError_t launch_scale_rows(hipStream_t s, float* C, const float* row_scale, int M, int N);
// rows x nparts softmax partials (see GemmArgs::ce_part) -> ce[row] = logsumexp - target logit
hipError_t launch_ce_combine(hipStream_t s, const float2* part, int nparts, const float* tgt_logit, int rows, float* ce);
// out[g] = sum over t and b in group g of ce[t*B+b] / (T*rows_per_group + 1e-12); fixed order
hipError_t launch_loss_reduce(hipStream_t s, const float* ce, int T, int B, int rows_per_group, int ngroups,
                              float* out);
// dEmb[tok] = sum over occurrences r (increasing r) of dX[r]; dEmb must be zero-filled before
// tok_first / tok_count: the occurrence table launch_token_prep filled for this X (nullptr: every block scans for duplicates);
// the owners put their entries back to (INT_MAX, 0)
// part (with the table only): [n][Ep] scratch -- a token with more than 48 occurrences (padding, the most frequent words of real data)
// is summed in two levels, per 256-position chunk into part[first position of the chunk] and then chunk by chunk (k_embed_grad_chunks)
// sum (with part only): launch_sum_partials' work as one more block of the first launch instead of a launch of its own behind the second
struct SumPartialsArgs { const double* partials; int n; float* dst; const int* flag_src; const float* ce; int ce_n; float* loss_out; };
hipError_t launch_embed_grad(hipStream_t s, const int* X, int n, const float* dX, int Ep, float* dEmb, int* tok_first = nullptr, int* tok_count = nullptr,
                             float* part = nullptr, const SumPartialsArgs* sum = nullptr);
// partial sums of squares (double) of x[0..n) into partials[pofs .. pofs+nblocks); returns nblocks via out param
int sqnorm_blocks(long long n);
hipError_t launch_sqnorm_partials(hipStream_t s, const float* x, long long n, double* partials);

struct UpdateArgs {
    float* p; float* m; float* v; const float* g; long long n;   // flat buffers
    const double* partials; int n_partials;       // squared-norm partials of everything that counts
    const float* tail;                            // grad tail scalars (tail[0] = slices_sq, used when slices; tail[2] .. tail[5] != 0: no update)
    int use_slices;                               // add tail[0]*grad_scale^2 to the norm
    float grad_scale;                             // 1/world (g is a SUM over ranks)
    float lr, n_decay, clip;
    const long long* step;                        // global_step BEFORE this update (device)
    float* gnorm_out;                             // optional: pre-clip global norm
    const int* err_flag;                          // optional: *err_flag != 0 (a persistent step kernel gave up / a token id was out of range) -> no update
    // launch_adam_update in two launches over two ranges of the flat buffers (the second one on another stream, later): the first
    // writes publish[0..2] = (go ? 1 : 0, clip scale, alpha) from thread 0 of block 0, the second (consume != nullptr) reads them
    // instead of deriving them -- flags, tail and step counter may have moved on by then
    float* publish; const float* consume;
};
hipError_t launch_adam_update(hipStream_t s, const UpdateArgs& a);
// p <- p - a.lr * clip_by_global_norm(g): the inner-loop step of cfg-E (m, v, step, n_decay, grad_scale unused)
hipError_t launch_sgd_update(hipStream_t s, const UpdateArgs& a);
// last kernel of a train step: counts the step (ring[step % cap] = loss, ++step) or, when *err_flag != 0 or the
// all-reduced time-out indicator tail[2] is set, tallies it in counters ([0] time-outs, [1] token-range rejections;
// host-mapped memory) and clears the flag
// handoff_dirty (the "refill the BPTT inboxes" flag) is sticky: set when *err_flag == 2, cleared only when clear_ok != 0 (the
// step ran an XCD-local BPTT pass, which did the conditional refill) and nothing timed out
struct StepIncArgs { long long* step; const float* loss_src; float loss_scale; float* ring; int ring_cap; int* err_flag; long long* counters;
                     int* handoff_dirty; int clear_ok; };
__device__ __forceinline__ void step_increment_body(const StepIncArgs& a) {
    const int e = a.err_flag != nullptr ? *a.err_flag : 0;
    if (a.handoff_dirty != nullptr) {
        if (e == 2) *a.handoff_dirty = 1;
        else if (a.clear_ok) *a.handoff_dirty = 0;
    }
    const bool peer_timeout = a.loss_src != nullptr && a.loss_src[1] != 0.0f;      // loss_src is tail[1]; tail[2] is the indicator
    const bool peer_token = a.loss_src != nullptr && a.loss_src[2] != 0.0f;        // tail[3]: some rank's batch held an out-of-range id
    const bool peer_failed = a.loss_src != nullptr && a.loss_src[3] != 0.0f;       // tail[4]: some rank's pass failed on the host before the exchange
    const bool peer_range = a.loss_src != nullptr && a.loss_src[4] != 0.0f;        // tail[5]: some rank's logits left the fused softmax's range
    if (e != 0 || peer_timeout || peer_token || peer_failed || peer_range) {
        if (a.counters != nullptr) {
            if (e == 2 || peer_timeout) a.counters[0] += 1; else if (e == 1 || peer_token) a.counters[1] += 1;
            else if (e == 4 || peer_range) a.counters[5] += 1; else a.counters[2] += 1;
            __threadfence_system();
        }
        if (a.err_flag != nullptr) *a.err_flag = 0;
        return;
    }
    const long long s = *a.step;
    if (a.ring != nullptr && a.loss_src != nullptr) a.ring[s % a.ring_cap] = *a.loss_src * a.loss_scale;
    *a.step = s + 1;
}
hipError_t launch_step_increment(hipStream_t s, const StepIncArgs& a);
// dst[0] = (float) sum of partials[0..n) (fixed order)
// ce != nullptr: also *loss_out = sum(ce[0 .. ce_n)) / (ce_n + 1e-12) -- launch_loss_reduce with one group, same bits
hipError_t launch_sum_partials(hipStream_t s, const double* partials, int n, float* dst, const int* flag_src = nullptr,
                               const float* ce = nullptr, int ce_n = 0, float* loss_out = nullptr);
// a[0 .. n_words) against b[0 .. n_words) as 32-bit words (n_words a multiple of 4, both 16-byte aligned): every differing 16-byte
// word adds 1 to *counter (host-mapped) and sets *err_flag = 2 -- the step is then skipped like a timed-out one
hipError_t launch_compare_words(hipStream_t s, const void* a, const void* b, long long n_words, int* err_flag, long long* counter);
// Do two streams run side by side?  role 0 (launched first, on stream A): one wave polls *flag for up to `realtime_ticks` of the 100 MHz
// counter and writes out[0] = 1 if it saw it set, 0 if it gave up; role 1 (on stream B): sets *flag.  Two streams that share a hardware
// queue run B behind A, and A gives up.
hipError_t launch_queue_probe(hipStream_t s, int* flag, int* out, int role, long long realtime_ticks);
// one wave for `realtime_ticks` of the 100 MHz counter: out[0] = shader-clock ticks elapsed, out[1] = real-time ticks elapsed
hipError_t launch_clock_probe(hipStream_t s, long long realtime_ticks, unsigned long long* out);
// unigram baseline (reference src/models/unigram_model.py:26-39): counts[w] += 1 per word; out[0] = -mean(log(count[w] / sum(counts))),
// out[1] = sum(counts); *out = argmax (lowest index on ties).  *err_flag |= 1 for a word outside [0, vocab)
hipError_t launch_unigram_update(hipStream_t s, const int* words, long long n, unsigned* counts, int vocab, int* err_flag);
hipError_t launch_unigram_nll(hipStream_t s, const int* words, long long n, const unsigned* counts, int vocab, float* out, int* err_flag);
hipError_t launch_unigram_argmax(hipStream_t s, const unsigned* counts, int vocab, int* out);
// greedy decode step pieces (sample)
hipError_t launch_decode_cell(hipStream_t s, const float* Kx, int in_dim, const float* Kh, const float* bias,
                              const float* x, const float* h_in, float* h_out, float* c, int Hp);
hipError_t launch_decode_argmax(hipStream_t s, const float* W, int ldw, const float* bias, const float* h,
                                int Hp, int n_vocab, int* out_token, float* scratch);

}  // namespace fsmg
