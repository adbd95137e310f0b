// HBM-bound and small kernels of the step: token staging, softmax cross-entropy rows,
// deterministic reductions, embedding-gradient segmented sum, global-norm + TF-style Adam,
// parameter init and the greedy-decode GEMVs.  wave64 everywhere; reductions use
// __shfl_xor over 64 lanes.
#include <algorithm>
#include "fsmg_kernels.h"
#include <cstdlib>

namespace fsmg {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- fill
// 32-bit pattern fill with 16-byte stores.  Used instead of hipMemsetAsync inside the step: a step is captured into a
// hipGraph, and on ROCm 7.2 a replayed graph with this step's memset nodes left 16-byte garbage patterns in the zero
// state block of the hidden states from the second launch on (tools/race_hunt2.py) -- a kernel node has no such surprises.
__global__ __launch_bounds__(256) void k_fill32(uint32_t* __restrict__ p, uint32_t word, long long n) {
    const long long n4 = n >> 2;
    const uint4 w4 = make_uint4(word, word, word, word);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        reinterpret_cast<uint4*>(p)[i] = w4;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = word;
}

// the same fill, done only when *cond != 0 (every thread reads the word; whoever changes it does so in a LATER kernel)
__global__ __launch_bounds__(256) void k_fill32_if(const int* __restrict__ cond, uint32_t* __restrict__ p, uint32_t word, long long n) {
    if (*cond == 0) return;
    const long long n4 = n >> 2;
    const uint4 w4 = make_uint4(word, word, word, word);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        reinterpret_cast<uint4*>(p)[i] = w4;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = word;
}

// several small memory passes in one launch (blockIdx.y = op): a step needs ~10 fills and up to five split-K slab sums, each a
// ~5 us dispatch on its own.  Op kinds: FILL -- 32-bit pattern, optionally only when *cond != 0; REDUCE -- out[i] = sum_z
// slabs[z][i] in slab order (deterministic split-K), optionally with the squared-norm partials of `out` as a by-product:
// partial c covers elements [c * SQ_CHUNK, (c + 1) * SQ_CHUNK) with the thread -> element map and the summation order of
// k_sqnorm_partials, so the fused pass and the separate one give the same bits.
constexpr int SQ_CHUNK = 256 * 16;   // elements per squared-norm partial
__global__ __launch_bounds__(256) void k_multi_op(const MultiOps r) {
    const MultiOp& o = r.op[blockIdx.y];
    if (o.kind == MULTI_FILL) {
        if (o.cond != nullptr && *o.cond == 0) return;
        uint32_t* p = (uint32_t*)o.dst;
        const uint32_t word = o.word;
        const long long n = o.n, n4 = n >> 2;
        const uint4 w4 = make_uint4(word, word, word, word);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
            reinterpret_cast<uint4*>(p)[i] = w4;
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = word;
        return;
    }
    __shared__ double sh[4];
    if (o.kind == MULTI_MEAN) {
        if (blockIdx.x != 0) return;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const float* x = (const float*)o.src;
        const int n = (int)o.n;
        double l = 0.0;
        for (int i0 = tid; i0 < n; i0 += 256 * 16) {          // sixteen loads in flight, added in index order (the order of k_loss_reduce)
            float b[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = (i0 + 256 * k < n) ? x[i0 + 256 * k] : 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) if (i0 + 256 * k < n) l += (double)b[k];
        }
        l = wave_sum_d(l);
        if (lane == 0) sh[wave] = l;
        __syncthreads();
        if (tid == 0) *(float*)o.dst = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / ((double)n + 1e-12));
        return;
    }
    // REDUCE: chunks of SQ_CHUNK elements, grid-stride over the chunks.  A thread owns four float4 of a chunk; the four loads of a
    // slab are issued together (and two slabs per round), so a thread has eight 16-byte loads in flight instead of one -- the first
    // version walked the slabs of one float4 at a time and ran at a third of the HBM rate (26 us for 70 MB).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* sl = (const float*)o.src;
    float* out = (float*)o.dst;
    const long long n = o.n, nchunks = (n + SQ_CHUNK - 1) / SQ_CHUNK;
    const bool vec = (o.stride & 3) == 0 && (n & 3) == 0;     // 16-byte aligned slabs and rows (every slab base is: allocations are 256-byte aligned)
    if (o.sq == nullptr && vec) {
        // few chunks, many slabs (the dH sum at the reference's default dims: 141 chunks x 16 slabs ran at 1.2 TB/s): without squared-
        // norm partials to keep in their fixed chunking, one float4 per thread and four slabs in flight, grid-stride over float4s
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * 256 + tid; i < n4; i += (long long)gridDim.x * 256) {
            float4 v = *reinterpret_cast<const float4*>(sl + 4 * i);
            int z = 1;
            for (; z + 3 < o.nslab; z += 4) {
                const float4 w0 = *reinterpret_cast<const float4*>(sl + z * o.stride + 4 * i);
                const float4 w1 = *reinterpret_cast<const float4*>(sl + (z + 1) * o.stride + 4 * i);
                const float4 w2 = *reinterpret_cast<const float4*>(sl + (z + 2) * o.stride + 4 * i);
                const float4 w3 = *reinterpret_cast<const float4*>(sl + (z + 3) * o.stride + 4 * i);
                v.x += w0.x; v.y += w0.y; v.z += w0.z; v.w += w0.w;
                v.x += w1.x; v.y += w1.y; v.z += w1.z; v.w += w1.w;
                v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
                v.x += w3.x; v.y += w3.y; v.z += w3.z; v.w += w3.w;
            }
            for (; z < o.nslab; ++z) {
                const float4 w = *reinterpret_cast<const float4*>(sl + z * o.stride + 4 * i);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (o.row_scale != nullptr) {               // (dH of the fused softmax: row r of the sum times c_r)
                const float c = o.row_scale[(4 * i) / o.row_len];
                v.x *= c; v.y *= c; v.z *= c; v.w *= c;
            }
            *reinterpret_cast<float4*>(out + 4 * i) = v;
        }
        return;
    }
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long base = c * SQ_CHUNK;
        double s = 0.0;
        if (vec && base + SQ_CHUNK <= n) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(sl + base + 4 * (tid + 256 * i));
            int z = 1;
            for (; z + 1 < o.nslab; z += 2) {
                float4 w0[4], w1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w0[i] = *reinterpret_cast<const float4*>(sl + z * o.stride + base + 4 * (tid + 256 * i));
                    w1[i] = *reinterpret_cast<const float4*>(sl + (z + 1) * o.stride + base + 4 * (tid + 256 * i));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i].x += w0[i].x; v[i].y += w0[i].y; v[i].z += w0[i].z; v[i].w += w0[i].w;
                    v[i].x += w1[i].x; v[i].y += w1[i].y; v[i].z += w1[i].z; v[i].w += w1[i].w;
                }
            }
            if (z < o.nslab) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 w = *reinterpret_cast<const float4*>(sl + z * o.stride + base + 4 * (tid + 256 * i));
                    v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(out + base + 4 * (tid + 256 * i)) = v[i];
                s += ((double)v[i].x * v[i].x + (double)v[i].y * v[i].y) + ((double)v[i].z * v[i].z + (double)v[i].w * v[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long idx = base + 4 * (tid + 256 * i);
                float q[4] = {0.f, 0.f, 0.f, 0.f};
                int k = 0;
                for (long long j = idx; j < n && j < idx + 4; ++j, ++k) {
                    float v = sl[j];
                    for (int z = 1; z < o.nslab; ++z) v += sl[z * o.stride + j];
                    out[j] = v;
                    q[k] = v;
                }
                if (k == 4) s += ((double)q[0] * q[0] + (double)q[1] * q[1]) + ((double)q[2] * q[2] + (double)q[3] * q[3]);
                else for (int e = 0; e < k; ++e) s += (double)q[e] * q[e];
            }
        }
        if (o.sq != nullptr) {
            s = wave_sum_d(s);
            if (lane == 0) sh[wave] = s;
            __syncthreads();
            if (tid == 0) o.sq[c] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- token prep (K0)
// Reference: convert_tokens_to_input_and_target + concat (src/models/base_model.py:63-86,
// src/models/lstm_baseline.py:91-96).  One thread per (sequence b, step t): coalesced reads
// of the [nseq][T] token rows, time-major writes.
__global__ void k_token_prep(const int* __restrict__ support, int n_support, const int* __restrict__ query,
                             int n_query, int T, int vocab, int start_word, int* __restrict__ X,
                             int* __restrict__ Y, int* __restrict__ err_flag, int* __restrict__ tok_first, int* __restrict__ tok_count) {
    __shared__ int s_key[512], s_cnt[512], s_min[512];
    const int B = n_support + n_query;
    const long long total = (long long)B * T;
    // (block-uniform trip count: the occurrence-table code below synchronises the block)
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < total; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = i0 + threadIdx.x;
        const bool valid = i < total;
        const int b = valid ? (int)(i / T) : 0, t = valid ? (int)(i % T) : 0;
        int tok = 0;
        if (valid) {
            const int* row = (b < n_support) ? support + (long long)b * T : query + (long long)(b - n_support) * T;
            tok = row[t];
            if (tok < 0 || tok >= vocab) {
                atomicOr(err_flag, 1);
                tok = min(max(tok, 0), vocab - 1);
            }
            Y[(long long)t * B + b] = tok;
            if (t + 1 < T) X[(long long)(t + 1) * B + b] = tok;
            if (t == 0) X[b] = start_word;
        }
        // occurrence table of the input ids (train passes): first position and count per token -- integer atomics, the result
        // does not depend on their order.  k_embed_grad's owner blocks read it and put it back to (INT_MAX, 0).
        if (tok_first != nullptr) {
            // Real data repeats: the zero padding behind a song's end and the frequent words -- 2 000 atomics on ONE word of the table cost
            // 27 us per pass on padded Zipf episodes.  The block's 256 positions meet in an LDS hash table first (512 slots, linear probing,
            // LDS atomics), and every occupied slot sends one pair of global atomics: integer min / add, the table does not depend on the
            // grouping.  (A position that finds no slot in eight probes goes to the global table itself.)
            for (int j = threadIdx.x; j < 512; j += blockDim.x) { s_key[j] = -1; s_cnt[j] = 0; s_min[j] = 0x7FFFFFFF; }
            __syncthreads();
            if (valid && t + 1 < T) {
                const int pos = (int)((long long)(t + 1) * B + b);
                unsigned slot = ((unsigned)tok * 2654435761u) >> 23;
                bool placed = false;
#pragma unroll 1
                for (int probe = 0; probe < 8 && !placed; ++probe) {
                    const int prev = atomicCAS(&s_key[slot], -1, tok);
                    if (prev == -1 || prev == tok) { atomicAdd(&s_cnt[slot], 1); atomicMin(&s_min[slot], pos); placed = true; }
                    else slot = (slot + 1) & 511u;
                }
                if (!placed) { atomicMin(tok_first + tok, pos); atomicAdd(tok_count + tok, 1); }
            }
            __syncthreads();
            for (int j = threadIdx.x; j < 512; j += blockDim.x)
                if (s_key[j] >= 0) { atomicMin(tok_first + s_key[j], s_min[j]); atomicAdd(tok_count + s_key[j], s_cnt[j]); }
            __syncthreads();
            if (valid && t == 0) { atomicMin(tok_first + start_word, b); atomicAdd(tok_count + start_word, 1); }
        }
    }
}

// Episode gather from the device-resident split table (reference src/data/episode.py:62-74 fills the same rows from its
// per-song cache): out[r][t] = table[idx[r]][t]; an index outside [0, n_songs) raises the token-range flag.
__global__ void k_gather_rows(const int* __restrict__ table, const int* __restrict__ idx, int n_rows, int T, int n_songs,
                              int* __restrict__ out, int* __restrict__ err_flag) {
    const long long total = (long long)n_rows * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / T), t = (int)(i % T);
        int song = idx[r];
        if (song < 0 || song >= n_songs) { atomicOr(err_flag, 1); song = 0; }
        out[i] = table[(long long)song * T + t];
    }
}

// ---------------------------------------------------------------- softmax cross entropy per row (K6)
// One 256-thread block per logits row; pass 1 row max, pass 2 sum exp (second read is an L2 hit).
// With dlogits != nullptr a third pass (the row is cache-hot) also writes the loss gradient
// dlogits = (softmax - onehot) * inv_n for the two backward projection GEMMs, zero in the pad columns.
// dlogits may BE logits (the train pass writes the gradient over the logits it has just read, fsmg_model::inplace_dlogits): neither
// pointer is `restrict`, and a thread writes only elements it has itself read before.
__global__ __launch_bounds__(256) void k_ce_rows(const float* logits, int ld, int n_vocab,
                                                 const int* __restrict__ tgt, float* __restrict__ lse,
                                                 float* __restrict__ ce, float* dlogits, float inv_n) {
    __shared__ float sh[4];
    __shared__ float s_lse;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (long long)r * ld;
    const int nv4 = n_vocab & ~3;
    float m = -INFINITY;
    for (int v = 4 * tid; v < nv4; v += 1024) {
        const float4 x = *reinterpret_cast<const float4*>(row + v);
        m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
    }
    if (tid < n_vocab - nv4) m = fmaxf(m, row[nv4 + tid]);
    m = wave_max(m);
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float s = 0.0f;
    for (int v = 4 * tid; v < nv4; v += 1024) {
        const float4 x = *reinterpret_cast<const float4*>(row + v);
        s += (expf(x.x - m) + expf(x.y - m)) + (expf(x.z - m) + expf(x.w - m));
    }
    if (tid < n_vocab - nv4) s += expf(row[nv4 + tid] - m);
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    const int t = tgt[r];
    if (tid == 0) {
        const float l = m + logf((sh[0] + sh[1]) + (sh[2] + sh[3]));
        lse[r] = l;
        ce[r] = l - row[t];
        s_lse = l;
    }
    if (dlogits == nullptr) return;
    __syncthreads();                                     // (also: row[t] has been read before anybody overwrites it in place)
    const float l = s_lse;
    float* drow = dlogits + (long long)r * ld;
    for (int v = 4 * tid; v < ld; v += 1024) {          // ld is a multiple of 4 and >= n_vocab
        const float4 x = *reinterpret_cast<const float4*>(row + v);
        float4 d;
        d.x = (v + 0 < n_vocab) ? (expf(x.x - l) - (v + 0 == t ? 1.0f : 0.0f)) * inv_n : 0.0f;
        d.y = (v + 1 < n_vocab) ? (expf(x.y - l) - (v + 1 == t ? 1.0f : 0.0f)) * inv_n : 0.0f;
        d.z = (v + 2 < n_vocab) ? (expf(x.z - l) - (v + 2 == t ? 1.0f : 0.0f)) * inv_n : 0.0f;
        d.w = (v + 3 < n_vocab) ? (expf(x.w - l) - (v + 3 == t ? 1.0f : 0.0f)) * inv_n : 0.0f;
        __builtin_nontemporal_store(d.x, drow + v); __builtin_nontemporal_store(d.y, drow + v + 1);
        __builtin_nontemporal_store(d.z, drow + v + 2); __builtin_nontemporal_store(d.w, drow + v + 3);
    }
}

// Register-resident variant for rows of up to NV * 1024 floats (cfg-B: 10 float4 per thread): the row is read ONCE,
// exp() is evaluated once per element, and the gradient is written from registers -- the three-pass kernel above
// re-reads 40 KB rows that have long left the L2 when thousands of rows are in flight (0.141 -> see DESIGN.md).
// softmax = exp(x - max) / sum instead of exp(x - lse): the same value to ~1 ulp.
// one row of the register-resident cross entropy, by the 256 threads of a block (sh: 8 floats of LDS; two block barriers inside)
template <int NV, bool NT>
__device__ __forceinline__ void ce_row_reg(float* sh, const int r, const float* logits, int ld, int n_vocab, const int* __restrict__ tgt,
                                           float* __restrict__ lse, float* __restrict__ ce, float* dlogits, float inv_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (long long)r * ld;
    float4 x[NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = 4 * tid + 1024 * i;
        x[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (v < ld) {                                   // ld is a multiple of 4 and >= n_vocab: whole float4 in bounds
            const float4 q = *reinterpret_cast<const float4*>(row + v);
            x[i].x = (v + 0 < n_vocab) ? q.x : -INFINITY; x[i].y = (v + 1 < n_vocab) ? q.y : -INFINITY;
            x[i].z = (v + 2 < n_vocab) ? q.z : -INFINITY; x[i].w = (v + 3 < n_vocab) ? q.w : -INFINITY;
        }
        m = fmaxf(fmaxf(m, fmaxf(x[i].x, x[i].y)), fmaxf(x[i].z, x[i].w));
    }
    m = wave_max(m);
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    const int t = tgt[r];
    float s = 0.0f, xt = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = 4 * tid + 1024 * i;
        if (t >= v && t < v + 4) xt = (t == v) ? x[i].x : (t == v + 1) ? x[i].y : (t == v + 2) ? x[i].z : x[i].w;
        x[i].x = expf(x[i].x - m); x[i].y = expf(x[i].y - m); x[i].z = expf(x[i].z - m); x[i].w = expf(x[i].w - m);
        s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
    }
    s = wave_sum(s);
    xt = wave_sum(xt);                                   // exactly one lane of the block holds the target logit
    if (lane == 0) { sh[4 + wave] = s; }
    __syncthreads();
    s = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    const int tw = ((t >> 2) & 255) >> 6;                // the wave whose lanes cover column t
    if (wave == tw && lane == 0) {
        const float l = m + logf(s);
        lse[r] = l;
        ce[r] = l - xt;
    }
    if (dlogits == nullptr) return;
    const float scale = inv_n / s;
    float* drow = dlogits + (long long)r * ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = 4 * tid + 1024 * i;
        if (v < ld) {
            float4 d;
            d.x = x[i].x * scale - (v + 0 == t ? inv_n : 0.0f);
            d.y = x[i].y * scale - (v + 1 == t ? inv_n : 0.0f);
            d.z = x[i].z * scale - (v + 2 == t ? inv_n : 0.0f);
            d.w = x[i].w * scale - (v + 3 == t ? inv_n : 0.0f);
            if (NT) {
                __builtin_nontemporal_store(d.x, drow + v); __builtin_nontemporal_store(d.y, drow + v + 1);
                __builtin_nontemporal_store(d.z, drow + v + 2); __builtin_nontemporal_store(d.w, drow + v + 3);
            } else {
                *reinterpret_cast<float4*>(drow + v) = d;
            }
        }
    }
}

template <int NV, bool NT>
__global__ __launch_bounds__(256) void k_ce_rows_reg(const float* logits, int ld, int n_vocab,
                                                     const int* __restrict__ tgt, float* __restrict__ lse,
                                                     float* __restrict__ ce, float* dlogits, float inv_n) {     // dlogits may be logits: the whole row is in registers behind the barriers
    __shared__ float sh[8];
    ce_row_reg<NV, NT>(sh, blockIdx.x, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
}

#ifdef FSMG_EXPERIMENTS         // measured and rejected (fsmg_model.h: ce_tail; profiles/r05_ce_under_tail_*): experiment builds only
// The same rows by a PERSISTENT grid (every block draws the next row from *next_row, zeroed by the caller: rows are taken in increasing
// order by whichever blocks have found a CU -- blocks the dispatcher cannot place yet cost nothing) whose rows are still being written
// by the work-queue projection on other CUs (the XCD-partitioned order: the cross entropy under the forward pair's tail).
// done[row / tile_rows] reaches done_expect when every column tile of that row tile has been stored AND released at agent scope
// (k_gemm_bx3h<.., QUEUE>: GemmArgs::done); one thread polls, the block barrier orders the other waves' loads behind it.  A row tile
// is 256 rows = a multiple of 1 KiB of any leading dimension, so no cache line is shared between a complete tile and one still
// being written; the lines are read with ordinary loads -- this kernel's launch invalidated the L2s, and a line is touched only
// behind its tile's counter.  Bounded like every spin of the library: after spin_cap polls (or when somebody else has raised the
// flag) *err_flag = 2 and the block leaves -- the step is skipped and repeated like any timed-out step.
template <int NV, bool NT>
__global__ __launch_bounds__(256) void k_ce_rows_gated(const float* logits, int ld, int rows, int n_vocab, const int* __restrict__ tgt,
                                                       float* __restrict__ lse, float* __restrict__ ce, float* dlogits, float inv_n,
                                                       const int* done, int done_expect, int tile_rows, int* err_flag, int spin_cap, int* next_row) {
    __shared__ float sh[8];
    __shared__ int s_row;
    int have = -1;                                       // the row tile this block has already seen complete
    for (;;) {
        if (threadIdx.x == 0) {
            int r = atomicAdd(next_row, 1);
            if (r < rows && r / tile_rows != have) {
                for (int spins = 0; __hip_atomic_load(done + r / tile_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < done_expect; ++spins) {
                    __builtin_amdgcn_s_sleep(32);
                    if (spins >= spin_cap || ((spins & 63) == 63 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2)) {
                        __hip_atomic_store(err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        r = rows;                        // give up: the step is skipped
                        break;
                    }
                }
            }
            s_row = r;
        }
        __syncthreads();
        const int r = s_row;
        __syncthreads();
        if (r >= rows) return;
        have = r / tile_rows;
        ce_row_reg<NV, NT>(sh, r, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
        __syncthreads();                                 // sh is reused by the next row
    }
}
#endif

// One wave per row: combine the per-slice (max, sum exp) partials written by the projection GEMM's epilogue.
__global__ __launch_bounds__(256) void k_ce_combine(const float2* __restrict__ part, int nparts,
                                                    const float* __restrict__ tgt_logit, int rows, float* __restrict__ ce) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float2* p = part + (long long)row * nparts;
    float m = -INFINITY;
    for (int i = lane; i < nparts; i += 64) m = fmaxf(m, p[i].x);
    m = wave_max(m);
    float s = 0.0f;
    for (int i = lane; i < nparts; i += 64) {
        const float2 v = p[i];
        if (v.x > -INFINITY) s += v.y * expf(v.x - m);
    }
    s = wave_sum(s);
    if (lane == 0) ce[row] = m + logf(s) - tgt_logit[row];
}

// Second half of the fused softmax of a train pass (launch_ce_finish, fsmg_kernels.h).  One wave per row.
__global__ __launch_bounds__(256) void k_ce_finish(const float2* __restrict__ part, int nparts,
                                                   const int* __restrict__ tgt, int rows, float inv_n, float* E, int ld,
                                                   float* __restrict__ lse, float* __restrict__ ce, float* __restrict__ crow,
                                                   const float* __restrict__ hs, float* __restrict__ hs_scaled, int hp,
                                                   int* err_flag, unsigned long long* range_counter) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    // the weighted column sums of dW read crow[K .. K + 15] past this pass's rows (times zero-filled B rows): never a stale Inf / NaN there
    if (blockIdx.x == 0 && threadIdx.x < 32) crow[rows + threadIdx.x] = 0.0f;
    if (row >= rows) return;
    const float2* p = part + (long long)row * nparts;
    float S = 0.0f;                               // = sum_v exp(x_v): the partials are un-shifted sums of the stored E values
    for (int i = lane; i < nparts; i += 64) S += p[i].y;
    S = wave_sum(S);
    const float c = inv_n / S;
    if (lane == 0) {
        float* e = E + (long long)row * ld + tgt[row];
        const float et = *e;
        const float l = logf(S);
        lse[row] = l;
        ce[row] = l - logf(et);                   // the target logit back from its E (exp and log agree to ~1 ulp of the logit's scale)
        crow[row] = c;
        // outside (or NaN / Inf): E, S or c left the normal fp32 range, or the target's E is too small to take its log
        if (!(S >= CE_SUM_MIN && S <= CE_SUM_MAX && et >= CE_TGT_MIN)) {           // this step takes the shifted softmax
            // 4 = "a row left the fused softmax's range": its own flag value -- nothing timed out, no hand-off is left half-done, and
            // the recurrent kernels' schedule is not at fault; a time-out or token-range flag already raised stays
            atomicCAS(err_flag, 0, 4);
            crow[row] = 0.0f;                         // (never an Inf / NaN weight in a later, smaller pass's column sums: NaN * 0 = NaN)
            atomicAdd_system(range_counter, 1ull);
            __threadfence_system();
        } else {
            *e = et - S;                              // (softmax - onehot) * inv_n == c * E' now holds for the whole row
        }
    }
    const float* hrow = hs + (long long)row * hp;
    float* orow = hs_scaled + (long long)row * hp;
    for (int i = 4 * lane; i < hp; i += 256) {
        float4 v = *reinterpret_cast<const float4*>(hrow + i);
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        *reinterpret_cast<float4*>(orow + i) = v;
    }
}

__global__ __launch_bounds__(256) void k_scale_rows(float* C, const float* __restrict__ row_scale, long long n4, int n_per_row4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float c = row_scale[i / n_per_row4];
        float4 v = reinterpret_cast<float4*>(C)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        reinterpret_cast<float4*>(C)[i] = v;
    }
}

// out[g] = sum_{t, b in group g} ce[t*B+b] / (T*rpg + 1e-12), double accumulation, fixed order.
__global__ __launch_bounds__(256) void k_loss_reduce(const float* __restrict__ ce, int T, int B, int rpg,
                                                     float* __restrict__ out) {
    __shared__ double sh[4];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = T * rpg;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) {
        const int t = i / rpg, b = g * rpg + i % rpg;
        s += (double)ce[(long long)t * B + b];
    }
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) out[g] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / ((double)n + 1e-12));
}

// ---------------------------------------------------------------- embedding gradient (K7 tail)
// dEmb[tok] = sum of dX rows of every occurrence of tok, accumulated in increasing position
// order by the block of the FIRST occurrence (owner computes: no atomics, deterministic).
// tok_first / tok_count (k_token_prep): the block of position r owns X[r] iff tok_first[X[r]] == r, and it stops scanning
// once it has seen tok_count[X[r]] occurrences -- at cfg-B three positions in four hold a token that occurs once, whose
// block copies one row; without the table every block scanned all earlier positions for a duplicate and all later ones
// for more occurrences (25 us for 5.9 MB).  The owner resets its table entry, so a completed pass leaves the table clean.
// (k_sum_partials below; also the last block of k_embed_grad_chunks, which would otherwise be a launch with nothing to do on uniform data)
__device__ __forceinline__ void sum_partials_body(const double* __restrict__ partials, int n, float* dst,
                                                  const int* flag_src, const float* __restrict__ ce, int ce_n, float* loss_out) {
    __shared__ double sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += partials[i];
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) {
        dst[0] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
        if (flag_src != nullptr) {
            dst[2] = (*flag_src == 2) ? 1.0f : 0.0f; dst[3] = (*flag_src == 1) ? 1.0f : 0.0f; dst[4] = 0.0f;
            dst[5] = (*flag_src == 4) ? 1.0f : 0.0f;        // fused-softmax range: travels in the reduced tail, every rank switches together
        }
    }
    if (ce == nullptr) return;
    // the mean loss of a train pass (k_loss_reduce with one group: same order, same bits) -- nobody reads it before the
    // step's last kernels, so it rides here instead of costing a launch behind the cross entropy
    __syncthreads();
    double l = 0.0;
    for (int i0 = tid; i0 < ce_n; i0 += 256 * 16) {        // sixteen loads in flight, added in index order (the order of k_loss_reduce)
        float b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) b[k] = (i0 + 256 * k < ce_n) ? ce[i0 + 256 * k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (i0 + 256 * k < ce_n) l += (double)b[k];
    }
    l = wave_sum_d(l);
    if (lane == 0) sh[wave] = l;
    __syncthreads();
    if (tid == 0) *loss_out = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / ((double)ce_n + 1e-12));
}


// Heavy tokens.  Real data is not uniform: the zero padding behind a song's end and the most frequent words occur thousands of times
// per pass (padded Zipf episodes at cfg-B: 2 000+ of 5 760 positions hold token 0), and an owner block that adds 2 000 rows one
// after the other took 736 us of a 1.6 ms step.  A token with more than EMBED_HEAVY occurrences is therefore summed in two levels:
// k_embed_grad_chunks -- the block of the token's first occurrence INSIDE each 256-position chunk adds the chunk's occurrences in
// position order into part[that position] -- and the owner in k_embed_grad adds those partial rows in chunk order.  Fixed order
// either way; which path a token takes depends on its count only.
constexpr int EMBED_HEAVY = 48;
__global__ __launch_bounds__(256) void k_embed_grad_chunks(const int* __restrict__ X, int n, const float* __restrict__ dX, int Ep,
                                                           float* __restrict__ part, const int* __restrict__ tok_count, const SumPartialsArgs sp) {
    // one block per 256-position chunk (uniform data: 23 blocks that find nothing to do); the heavy tokens of the chunk one after the other.
    // One more block does what k_sum_partials would do in a launch of its own right behind (sp.dst != nullptr): it depends on nothing here.
    if (blockIdx.x == (unsigned)((n + 255) / 256)) { sum_partials_body(sp.partials, sp.n, sp.dst, sp.flag_src, sp.ce, sp.ce_n, sp.loss_out); return; }
    __shared__ unsigned long long mask[4];
    __shared__ __attribute__((aligned(16))) float wsum[4][1024];
    const int base = blockIdx.x * 256, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = base + tid;
    const int tok = (i < n) ? X[i] : -1;
    bool pending = (i < n) && tok_count[tok] > EMBED_HEAVY;
    for (;;) {
        const unsigned long long pb = __ballot(pending);
        if (lane == 0) mask[wave] = pb;
        __syncthreads();
        int first = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (first < 0 && mask[w]) first = 64 * w + __ffsll((long long)mask[w]) - 1;
        __syncthreads();
        if (first < 0) break;                                   // (block-uniform: every thread read the same four words)
        const int tk = X[base + first];
        const bool hit = pending && tok == tk;                   // every occurrence of tk in the chunk (all of them are still pending)
        const unsigned long long hb = __ballot(hit);
        if (lane == 0) mask[wave] = hb;
        __syncthreads();
        // every wave adds the rows of ITS 64 positions (lane = a float4 of the row; four rows' loads in flight), the four wave sums are added
        // in wave order: a fixed order again, and the padding's ~100 rows of a chunk cost 25 dependent adds instead of 100
        {
            float4 a4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) a4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned long long mk = mask[wave];
            const int nf4 = Ep >> 2;                       // float4 per row (Ep % 4 == 0, <= 1024)
            while (mk) {
                int bits[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { bits[k] = mk ? __ffsll((long long)mk) - 1 : -1; mk &= mk - 1; }
                float4 v[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4* src = reinterpret_cast<const float4*>(dX + (long long)(base + 64 * wave + max(bits[k], 0)) * Ep);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[k][c] = (bits[k] >= 0 && lane + 64 * c < nf4) ? src[lane + 64 * c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bits[k] >= 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { a4[c].x += v[k][c].x; a4[c].y += v[k][c].y; a4[c].z += v[k][c].z; a4[c].w += v[k][c].w; }
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (lane + 64 * c < nf4) reinterpret_cast<float4*>(&wsum[wave][0])[lane + 64 * c] = a4[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (tid + 256 * c < Ep) {
                const int col = tid + 256 * c;
                part[(long long)(base + first) * Ep + col] = ((wsum[0][col] + wsum[1][col]) + wsum[2][col]) + wsum[3][col];
            }
        if (hit) pending = false;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_embed_grad(const int* __restrict__ X, int n, const float* __restrict__ dX,
                                                    int Ep, float* __restrict__ dEmb, int* __restrict__ tok_first, int* __restrict__ tok_count,
                                                    const float* __restrict__ part) {
    __shared__ unsigned long long mask[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = X[r];
    int want = n + 1;                           // occurrences to find (no table: scan to the end)
    if (tok_first != nullptr) {
        if (tok_first[tok] != r) return;        // (uniform: every thread reads the same word)
        want = tok_count[tok];
        if (part != nullptr && want > EMBED_HEAVY) {        // heavy token: the chunks' partial rows (k_embed_grad_chunks), chunk by chunk
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int found = 0;
            for (int base = r & ~255; base < n && found < want; base += 256) {
                const int i = base + tid;
                const bool hit = (i < n) && (X[i] == tok);
                const unsigned long long bal = __ballot(hit);
                if (lane == 0) mask[wave] = bal;
                __syncthreads();
                int first = -1;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    found += __popcll(mask[w]);
                    if (first < 0 && mask[w]) first = 64 * w + __ffsll((long long)mask[w]) - 1;
                }
                if (first >= 0) {
                    const float* src = part + (long long)(base + first) * Ep;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (tid + 256 * c < Ep) acc[c] += src[tid + 256 * c];
                }
                __syncthreads();
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (tid + 256 * c < Ep) dEmb[(long long)tok * Ep + tid + 256 * c] = acc[c];
            if (tid == 0) { tok_first[tok] = 0x7FFFFFFF; tok_count[tok] = 0; }
            return;
        }
    } else {
        // earlier duplicate? then another block owns this token
        int dup = 0;
        for (int i = tid; i < r; i += 256) dup |= (X[i] == tok);
        if (__syncthreads_or(dup)) return;
    }

    float acc[4] = {0.f, 0.f, 0.f, 0.f};       // columns tid, tid+256, ... (Ep <= 1024)
    int found = 0;
    for (int base = r; base < n && found < want; base += 256) {
        const int i = base + tid;
        const bool hit = (i < n) && (X[i] == tok);
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) mask[wave] = bal;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mk = mask[w];
            found += __popcll(mk);
            while (mk) {
                const int bit = __ffsll((long long)mk) - 1;
                mk &= mk - 1;
                const float* src = dX + (long long)(base + 64 * w + bit) * Ep;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (tid + 256 * c < Ep) acc[c] += src[tid + 256 * c];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (tid + 256 * c < Ep) dEmb[(long long)tok * Ep + tid + 256 * c] = acc[c];
    if (tok_first != nullptr && tid == 0) { tok_first[tok] = 0x7FFFFFFF; tok_count[tok] = 0; }
}

// ---------------------------------------------------------------- global norm + Adam (K8 + K9)
__global__ __launch_bounds__(256) void k_sqnorm_partials(const float* __restrict__ x, long long n,
                                                         double* __restrict__ partials) {
    __shared__ double sh[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SQ_CHUNK;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long idx = base + 4 * (tid + 256 * i);
        if (idx + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + idx);
            s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        } else {
            for (long long j = idx; j < n && j < idx + 4; ++j) s += (double)x[j] * x[j];
        }
    }
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// clip_by_global_norm + exponential_decay + TF AdamOptimizer (reference
// src/models/lstm_baseline.py:77-87; SURVEY.md A.4: epsilon added to the un-corrected sqrt(v)).
// Every block re-derives the scalars from the same partials in the same order, so all blocks
// (and all ranks of an episode-parallel job) use bit-identical alpha and scale.
__global__ __launch_bounds__(256) void k_adam_update(const UpdateArgs a) {
    __shared__ double sh[4];
    __shared__ float s_scale, s_alpha;
    // the gradients of a step whose persistent recurrent kernel timed out are garbage, and a batch with an out-of-range
    // token is rejected as a whole (the reference would fail the feed): keep the parameters
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.consume != nullptr) {             // second launch of a split update: what the first one decided
        if (a.consume[0] == 0.0f) return;
        if (tid == 0) { s_scale = a.consume[1]; s_alpha = a.consume[2]; }
    } else {
    if ((a.err_flag != nullptr && *a.err_flag != 0) || a.tail[2] != 0.0f || a.tail[3] != 0.0f || a.tail[4] != 0.0f || a.tail[5] != 0.0f) {
        if (a.publish != nullptr && blockIdx.x == 0 && tid == 0) a.publish[0] = 0.0f;
        return;
    }
    double s = 0.0;
    for (int i = tid; i < a.n_partials; i += 256) s += a.partials[i];
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) {
        double sq = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * (double)a.grad_scale * (double)a.grad_scale;
        if (a.use_slices) sq += (double)a.tail[0] * (double)a.grad_scale * (double)a.grad_scale;
        const double gnorm = sqrt(sq);
        const double clip = (double)a.clip;
        s_scale = (float)(clip / fmax(gnorm, clip) * (double)a.grad_scale);
        const long long step = *a.step;
        const double t = (double)(step + 1);
        const double lr_s = (double)a.lr * pow(0.5, (double)step / (double)a.n_decay);
        s_alpha = (float)(lr_s * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
        if (a.gnorm_out != nullptr && blockIdx.x == 0) *a.gnorm_out = (float)gnorm;
        if (a.publish != nullptr && blockIdx.x == 0) { a.publish[0] = 1.0f; a.publish[1] = s_scale; a.publish[2] = s_alpha; }
    }
    }
    __syncthreads();
    const float scale = s_scale, alpha = s_alpha;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n4; i += (long long)gridDim.x * 256) {
        float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float4 p = reinterpret_cast<float4*>(a.p)[i];
#define FSMG_ADAM1(c)                                         \
        {                                                     \
            const float gc = g.c * scale;                     \
            m.c = b1 * m.c + (1.0f - b1) * gc;                \
            v.c = b2 * v.c + (1.0f - b2) * gc * gc;           \
            p.c = p.c - alpha * m.c / (sqrtf(v.c) + eps);     \
        }
        FSMG_ADAM1(x) FSMG_ADAM1(y) FSMG_ADAM1(z) FSMG_ADAM1(w)
#undef FSMG_ADAM1
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        reinterpret_cast<float4*>(a.p)[i] = p;
    }
}

// Inner-loop update of the MAML-style step (cfg-E; oracle/lstm_oracle.py maml_adapt): p <- p - lr * clip_by_global_norm(g).
// Same norm bookkeeping as k_adam_update (every block re-derives the scale from the same partials in the same order);
// Adam state and global_step are not touched.
__global__ __launch_bounds__(256) void k_sgd_update(const UpdateArgs a) {
    __shared__ double sh[4];
    __shared__ float s_scale;
    if ((a.err_flag != nullptr && *a.err_flag != 0) || a.tail[2] != 0.0f || a.tail[3] != 0.0f || a.tail[4] != 0.0f || a.tail[5] != 0.0f) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < a.n_partials; i += 256) s += a.partials[i];
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) {
        double sq = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (a.use_slices) sq += (double)a.tail[0];
        const double gnorm = sqrt(sq), clip = (double)a.clip;
        s_scale = (float)(clip / fmax(gnorm, clip)) * a.lr;
        if (a.gnorm_out != nullptr && blockIdx.x == 0) *a.gnorm_out = (float)gnorm;
    }
    __syncthreads();
    const float step = s_scale;
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n4; i += (long long)gridDim.x * 256) {
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        p.x -= step * g.x; p.y -= step * g.y; p.z -= step * g.z; p.w -= step * g.w;
        reinterpret_cast<float4*>(a.p)[i] = p;
    }
}

// Last kernel of a train step.  A step that must not count -- a persistent recurrent kernel timed out on this rank
// (*err_flag == 2) or on another one (tail[2], all-reduced), or a token id was out of range (*err_flag == 1) -- is
// tallied in `counters` (host-mapped: [0] time-outs, [1] token-range rejections; the host compares them with what it has
// seen, no read-back needed) and the flag is CLEARED, so that the next step starts clean instead of every later launch
// bailing out on a stale flag.
__global__ void k_step_increment(const StepIncArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    step_increment_body(a);
}

// flag_src != nullptr: also dst[2] = (*flag_src == 2), dst[3] = (*flag_src == 1) -- the "a persistent recurrent kernel timed out" /
// "a token id was out of range" indicators travel
// in the gradient tail, so after the all-reduce EVERY rank of an episode-parallel step knows that some rank's gradients
// are garbage
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partials, int n, float* dst,
                                                      const int* flag_src, const float* __restrict__ ce, int ce_n, float* loss_out) {
    sum_partials_body(partials, n, dst, flag_src, ce, ce_n, loss_out);
}

// ---------------------------------------------------------------- shader-clock probe
// One wave that watches two counters for `ticks` of the constant 100 MHz real-time counter (s_memrealtime): out[0] = elapsed shader-clock
// ticks (s_memtime), out[1] = elapsed real-time ticks.  Launched on a stream of its own BESIDE a few train steps it reports the clock the
// chip sustains under the step's load -- what a fraction of the 2.4 GHz spec peak has to be read against (bench.py: roofline.clock_ghz).
__global__ void k_clock_probe(long long ticks, unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while ((long long)(r1 - r0) < ticks) { __builtin_amdgcn_s_sleep(64); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_copy_words(uint4* __restrict__ dst, const uint4* __restrict__ src, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

// Self-check of the XCD-partitioned order (api_forward.hip: xov_selfcheck): the logits of the gated work-queue projection against the same
// GEMM recomputed behind the chain, word for word.  A difference is counted per wave (one system-scope atomic) and raises the flag.
__global__ __launch_bounds__(256) void k_compare_words(const uint4* __restrict__ a, const uint4* __restrict__ b, long long n4, int* err_flag,
                                                       unsigned long long* counter) {
    if (*err_flag != 0) return;         // the step is being skipped already (a gate or a chain timed out: the logits are incomplete by design)
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint4 x = a[i], y = b[i];
        bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    const unsigned long long votes = __ballot(bad != 0);
    if (votes == 0) return;
    unsigned total = bad;
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if ((threadIdx.x & 63) == __ffsll((long long)votes) - 1) {
        atomicAdd_system(counter, (unsigned long long)total);
        *err_flag = 2;
        __threadfence_system();
    }
}

// See launch_queue_probe (fsmg_kernels.h): which of two streams' kernels can overlap is decided by the hardware queues the runtime
// mapped the streams to -- a process has GPU_MAX_HW_QUEUES (4) of them and hands them out round-robin.
__global__ void k_queue_probe(int* flag, int* out, int role, long long ticks) {
    if (threadIdx.x != 0) return;
    if (role == 1) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int seen = 0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    out[0] = seen;
}

// ---------------------------------------------------------------- unigram baseline
// Reference src/models/unigram_model.py:26-39: word_count (alpha = 1) + scatter_add of ones, prob = gather / reduce_sum,
// loss = -mean(log prob).  Counts are integers (unsigned atomics: exact and order-independent), handed to the caller as floats.
__global__ void k_unigram_update(const int* __restrict__ words, long long n, unsigned* __restrict__ counts, int vocab, int* err_flag) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int w = words[i];
        if (w < 0 || w >= vocab) { atomicOr(err_flag, 1); continue; }
        atomicAdd(counts + w, 1u);
    }
}
// one block: total = sum(counts) (exact in double), then -mean(log(float(count[w]) / float(total))) in the reference's fp32
// arithmetic per word, accumulated in double in a fixed order
__global__ __launch_bounds__(1024) void k_unigram_nll(const int* __restrict__ words, long long n, const unsigned* __restrict__ counts, int vocab,
                                                      float* out, int* err_flag) {
    __shared__ double sh[16];
    __shared__ float s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < vocab; i += 1024) s += (double)counts[i];
    s = wave_sum_d(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < 16; ++w) t += sh[w]; s_total = (float)t; }
    __syncthreads();
    const float total = s_total;
    double l = 0.0;
    for (long long i = tid; i < n; i += 1024) {
        const int w = words[i];
        if (w < 0 || w >= vocab) { atomicOr(err_flag, 1); continue; }
        l += (double)logf((float)counts[w] / total);
    }
    l = wave_sum_d(l);
    __syncthreads();
    if (lane == 0) sh[wave] = l;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < 16; ++w) t += sh[w]; out[0] = (float)(-t / (double)(n > 0 ? n : 1)); out[1] = total; }
}
// argmax of the counts, lowest index on ties (np.argmax)
__global__ __launch_bounds__(1024) void k_unigram_argmax(const unsigned* __restrict__ counts, int vocab, int* out) {
    __shared__ unsigned sc[1024];
    __shared__ int si[1024];
    const int tid = threadIdx.x;
    unsigned best = 0; int bi = 0x7FFFFFFF;
    for (int i = tid; i < vocab; i += 1024) { const unsigned c = counts[i]; if (c > best || (c == best && i < bi)) { best = c; bi = i; } }
    sc[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            const unsigned c = sc[tid + o]; const int i = si[tid + o];
            if (c > sc[tid] || (c == sc[tid] && i < si[tid])) { sc[tid] = c; si[tid] = i; }
        }
        __syncthreads();
    }
    if (tid == 0) *out = si[0];
}

// ---------------------------------------------------------------- greedy decode (K10)
// One cell step for one sequence: z = x*Kx + h*Kh + b over packed gate columns; block nb owns
// units 4nb..4nb+3 (16 packed columns); 256 threads split the (in + Hp) reduction.
__global__ __launch_bounds__(256) void k_decode_cell(const float* __restrict__ Kx, int in_dim,
                                                     const float* __restrict__ Kh, const float* __restrict__ bias,
                                                     const float* __restrict__ x, const float* __restrict__ h_in,
                                                     float* __restrict__ h_out, float* __restrict__ c, int Hp) {
    __shared__ float part[16][17];
    const int nb = blockIdx.x, tid = threadIdx.x;
    const int col = tid & 15, slice = tid >> 4;         // 16 k-slices x 16 columns
    const int G4 = 4 * Hp;
    float s = 0.0f;
    for (int k = slice; k < in_dim; k += 16) s += x[k] * Kx[(long long)k * G4 + 16 * nb + col];
    for (int k = slice; k < Hp; k += 16) s += h_in[k] * Kh[(long long)k * G4 + 16 * nb + col];
    part[slice][col] = s;
    __syncthreads();
    if (tid < 4) {
        float zg[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int cc = 4 * gi + tid;
            float t = bias[16 * nb + cc];
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) t += part[sl][cc];
            zg[gi] = t;
        }
        const int u = 4 * nb + tid;
        const float si = 1.0f / (1.0f + expf(-zg[0])), tj = tanhf(zg[1]);
        const float sf = 1.0f / (1.0f + expf(-(zg[2] + 1.0f))), so = 1.0f / (1.0f + expf(-zg[3]));
        const float cn = c[u] * sf + si * tj;
        c[u] = cn;
        h_out[u] = tanhf(cn) * so;
    }
}

// logits = h*W + b over n_vocab columns, argmax (lowest index on ties, like np.argmax).
// Stage 1: each block handles 256 columns -> (max, idx) per block; stage 2: one block reduces.
__global__ __launch_bounds__(256) void k_decode_logits(const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, const float* __restrict__ h,
                                                       int Hp, int n_vocab, float* __restrict__ blk_max,
                                                       int* __restrict__ blk_idx) {
    __shared__ float smax[256];
    __shared__ int sidx[256];
    const int tid = threadIdx.x, v = blockIdx.x * 256 + tid;
    float s = -INFINITY;
    if (v < n_vocab) {
        s = bias[v];
        for (int k = 0; k < Hp; ++k) s += h[k] * W[(long long)k * ldw + v];
    }
    smax[tid] = s; sidx[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float b = smax[tid + o];
            const int bi = sidx[tid + o];
            if (b > smax[tid] || (b == smax[tid] && bi < sidx[tid])) { smax[tid] = b; sidx[tid] = bi; }
        }
        __syncthreads();
    }
    if (tid == 0) { blk_max[blockIdx.x] = smax[0]; blk_idx[blockIdx.x] = sidx[0]; }
}
__global__ void k_decode_pick(const float* blk_max, const int* blk_idx, int nblk, int* out_token) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float best = blk_max[0]; int bi = blk_idx[0];
        for (int i = 1; i < nblk; ++i)
            if (blk_max[i] > best || (blk_max[i] == best && blk_idx[i] < bi)) { best = blk_max[i]; bi = blk_idx[i]; }
        *out_token = bi;
    }
}

}  // namespace

hipError_t launch_token_prep(hipStream_t s, const int* support, int n_support, const int* query, int n_query,
                             int T, int vocab, int start_word, int* X, int* Y, int* err_flag, int* tok_first, int* tok_count) {
    const long long total = (long long)(n_support + n_query) * T;
    if (total <= 0) return hipSuccess;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_token_prep, dim3(blocks), dim3(256), 0, s, support, n_support, query, n_query, T, vocab,
                       start_word, X, Y, err_flag, tok_first, tok_count);
    return hipGetLastError();
}

hipError_t launch_ce_rows(hipStream_t s, const float* logits, int ld, int rows, int n_vocab, const int* tgt,
                          float* lse, float* ce, float* dlogits, float inv_n) {
    if (rows <= 0) return hipSuccess;
    // non-temporal dlogits stores (A/B settled in round 3 for a separate dlogits buffer; FSMG_CE_NT=0: regular stores -- the A/B of
    // round 5 for the in-place form, where the lines being written are the lines just read)
    static const int nt = std::getenv("FSMG_CE_NT") ? (std::atoi(std::getenv("FSMG_CE_NT")) != 0) : 1;
    if (ld <= 6 * 1024) {
        if (nt) hipLaunchKernelGGL((k_ce_rows_reg<6, true>), dim3(rows), dim3(256), 0, s, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
        else hipLaunchKernelGGL((k_ce_rows_reg<6, false>), dim3(rows), dim3(256), 0, s, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
    } else if (ld <= 12 * 1024) {
        if (nt) hipLaunchKernelGGL((k_ce_rows_reg<12, true>), dim3(rows), dim3(256), 0, s, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
        else hipLaunchKernelGGL((k_ce_rows_reg<12, false>), dim3(rows), dim3(256), 0, s, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
    }
    else hipLaunchKernelGGL(k_ce_rows, dim3(rows), dim3(256), 0, s, logits, ld, n_vocab, tgt, lse, ce, dlogits, inv_n);
    return hipGetLastError();
}

#ifdef FSMG_EXPERIMENTS
hipError_t launch_ce_rows_gated(hipStream_t s, const float* logits, int ld, int rows, int n_vocab, const int* tgt, float* lse, float* ce,
                                float* dlogits, float inv_n, const int* done, int done_expect, int tile_rows, int* err_flag, int spin_cap, int blocks, int* next_row) {
    if (rows <= 0) return hipSuccess;
    if (ld > 12 * 1024 || done == nullptr || err_flag == nullptr || next_row == nullptr || blocks <= 0) return hipErrorInvalidValue;
    blocks = std::min(blocks, rows);
    if (ld <= 6 * 1024) hipLaunchKernelGGL((k_ce_rows_gated<6, true>), dim3(blocks), dim3(256), 0, s, logits, ld, rows, n_vocab, tgt, lse, ce, dlogits, inv_n, done, done_expect, tile_rows, err_flag, spin_cap, next_row);
    else hipLaunchKernelGGL((k_ce_rows_gated<12, true>), dim3(blocks), dim3(256), 0, s, logits, ld, rows, n_vocab, tgt, lse, ce, dlogits, inv_n, done, done_expect, tile_rows, err_flag, spin_cap, next_row);
    return hipGetLastError();
}
#endif

hipError_t launch_ce_combine(hipStream_t s, const float2* part, int nparts, const float* tgt_logit, int rows, float* ce) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ce_combine, dim3((rows + 3) / 4), dim3(256), 0, s, part, nparts, tgt_logit, rows, ce);
    return hipGetLastError();
}

hipError_t launch_ce_finish(hipStream_t s, const float2* part, int nparts, const int* tgt, int rows, float inv_n,
                            float* E, int ld, float* lse, float* ce, float* crow, const float* hs, float* hs_scaled, int hp,
                            int* err_flag, long long* range_counter) {
    if (rows <= 0) return hipSuccess;
    if ((hp & 3) != 0 || err_flag == nullptr || range_counter == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ce_finish, dim3((rows + 3) / 4), dim3(256), 0, s, part, nparts, tgt, rows, inv_n, E, ld, lse, ce, crow,
                       hs, hs_scaled, hp, err_flag, (unsigned long long*)range_counter);
    return hipGetLastError();
}

hipError_t launch_scale_rows(hipStream_t s, float* C, const float* row_scale, int M, int N) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if ((N & 3) != 0) return hipErrorInvalidValue;
    const long long n4 = (long long)M * N / 4;
    hipLaunchKernelGGL(k_scale_rows, dim3((int)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, C, row_scale, n4, N / 4);
    return hipGetLastError();
}

hipError_t launch_loss_reduce(hipStream_t s, const float* ce, int T, int B, int rows_per_group, int ngroups,
                              float* out) {
    hipLaunchKernelGGL(k_loss_reduce, dim3(ngroups), dim3(256), 0, s, ce, T, B, rows_per_group, out);
    return hipGetLastError();
}

hipError_t launch_embed_grad(hipStream_t s, const int* X, int n, const float* dX, int Ep, float* dEmb, int* tok_first, int* tok_count, float* part,
                             const SumPartialsArgs* sum) {
    if (n <= 0) return sum != nullptr ? hipErrorInvalidValue : hipSuccess;
    if (Ep > 1024) return hipErrorInvalidValue;
    if (tok_first == nullptr || tok_count == nullptr || (Ep & 3) != 0 || ((uintptr_t)dX & 15) != 0) part = nullptr;       // (the two-level sum of heavy tokens goes by the occurrence table; rows as float4)
    if (part == nullptr && sum != nullptr) return hipErrorInvalidValue;
    if (part != nullptr) {
        SumPartialsArgs sp{};
        if (sum != nullptr) sp = *sum;
        hipLaunchKernelGGL(k_embed_grad_chunks, dim3((n + 255) / 256 + (sum != nullptr ? 1 : 0)), dim3(256), 0, s, X, n, dX, Ep, part, tok_count, sp);
    }
    hipLaunchKernelGGL(k_embed_grad, dim3(n), dim3(256), 0, s, X, n, dX, Ep, dEmb, tok_first, tok_count, part);
    return hipGetLastError();
}

hipError_t launch_multi_op(hipStream_t s, const MultiOps& r) {
    if (r.count <= 0) return hipSuccess;
    long long bx = 1;
    for (int k = 0; k < r.count; ++k) {
        const long long want = r.op[k].kind == MULTI_FILL ? (r.op[k].n / 4 + 255) / 256 : r.op[k].kind == MULTI_MEAN ? 1 :
                               (r.op[k].sq == nullptr ? (r.op[k].n / 4 + 255) / 256 : (r.op[k].n + SQ_CHUNK - 1) / SQ_CHUNK);
        bx = std::max(bx, want);
    }
    bx = std::max<long long>(1, std::min<long long>(bx, 1024));
    hipLaunchKernelGGL(k_multi_op, dim3((int)bx, r.count), dim3(256), 0, s, r);
    return hipGetLastError();
}

hipError_t launch_gather_rows(hipStream_t s, const int* table, const int* idx, int n_rows, int T, int n_songs, int* out, int* err_flag) {
    if (n_rows <= 0) return hipSuccess;
    const long long total = (long long)n_rows * T;
    hipLaunchKernelGGL(k_gather_rows, dim3((int)std::min<long long>((total + 255) / 256, 1024)), dim3(256), 0, s, table, idx, n_rows, T, n_songs, out, err_flag);
    return hipGetLastError();
}

hipError_t launch_fill32_if(hipStream_t s, const int* cond, void* p, uint32_t word, long long n_words) {
    if (n_words <= 0) return hipSuccess;
    long long blocks = (n_words / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill32_if, dim3((int)blocks), dim3(256), 0, s, cond, (uint32_t*)p, word, n_words);
    return hipGetLastError();
}

hipError_t launch_fill32(hipStream_t s, void* p, uint32_t word, long long n_words) {
    if (n_words <= 0) return hipSuccess;
    long long blocks = (n_words / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill32, dim3((int)blocks), dim3(256), 0, s, (uint32_t*)p, word, n_words);
    return hipGetLastError();
}

int sqnorm_blocks(long long n) { return (int)((n + SQ_CHUNK - 1) / SQ_CHUNK); }

hipError_t launch_sqnorm_partials(hipStream_t s, const float* x, long long n, double* partials) {
    const int nb = sqnorm_blocks(n);
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sqnorm_partials, dim3(nb), dim3(256), 0, s, x, n, partials);
    return hipGetLastError();
}

hipError_t launch_adam_update(hipStream_t s, const UpdateArgs& a) {
    long long n4 = a.n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_adam_update, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sgd_update(hipStream_t s, const UpdateArgs& a) {
    long long n4 = a.n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sgd_update, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_step_increment(hipStream_t s, const StepIncArgs& a) {
    hipLaunchKernelGGL(k_step_increment, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sum_partials(hipStream_t s, const double* partials, int n, float* dst, const int* flag_src,
                               const float* ce, int ce_n, float* loss_out) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partials, n, dst, flag_src, ce, ce_n, loss_out);
    return hipGetLastError();
}

hipError_t launch_queue_probe(hipStream_t s, int* flag, int* out, int role, long long realtime_ticks) {
    hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, s, flag, out, role, realtime_ticks);
    return hipGetLastError();
}
hipError_t launch_clock_probe(hipStream_t s, long long realtime_ticks, unsigned long long* out) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, realtime_ticks, out);
    return hipGetLastError();
}
hipError_t launch_copy_words(hipStream_t s, void* dst, const void* src, long long n_words) {
    const long long n4 = n_words >> 2;
    if (n4 <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_words, dim3((int)std::min<long long>((n4 + 255) / 256, 4096)), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n4);
    return hipGetLastError();
}
hipError_t launch_compare_words(hipStream_t s, const void* a, const void* b, long long n_words, int* err_flag, long long* counter) {
    const long long n4 = n_words >> 2;
    if (n4 <= 0) return hipSuccess;
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_compare_words, dim3(blocks), dim3(256), 0, s, (const uint4*)a, (const uint4*)b, n4, err_flag, (unsigned long long*)counter);
    return hipGetLastError();
}
hipError_t launch_unigram_update(hipStream_t s, const int* words, long long n, unsigned* counts, int vocab, int* err_flag) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_unigram_update, dim3((int)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, s, words, n, counts, vocab, err_flag);
    return hipGetLastError();
}
hipError_t launch_unigram_nll(hipStream_t s, const int* words, long long n, const unsigned* counts, int vocab, float* out, int* err_flag) {
    hipLaunchKernelGGL(k_unigram_nll, dim3(1), dim3(1024), 0, s, words, n, counts, vocab, out, err_flag);
    return hipGetLastError();
}
hipError_t launch_unigram_argmax(hipStream_t s, const unsigned* counts, int vocab, int* out) {
    hipLaunchKernelGGL(k_unigram_argmax, dim3(1), dim3(1024), 0, s, counts, vocab, out);
    return hipGetLastError();
}

hipError_t launch_decode_cell(hipStream_t s, const float* Kx, int in_dim, const float* Kh, const float* bias,
                              const float* x, const float* h_in, float* h_out, float* c, int Hp) {
    // h_in is read by every block while every block writes its units of h_out: they must differ
    hipLaunchKernelGGL(k_decode_cell, dim3(Hp / 4), dim3(256), 0, s, Kx, in_dim, Kh, bias, x, h_in, h_out, c, Hp);
    return hipGetLastError();
}

hipError_t launch_decode_argmax(hipStream_t s, const float* W, int ldw, const float* bias, const float* h,
                                int Hp, int n_vocab, int* out_token, float* scratch) {
    const int nblk = (n_vocab + 255) / 256;
    float* blk_max = scratch;
    int* blk_idx = reinterpret_cast<int*>(scratch + nblk);
    hipLaunchKernelGGL(k_decode_logits, dim3(nblk), dim3(256), 0, s, W, ldw, bias, h, Hp, n_vocab, blk_max, blk_idx);
    hipLaunchKernelGGL(k_decode_pick, dim3(1), dim3(64), 0, s, blk_max, blk_idx, nblk, out_token);
    return hipGetLastError();
}

}  // namespace fsmg
