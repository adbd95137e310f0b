// Which resource does a co-running GEMM take from the recurrent step kernels?  Links build/lstm_step.o and times
// chains of cfg-B forward / backward steps (B = 45, Hp = 512) on one stream while a synthetic co-runner occupies
// a second stream with 2 blocks of 256 threads per CU doing
//   mfma : back-to-back v_mfma_f32_32x32x2_f32 from registers (MFMA pipe only, no memory)
//   l2   : 16-byte global loads sweeping a 2 MiB L2-resident buffer (TA / L1 / L2 path only, no MFMA)
//   lds  : ds_read_b128 sweeps (LDS only)
//   hbm  : streaming 16-byte loads over 1 GiB (HBM)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifew-shot-music-generation_amd/csrc -Iinclude -c tools/step_contention.cpp -o /tmp/sc.o
//        hipcc --offload-arch=gfx950 /tmp/sc.o few-shot-music-generation_amd/build/lstm_step.o -o tools/step_contention.bin
#include "fsmg_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace fsmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PRIO>
__global__ __launch_bounds__(256) void k_co_mfma_prio(float* out, int iters) {
    __builtin_amdgcn_s_setprio(PRIO);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f - a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1.2345f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_co_mfma(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f - a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1.2345f) out[threadIdx.x] = s;
}
// the same MFMA stream with a pause after every group of 4 (GAP: 0 = s_nop 15, 1 = s_sleep 1, 2 = 16x16x4 MFMAs instead)
template <int GAP>
__global__ __launch_bounds__(256) void k_co_mfma_gap(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 sm[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) sm[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f - a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (GAP == 2) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    sm[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, sm[0], 0, 0, 0);
                    sm[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, sm[1], 0, 0, 0);
                    sm[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, sm[2], 0, 0, 0);
                    sm[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, sm[3], 0, 0, 0);
                }
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
                if (GAP == 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
                if (GAP == 1) __builtin_amdgcn_s_sleep(1);
            }
        }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += sm[i][r];
    if (s == 1.2345f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_co_loads(const float4* __restrict__ buf, size_t n_vec, float* out, int iters) {
    // n_vec is a power of two; every block walks the buffer with its own phase
    size_t i = ((size_t)blockIdx.x * 7919 * 256 + threadIdx.x) & (n_vec - 1);
    float4 s = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = buf[(i + (size_t)u * 256) & (n_vec - 1)];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        i = (i + 8 * 256) & (n_vec - 1);
    }
    if (s.x + s.y + s.z + s.w == 1.2345f) out[threadIdx.x] = s.x;
}
__global__ __launch_bounds__(256) void k_co_lds(float* out, int iters) {
    __shared__ float4 l[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) l[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float4 s = make_float4(0, 0, 0, 0);
    int i = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float4 v = l[(i + 256 * u) & 2047]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        i = (i + 64) & 2047;
    }
    if (s.x + s.y + s.z + s.w == 1.2345f) out[threadIdx.x] = s.x;
}

int main(int argc, char** argv) {
    const int B = 45, Hp = 512, G4 = 4 * Hp, T = 128, Bp16 = 48;
    hipStream_t sm, sc; CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    auto dz = [](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    float* khf = dz((size_t)2 * Hp * G4); float* hF = dz((size_t)(T + 1) * Bp16 * Hp); float* z = dz((size_t)T * B * G4);
    float* cs = dz((size_t)(T + 1) * B * Hp); float* hs = dz((size_t)(T + 1) * B * Hp);
    float* dzF = dz((size_t)2 * Bp16 * G4); float* dc = dz((size_t)B * Hp); float* dh = dz((size_t)T * B * Hp);
    float* sink = dz(4096);
    const size_t l2n = (2u << 20) / 16, hbmn = (1u << 30) / 16;
    float4* l2buf; CK(hipMalloc(&l2buf, l2n * 16)); CK(hipMemset(l2buf, 0, l2n * 16));
    float4* hbmbuf; CK(hipMalloc(&hbmbuf, hbmn * 16)); CK(hipMemset(hbmbuf, 0, hbmn * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto chain = [&](int which) {
        for (int t = 0; t < T; ++t) {
            if (which == 0) {
                LstmFwdArgs a{};
                a.KhF = khf; a.hF_prev = hF + (size_t)t * Bp16 * Hp; a.hF_next = hF + (size_t)(t + 1) * Bp16 * Hp;
                a.z = z + (size_t)t * B * G4; a.c_prev = cs + (size_t)t * B * Hp; a.c_next = cs + (size_t)(t + 1) * B * Hp;
                a.h_next = hs + (size_t)(t + 1) * B * Hp; a.B = B; a.Hp = Hp;
                CK(launch_lstm_fwd_step(sm, a));
            } else {
                const int tt = T - 1 - t;
                LstmBwdArgs a{};
                a.KhF = khf + (size_t)Hp * G4; a.dzF_next = (tt + 1 < T) ? dzF + (size_t)((tt + 1) & 1) * Bp16 * G4 : nullptr;
                a.dzF_cur = dzF + (size_t)(tt & 1) * Bp16 * G4; a.gates = z + (size_t)tt * B * G4;
                a.c_t = cs + (size_t)(tt + 1) * B * Hp; a.c_prev = cs + (size_t)tt * B * Hp; a.dc = dc; a.dh_top = dh + (size_t)tt * B * Hp;
                a.B = B; a.Hp = Hp;
                CK(launch_lstm_bwd_step(sm, a));
            }
        }
    };
    // real co-runner: the dW GEMM of cfg-B (M 512, N 10004, K 5760, split 3) back to back on the second stream
    float* gA = dz((size_t)5760 * 512); float* gB = dz((size_t)5760 * 10004); float* gC = dz((size_t)3 * 512 * 10004);
    hipEvent_t g0, g1; CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
    auto gemm_corun = [&](int cap, int reps) {
        GemmArgs g{};
        g.A = gA; g.lda = 512; g.B = gB; g.ldb = 10004; g.C = gC; g.ldc = 10004; g.M = 512; g.N = 10004; g.K = 5760;
        g.ksplit = 3; g.c_slab = (long long)512 * 10004;
        CK(hipEventRecord(g0, sc));
        for (int r = 0; r < reps; ++r) CK(launch_gemm(sc, OP_XC, OP_XC, g, gemm_lds_pad_for(cap)));
        CK(hipEventRecord(g1, sc));
    };
    const char* names[] = {"alone", "mfma", "l2", "lds", "hbm", "mfma+nop", "mfma+sleep", "mfma16x16", "mfma-short-blocks(2us)", "mfma-short-blocks(8us)", "mfma prio3"};
    const int bpc = argc > 1 ? atoi(argv[1]) : 2;
    // GEMM alone (no chain) for reference
    for (int cap = 1; cap <= 4; ++cap) {
        gemm_corun(cap, 4); CK(hipStreamSynchronize(sc));
        float gms; CK(hipEventElapsedTime(&gms, g0, g1));
        printf("dW GEMM alone, cap %d blocks/CU: %.3f ms (%.1f TF)\n", cap, gms / 4, 2.0 * 512 * 10004 * 5760 / (gms / 4 * 1e-3) / 1e12);
    }
    for (int which = 0; which < 2; ++which)
        for (int cap = 1; cap <= 4; ++cap) {
            chain(which); CK(hipStreamSynchronize(sm));
            const int reps = 12, steps = 2 * T;          // the GEMMs must outlast the chain: checked below
            gemm_corun(cap, reps);
            CK(hipEventRecord(e0, sm));
            chain(which); chain(which);
            CK(hipEventRecord(e1, sm));
            CK(hipEventSynchronize(e1));
            const bool covered = hipEventQuery(g1) == hipErrorNotReady;
            CK(hipStreamSynchronize(sc));
            float ms, gms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&gms, g0, g1));
            printf("%s chain beside the dW GEMM capped at %d blocks/CU: %.2f us/step over %d steps%s; %d GEMMs %.3f ms each incl. the uncontended tail\n", which ? "bwd" : "fwd", cap,
                   ms * 1000 / steps, steps, covered ? "" : " [GEMMs ended first: lower bound]", reps, gms / reps);
        }
    // per-wave phase stamps (instrumented build of the forward step) alone and beside the cap-2 GEMM
    {
        const int nblk = 128 * 3, nw = 4;
        unsigned long long* prof; CK(hipMalloc(&prof, (size_t)nblk * nw * 8 * 8));
        std::vector<unsigned long long> hp((size_t)nblk * nw * 8);
        for (int beside = 0; beside < 2; ++beside) {
            if (beside) gemm_corun(2, 6);
            std::vector<long long> ph[4], span;
            for (int rep = 0; rep < 24; ++rep) {
                const int t = rep;
                LstmFwdArgs a{};
                a.KhF = khf; a.hF_prev = hF + (size_t)t * Bp16 * Hp; a.hF_next = hF + (size_t)(t + 1) * Bp16 * Hp;
                a.z = z + (size_t)t * B * G4; a.c_prev = cs + (size_t)t * B * Hp; a.c_next = cs + (size_t)(t + 1) * B * Hp;
                a.h_next = hs + (size_t)(t + 1) * B * Hp; a.B = B; a.Hp = Hp;
                CK(launch_lstm_fwd_step(sm, a, prof));
                CK(hipStreamSynchronize(sm));
                CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
                if (rep < 4) continue;
                for (int b = 0; b < nblk; ++b) {
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int w = 0; w < nw; ++w) {
                        const unsigned long long* q = &hp[((size_t)b * nw + w) * 8];
                        for (int i = 0; i < 4; ++i) ph[i].push_back((long long)(q[i + 1] - q[i]));
                        lo = std::min(lo, q[0]); hi = std::max(hi, q[4]);
                    }
                    span.push_back((long long)(hi - lo));
                }
            }
            const bool covered = !beside || hipEventQuery(g1) == hipErrorNotReady;
            CK(hipStreamSynchronize(sc));
            auto med = [](std::vector<long long>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            printf("fwd step %s: per-wave ticks  loads+mfma med %lld p90 %lld | LDS partials med %lld p90 %lld | barrier med %lld p90 %lld | epilogue med %lld p90 %lld | block span med %lld p90 %lld%s\n",
                   beside ? "beside cap-2 GEMM" : "alone            ", med(ph[0], .5), med(ph[0], .9), med(ph[1], .5), med(ph[1], .9), med(ph[2], .5), med(ph[2], .9),
                   med(ph[3], .5), med(ph[3], .9), med(span, .5), med(span, .9), covered ? "" : " [GEMM ended early]");
        }
    }
    if (argc > 2)
    for (int which = 0; which < 2; ++which)
        for (int co = 0; co < 11; ++co) {
            if (co >= 2 && co <= 9) continue;
            chain(which); CK(hipStreamSynchronize(sm));
            // co-runner long enough to cover the chain (a few ms)
            if (co == 1) hipLaunchKernelGGL(k_co_mfma, dim3(256 * bpc), dim3(256), 0, sc, sink, 6000);
            if (co == 2) hipLaunchKernelGGL(k_co_loads, dim3(256 * bpc), dim3(256), 0, sc, l2buf, l2n, sink, 12000);
            if (co == 3) hipLaunchKernelGGL(k_co_lds, dim3(256 * bpc), dim3(256), 0, sc, sink, 60000);
            if (co == 4) hipLaunchKernelGGL(k_co_loads, dim3(256 * bpc), dim3(256), 0, sc, hbmbuf, hbmn, sink, 6000);
            if (co == 5) hipLaunchKernelGGL(k_co_mfma_gap<0>, dim3(256 * bpc), dim3(256), 0, sc, sink, 3000);
            if (co == 6) hipLaunchKernelGGL(k_co_mfma_gap<1>, dim3(256 * bpc), dim3(256), 0, sc, sink, 3000);
            // the same MFMA load as a stream of short-lived blocks: the co-runner's waves are then often YOUNGER than the step's
            if (co == 8) hipLaunchKernelGGL(k_co_mfma, dim3(256 * bpc * 1500), dim3(256), 0, sc, sink, 2);
            if (co == 9) hipLaunchKernelGGL(k_co_mfma, dim3(256 * bpc * 400), dim3(256), 0, sc, sink, 8);
            if (co == 10) hipLaunchKernelGGL(k_co_mfma_prio<3>, dim3(256 * bpc), dim3(256), 0, sc, sink, 6000);
            if (co == 7) hipLaunchKernelGGL(k_co_mfma_gap<2>, dim3(256 * bpc), dim3(256), 0, sc, sink, 6000);
            CK(hipEventRecord(e0, sm));
            chain(which);
            CK(hipEventRecord(e1, sm));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const bool still = hipStreamQuery(sc) == hipErrorNotReady;
            CK(hipStreamSynchronize(sc));
            printf("%s chain of %d steps, co-runner %-5s (%d blocks/CU): %.2f us/step%s\n", which ? "bwd" : "fwd", T, names[co], bpc, ms * 1000 / T,
                   (co == 0 || still) ? "" : "   [co-runner ended before the chain: lower bound]");
        }
    return 0;
}
