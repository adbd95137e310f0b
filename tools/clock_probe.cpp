// What shader clock does the chip actually run at while the fp32 GEMM is resident?  One wave per XCD-ish (64 blocks)
// samples s_memtime (shader-clock ticks) against wall_clock64 (constant 100 MHz) over ~200 us, alone and while the
// cfg-B dW GEMM (links build/gemm.o) or a register-only MFMA loop runs on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifew-shot-music-generation_amd/csrc -Iinclude -c tools/clock_probe.cpp -o /tmp/cp.o
//   hipcc --offload-arch=gfx950 /tmp/cp.o few-shot-music-generation_amd/build/gemm.o -o tools/clock_probe.bin
#include "fsmg_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace fsmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_probe(unsigned long long* out, unsigned long long wall_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long w = w0;
    while (w - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(32); w = wall_clock64(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w - w0;
}
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
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
static float* dev_random(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
int main() {
    hipStream_t sp, sg; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    unsigned long long* d; CK(hipMalloc(&d, 64 * 16));
    float* A = dev_random((size_t)5760 * 512, 1); float* B = dev_random((size_t)5760 * 10004, 2);
    float* C; CK(hipMalloc(&C, (size_t)3 * 512 * 10004 * 4)); float* sink; CK(hipMalloc(&sink, 4096));
    GemmArgs g{};
    g.A = A; g.lda = 512; g.B = B; g.ldb = 10004; g.C = C; g.ldc = 10004; g.M = 512; g.N = 10004; g.K = 5760; g.ksplit = 3; g.c_slab = (long long)512 * 10004;
    g.bx3 = getenv("BX3") ? atoi(getenv("BX3")) : 0;          // BX3=1: the bf16-split GEMM as the co-runner
    const char* names[] = {"idle chip", "beside the dW GEMM (random operands)", "beside a register-only MFMA loop", "after 50 ms of GEMMs, still running"};
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 1) for (int r = 0; r < 6; ++r) CK(launch_gemm(sg, OP_XC, OP_XC, g, 0));
        if (mode == 2) hipLaunchKernelGGL(k_mfma, dim3(256 * 3), dim3(256), 0, sg, sink, 3000);
        if (mode == 3) { for (int r = 0; r < 120; ++r) CK(launch_gemm(sg, OP_XC, OP_XC, g, 0)); hipEvent_t ev; CK(hipEventCreate(&ev)); for (int r = 0; r < 100; ++r) CK(launch_gemm(sg, OP_XC, OP_XC, g, 0)); CK(hipEventRecord(ev, sg)); for (int r = 0; r < 20; ++r) CK(launch_gemm(sg, OP_XC, OP_XC, g, 0)); CK(hipEventSynchronize(ev)); }
        hipLaunchKernelGGL(k_probe, dim3(64), dim3(64), 0, sp, d, 20000ull);     // 200 us of wall clock
        CK(hipStreamSynchronize(sp));
        const bool still = hipStreamQuery(sg) == hipErrorNotReady;
        unsigned long long h[128]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int b = 0; b < 64; ++b) f.push_back((double)h[2 * b] / ((double)h[2 * b + 1] / 100.0));   // ticks per us = MHz
        std::sort(f.begin(), f.end());
        printf("%-42s: shader clock min %.0f  median %.0f  max %.0f MHz%s\n", names[mode], f.front(), f[32], f.back(), (mode == 0 || still) ? "" : "  [co-runner ended early]");
        CK(hipStreamSynchronize(sg));
    }
    return 0;
}
