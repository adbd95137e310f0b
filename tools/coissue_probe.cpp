// Can VALU FMAs (with SGPR operands) issue in the shadow of back-to-back v_mfma_f32_4x4x1_16B_f32?  One wave per SIMD,
// 128 MFMAs (two accumulator chains) alone / 128 v_pk_fma_f32 alone / 256 v_fma_f32 alone / interleaved 1:1 and 1:2.
// hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.cpp -o tools/coissue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters, float s0, float s1) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    f32x2 p0 = {0, 0}, p1 = {0, 0};
    float q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    const float w = 1.0f + threadIdx.x * 1e-4f, h = threadIdx.x * 1e-3f;
    const f32x2 ww = {w, w};
    const f32x2 ss0 = {s0, s1}, ss1 = {s1, s0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (MODE == 0 || MODE >= 3) {
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:3" : "+v"(a0) : "v"(h), "v"(w));
            }
            if (MODE == 1 || MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(ww), "s"(ss0));
            if (MODE == 2 || MODE == 4) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q0) : "v"(w), "s"(s0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q1) : "v"(w), "s"(s1)); }
            if (MODE == 0 || MODE >= 3) {
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(a1) : "v"(h), "v"(w));
            }
            if (MODE == 1 || MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(ww), "s"(ss1));
            if (MODE == 2 || MODE == 4) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q2) : "v"(w), "s"(s1)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q3) : "v"(w), "s"(s0)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + p0[0] + p0[1] + p1[0] + p1[1] + q0 + q1 + q2 + q3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ACTIVE>
__global__ __launch_bounds__(256, 1) void k_half(float* out, unsigned long long* cyc, int iters) {
    float q0 = threadIdx.x * 1e-3f, q1 = 0.5f, q2 = 0.25f, q3 = 0.125f;
    const float w = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = 0, t1 = 0;
    if ((threadIdx.x & 63) < ACTIVE) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(q0) : "v"(w), "v"(q1));
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(q1) : "v"(w), "v"(q2));
                asm volatile("v_exp_f32 %0, %0" : "+v"(q2));
                asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(q3) : "v"(w), "v"(q0));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 256 + threadIdx.x] = q0 + q1 + q2 + q3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ACTIVE>
int run_half(float* out, unsigned long long* cyc) {
    const int iters = 50;
    hipLaunchKernelGGL((k_half<ACTIVE>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k_half<ACTIVE>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= 256;
    printf("192 v_fma + 64 v_exp (dependent mix), %2d of 64 lanes active: %7.1f ticks per 256 instructions\n", ACTIVE, m / iters);
    return 0;
}

template <int MODE>
int run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 50;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.5f, 0.25f);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.5f, 0.25f);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= 256;
    printf("%-58s %7.1f ticks per 128-MFMA-equivalent block (64 loop bodies)\n", name, m / iters);
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4 * 65536)); CK(hipMalloc(&cyc, 8 * 256));
    run<0>("128 MFMA 4x4x1 (2 chains) alone", out, cyc);
    run<1>("128 v_pk_fma_f32 (SGPR pair operand) alone", out, cyc);
    run<2>("256 v_fma_f32 (SGPR operand) alone", out, cyc);
    run<3>("128 MFMA + 128 v_pk_fma_f32 interleaved 1:1", out, cyc);
    run<4>("128 MFMA + 256 v_fma_f32 interleaved 1:2", out, cyc);
    run_half<64>(out, cyc); run_half<32>(out, cyc); run_half<16>(out, cyc);
    return 0;
}
// ---- does a wave64 VALU instruction with only 32 (or 16) active lanes issue faster?  (k_half below)
