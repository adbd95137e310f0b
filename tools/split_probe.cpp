// Is v_dot2_f32_bf16 usable for the residuals of the three-way bf16 split (csrc/gemm.hip: split2)?
//   x - bf16_lo(p) == dot2(p, (-1, 0)) + x   and   y - bf16_hi(p) == dot2(p, (0, -1)) + y
// (1) bit-exactness against the shift / mask / subtract form over random values of every binade, denormals, huge values;
// (2) issue cost: s_memtime cycles per pair of values for both forms, one wave and four waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/split_probe.cpp -o tools/split_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;
}
__device__ __forceinline__ void split_ref(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(x, y);
    float rx, ry, sx, sy;
    asm("v_sub_f32 %0, %1, %2" : "=v"(rx) : "v"(x), "v"(p0 << 16));
    asm("v_sub_f32 %0, %1, %2" : "=v"(ry) : "v"(y), "v"(p0 & 0xffff0000u));
    p1 = cvt_pk_bf16(rx, ry);
    asm("v_sub_f32 %0, %1, %2" : "=v"(sx) : "v"(rx), "v"(p1 << 16));
    asm("v_sub_f32 %0, %1, %2" : "=v"(sy) : "v"(ry), "v"(p1 & 0xffff0000u));
    p2 = cvt_pk_bf16(sx, sy);
}
__device__ __forceinline__ void split_dot(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2, unsigned mlo, unsigned mhi) {
    p0 = cvt_pk_bf16(x, y);
    float rx, ry, sx, sy;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(rx) : "v"(p0), "v"(mlo), "v"(x));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ry) : "v"(p0), "v"(mhi), "v"(y));
    p1 = cvt_pk_bf16(rx, ry);
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(sx) : "v"(p1), "v"(mlo), "v"(rx));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(sy) : "v"(p1), "v"(mhi), "v"(ry));
    p2 = cvt_pk_bf16(sx, sy);
}
__global__ void k_check(const float* v, size_t n, unsigned long long* bad, unsigned* first) {
    const unsigned mlo = 0x0000BF80u, mhi = 0xBF800000u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = v[2 * i], y = v[2 * i + 1];
        unsigned a0, a1, a2, b0, b1, b2;
        split_ref(x, y, a0, a1, a2);
        split_dot(x, y, b0, b1, b2, mlo, mhi);
        if (a0 != b0 || a1 != b1 || a2 != b2) {
            if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(x); first[1] = __float_as_uint(y); first[2] = a1; first[3] = b1; first[4] = a2; first[5] = b2; }
        }
    }
}
template <int MODE>
__global__ void k_time(const float* v, unsigned* out, unsigned long long* ticks, int iters) {
    const unsigned mlo = 0x0000BF80u, mhi = 0xBF800000u;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = v[threadIdx.x * 8 + i];
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned p0, p1, p2;
            if (MODE == 0) split_ref(x[2 * j], x[2 * j + 1], p0, p1, p2); else split_dot(x[2 * j], x[2 * j + 1], p0, p1, p2, mlo, mhi);
            acc ^= p0 + p1 + p2;
            asm volatile("" : "+v"(x[2 * j]), "+v"(x[2 * j + 1]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    const size_t n = 1 << 26;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u; unsigned bits = s;
        s = s * 1664525u + 1013904223u; bits ^= s >> 7;
        if (((bits >> 23) & 0xff) == 0xff) bits &= ~(1u << 30);      // no Inf / NaN here
        memcpy(&h[i], &bits, 4);
    }
    // specials up front: zeros, denormals, largest finite, values that round up to the next binade in bf16
    const unsigned sp[] = {0u, 0x80000000u, 1u, 0x007fffffu, 0x00800000u, 0x7f7fffffu, 0xff7fffffu, 0x3f7fffffu, 0x3f80ffffu, 0x7f7f8000u, 0x00008000u, 0x00010000u, 0x7f000001u, 0x3effffffu};
    for (size_t i = 0; i < sizeof(sp) / 4; ++i) memcpy(&h[i], &sp[i], 4);
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    unsigned long long* bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    unsigned* first; CK(hipMalloc(&first, 64)); CK(hipMemset(first, 0, 64));
    hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, d, n, bad, first);
    CK(hipDeviceSynchronize());
    unsigned long long hb; unsigned hf[6];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 24, hipMemcpyDeviceToHost));
    printf("exactness over %zu pairs (all binades, denormals, specials): %llu mismatches", n / 2, hb);
    if (hb) printf("  first: x %08x y %08x  p1 ref %08x dot %08x  p2 ref %08x dot %08x", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5]);
    printf("\n");
    unsigned* out; CK(hipMalloc(&out, 4 * 256 * 4096)); unsigned long long* ticks; CK(hipMalloc(&ticks, 8 * 4096));
    for (int wpb : {64, 256, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = 2000;
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(wpb), 0, 0, d, out, ticks, iters);
            else hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(wpb), 0, 0, d, out, ticks, iters);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> t(256); CK(hipMemcpy(t.data(), ticks, 8 * 256, hipMemcpyDeviceToHost));
            double m = 0; for (auto v : t) m += v; m /= 256;
            printf("%s, %2d wave(s) per SIMD: %.1f cycles per pair of values and wave (%.1f per SIMD)\n", mode ? "dot2 form (7 instr / pair) " : "shift-mask form (11 / pair)", wpb / 256 ? wpb / 256 : 1, m / iters / 4, m / iters / 4 / (wpb / 256 ? wpb / 256 : 1));
        }
    }
    return 0;
}
