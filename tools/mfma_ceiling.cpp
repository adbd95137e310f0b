// Microbenchmark: what fraction of the 157.3 TF fp32 MFMA datasheet peak a kernel can sustain on this
// MI355X at all (power / clock limited), and what feeding the operands from LDS costs on top of that.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.cpp -o /tmp/mfma_ceiling
// Variants: REG = operands live in registers (pure MFMA issue); LDS = every k-pair reads its four operand
// floats from LDS exactly like k_gemm's inner loop (ds_read2_b32 x2 -> 4 MFMAs), no barriers, no HBM.
// Swept over resident waves per SIMD (1..4) and over run length (short bursts run at boost clock, long
// runs at the sustained clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FROM_LDS>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float seed, unsigned long long* trace) {
    __shared__ float lds[2 * 16 * 132];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long t_begin = wall_clock64();
    for (int i = tid; i < 2 * 16 * 132; i += 256) {
        // seed > 1: full-entropy operands in [-0.5, 0.5) (what a real GEMM feeds the multipliers: data toggling sets the
        // power draw and with it the sustained clock); otherwise 8 small integers
        unsigned hsh = (unsigned)(i + 1) * 2654435761u + blockIdx.x * 40503u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        lds[i] = (seed > 1.0f) ? ((hsh >> 8) & 0xffff) / 65536.0f - 0.5f : seed * (float)(i & 7);
    }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = seed + lane, a1 = seed - lane, b0 = seed * 2.f, b1 = seed * 3.f;
    const float* ap = lds + (lane >> 5) * 132 + (lane & 31);
    const float* bp = lds + 16 * 132 + (lane >> 5) * 132 + (lane & 31);
    if (FROM_LDS == 2) {
        float na0 = ap[0], na1 = ap[32], nb0 = bp[0], nb1 = bp[32];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
                const int kn = (kk + 2) & 15;
                na0 = ap[kn * 132]; na1 = ap[kn * 132 + 32];
                nb0 = bp[kn * 132]; nb1 = bp[kn * 132 + 32];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
        }
    } else
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            if (FROM_LDS) {
                a0 = ap[kk * 132]; a1 = ap[kk * 132 + 32];
                b0 = bp[kk * 132]; b1 = bp[kk * 132 + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (FROM_LDS) asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;      // keep the accumulators alive
    if (trace != nullptr && tid == 0) {       // where and when this workgroup ran: HW_ID (reg 4), XCC_ID (reg 20), 100 MHz wall clock
        trace[4 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        trace[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        trace[4 * blockIdx.x + 2] = t_begin;
        trace[4 * blockIdx.x + 3] = wall_clock64();
    }
}

static int g_cap = 0, g_trace = 1;
template <int FROM_LDS>
static int run(const char* name, float* out, int blocks_per_cu, int iters, int reps, float seed = 1.0f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    // dynamic LDS pad: exactly blocks_per_cu blocks fit in a CU's 160 KiB, so the dispatcher cannot stack a CU deeper
    const int dyn = g_cap ? (160 * 1024 / blocks_per_cu) - 2 * 16 * 132 * 4 - 512 : 0;
    CK(hipFuncSetAttribute((const void*)k_mfma<FROM_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    hipLaunchKernelGGL(k_mfma<FROM_LDS>, dim3(grid), dim3(256), dyn, 0, out, iters, seed, (unsigned long long*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_mfma<FROM_LDS>, dim3(grid), dim3(256), dyn, 0, out, iters, seed, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 /*waves*/ * iters * 8 * 4 * (32.0 * 32 * 2 * 2) * reps;
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (g_trace) {
        unsigned long long* tr; CK(hipMalloc(&tr, grid * 32)); 
        hipLaunchKernelGGL(k_mfma<FROM_LDS>, dim3(grid), dim3(256), dyn, 0, out, iters, seed, tr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid * 4);
        CK(hipMemcpy(h.data(), tr, grid * 32, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_cu; unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < grid; ++b) { t0 = std::min(t0, h[4 * b + 2]); t1 = std::max(t1, h[4 * b + 3]); }
        int late = 0; double dur = 0;
        for (int b = 0; b < grid; ++b) {
            unsigned key = (unsigned)((h[4 * b + 1] & 0xf) << 16) | (unsigned)(h[4 * b] & 0xff00);   // xcc | se/sh/cu bits
            per_cu[key]++;
            if (h[4 * b + 2] - t0 > 1000) ++late;                    // started > 10 us after the first block
            dur += (double)(h[4 * b + 3] - h[4 * b + 2]);
        }
        std::map<int, int> hist; for (auto& kv : per_cu) hist[kv.second]++;
        printf("    placement: %zu distinct CUs;", per_cu.size());
        for (auto& kv : hist) printf(" %d CUs x %d blocks;", kv.second, kv.first);
        printf(" %d blocks started late; kernel span %.1f us, mean block %.1f us\n", late, (t1 - t0) / 100.0, dur / grid / 100.0);
        CK(hipFree(tr));
    }
    printf("%s  waves/SIMD %d  kernel %.3f ms x %3d : %6.1f TF  (%.1f %% of 157.3)\n", name, blocks_per_cu, ms / reps, reps, tf, 100 * tf / 157.3);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    g_trace = 0;
    for (int w = 1; w <= 3; ++w) {
        if (run<1>("LDS      random data", out, w, 600 / w, 200, 2.0f)) return 1;
        if (run<2>("LDS-pipe random data", out, w, 600 / w, 200, 2.0f)) return 1;
    }
    for (int w = 3; w <= 3; ++w) {
        if (g_cap) printf("capped: ");
        if (run<0>("REG      long ", out, w, 600 / w, 200)) return 1;     // ~0.1 s sustained
        if (run<1>("LDS      long ", out, w, 600 / w, 200)) return 1;
        if (run<2>("LDS-pipe long ", out, w, 600 / w, 200)) return 1;
    }
    return 0;
}
