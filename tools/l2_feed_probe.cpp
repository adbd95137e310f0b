// How many bytes per cycle can ONE CU pull through its vector L1 when every CU of the chip streams at once?
// (The bf16-split GEMMs settle at 14.7 B per cycle and CU whatever the kernel structure -- DESIGN.md section 4, round 3.)
// Every block streams a region with 16-byte loads per lane (1 KiB contiguous per wave instruction), unrolled 8 deep,
// data consumed by a running XOR; the region per XCD is sized to sit in its L2 (1 MiB), in the memory-side cache
// (32 MiB per XCD -> 256 MiB) or nowhere (2 GiB total), and the waves per CU are swept.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_feed_probe.cpp -o tools/l2_feed_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_stream(const uint4* __restrict__ base, size_t region_u4, int iters, unsigned* out, unsigned long long* ticks) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    const uint4* p = base + (size_t)xcc * region_u4;
    // a block walks the region in 1 KiB-per-wave pieces, blocks of the same XCD offset against each other
    const size_t nthr_stride = (size_t)blockDim.x;          // one block covers blockDim.x * 16 B per step
    size_t pos = ((size_t)blockIdx.x * 977 * nthr_stride + threadIdx.x) % region_u4;
    const size_t step = nthr_stride * 8;                    // 8 loads per iteration, contiguous per wave
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { size_t q = pos + j * nthr_stride; if (q >= region_u4) q -= region_u4; v[j] = p[q]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        pos += step; if (pos >= region_u4) pos -= region_u4;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t total = (size_t)2 << 30;
    uint4* d; CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total));
    unsigned* out; CK(hipMalloc(&out, 4 * 1024 * 1024)); unsigned long long* ticks; CK(hipMalloc(&ticks, 8 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t regions[] = {(size_t)1 << 20, (size_t)32 << 20, (size_t)256 << 20};
    const char* names[] = {"1 MiB per XCD (L2)", "32 MiB per XCD (memory-side cache)", "256 MiB per XCD (HBM)"};
    for (int r = 0; r < 3; ++r) {
        for (int wpc : {4, 8, 16}) {                                 // waves per CU = blocks of 256 threads per CU * 4
            const int blocks = 256 * wpc / 4, iters = 400;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, d, regions[r] / 16, iters, out, ticks);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> t(blocks); CK(hipMemcpy(t.data(), ticks, 8 * blocks, hipMemcpyDeviceToHost));
            double m = 0; for (auto v : t) m += v; m /= blocks;
            const double bytes_per_block = (double)iters * 8 * 256 * 16;
            printf("%-36s %2d waves/CU: %.1f B per cycle and CU (%.0f cycles per block), %.2f TB/s over the launch\n", names[r], wpc,
                   bytes_per_block * (wpc / 4) / m, m, bytes_per_block * blocks / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
