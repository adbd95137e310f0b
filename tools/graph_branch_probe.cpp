// Do the branches of a captured hipGraph run side by side on this runtime, and what does a graph boundary cost?
//   hipcc --offload-arch=gfx950 -O3 tools/graph_branch_probe.cpp -o tools/graph_branch_probe.bin
// [1] fork / join of two spinning one-block kernels (300 us each): captured graph vs the same launches eager on two
//     streams vs a hand-built graph with two root nodes -- 300 us = side by side, 600 = one after the other;
// [2] the same with 96-block "resident" kernels (512 threads, a large static LDS block: one block per CU, the shape of the
//     chain / queue-GEMM pair of the overlapped step);
// [3] back-to-back launches of a graph holding one tiny kernel vs the same kernel eager: device time per launch;
// [4] host time to issue an eager launch with a 400-byte argument struct (the GemmArgs of the step).
// Environment (read by the HIP runtime at start-up; run once per setting): DEBUG_HIP_FORCE_GRAPH_QUEUES,
// DEBUG_CLR_GRAPH_PACKET_CAPTURE.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(long long ticks, unsigned long long* out) {          // s_memtime runs at 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = t0;
}
__global__ __launch_bounds__(512) void k_spin_fat(long long ticks, unsigned long long* out) {
    __shared__ float hog[24 * 1024];                                       // 96 KiB: one block per CU
    hog[threadIdx.x] = (float)ticks;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = t0 + (unsigned long long)hog[1];
}
struct Big { char pad[400]; int* p; };
__global__ void k_tiny(Big b) { if (threadIdx.x == 0 && b.p) b.p[0] = 1; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const char* e1 = std::getenv("DEBUG_HIP_FORCE_GRAPH_QUEUES"); const char* e2 = std::getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
    printf("DEBUG_HIP_FORCE_GRAPH_QUEUES=%s DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s\n", e1 ? e1 : "(unset)", e2 ? e2 : "(unset)");
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, join, a, b; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned long long* d; CK(hipMalloc(&d, 8 * 4096)); int* di; CK(hipMalloc(&di, 64));
    const long long ticks = 30000;                                          // 300 us
    for (int fat = 0; fat < 2; ++fat) {
        auto pair = [&](hipStream_t sa, hipStream_t sb) {
            if (fat) { hipLaunchKernelGGL(k_spin_fat, dim3(96), dim3(512), 0, sa, ticks, d); hipLaunchKernelGGL(k_spin_fat, dim3(96), dim3(512), 0, sb, ticks, d + 1024); }
            else { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sa, ticks, d); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sb, ticks, d + 1024); }
        };
        // eager on two streams
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(a, s));
            CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
            pair(s, s2);
            CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("[%d] %s pair eager two streams : %.1f us\n", 1 + fat, fat ? "96-block" : "1-block", ms * 1e3);
        }
        // captured fork / join
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
        pair(s, s2);
        CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ex, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("[%d] %s pair captured graph    : %.1f us\n", 1 + fat, fat ? "96-block" : "1-block", ms * 1e3);
        }
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
        // hand-built graph: two root kernel nodes, no edges
        CK(hipGraphCreate(&g, 0));
        hipGraphNode_t n1, n2;
        long long tk = ticks; unsigned long long* p1 = d; unsigned long long* p2 = d + 1024;
        void* args1[] = {&tk, &p1}; void* args2[] = {&tk, &p2};
        hipKernelNodeParams kp{};
        kp.func = fat ? (void*)k_spin_fat : (void*)k_spin; kp.gridDim = fat ? dim3(96) : dim3(1); kp.blockDim = fat ? dim3(512) : dim3(64);
        kp.sharedMemBytes = 0; kp.kernelParams = args1; kp.extra = nullptr;
        CK(hipGraphAddKernelNode(&n1, g, nullptr, 0, &kp));
        kp.kernelParams = args2;
        CK(hipGraphAddKernelNode(&n2, g, nullptr, 0, &kp));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ex, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("[%d] %s pair hand-built graph  : %.1f us\n", 1 + fat, fat ? "96-block" : "1-block", ms * 1e3);
        }
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
    // [3] graph boundary: N launches of a one-kernel graph vs N eager launches (device time per launch)
    {
        Big bg{}; bg.p = di;
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, s, bg);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const int N = 400;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ex, s));
            double t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
            if (rep) printf("[3] one-kernel graph, back to back : %.2f us per launch (host issue %.2f)\n", (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, s, bg);
            double t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
            if (rep) printf("[4] the same kernel eager (400-byte arguments): %.2f us per launch (host issue %.2f)\n", (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
        // graph, eager kernel, graph, eager kernel ... : the boundary of a mixed schedule
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int i = 0; i < N; ++i) { CK(hipGraphLaunch(ex, s)); hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, s, bg); }
            double t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
            if (rep) printf("[3] graph + eager kernel alternating : %.2f us per pair (host issue %.2f)\n", (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
    return 0;
}
