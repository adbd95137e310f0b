// Microbenchmark: cost per kernel of a chain of dependent tiny kernels on one stream (eager and hipGraph),
// for several grid sizes.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.cpp -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
__global__ void k_empty() {}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *a, *b; CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24));
    CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(b, 0, 1 << 24));
    const int N = 2000;
    int grids[] = {1, 96, 128, 256, 1024};
    for (int mode = 0; mode < 2; ++mode) {
        for (int g : grids) {
            const int n = g * 256;
            // eager
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipStreamSynchronize(s));
                double t0 = now();
                for (int i = 0; i < N; ++i) {
                    if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s);
                    else hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
                }
                double t1 = now();
                CK(hipStreamSynchronize(s));
                double t2 = now();
                if (rep) printf("%s grid %4d eager : %.2f us/kernel total, host issue %.2f us/kernel\n", mode ? "touch" : "empty", g, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
            }
            // graph of 250 nodes
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 250; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s);
                else hipLaunchKernelGGL(k_touch, dim3(g), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
            }
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(exec, s));
            CK(hipStreamSynchronize(s));
            double t2 = now();
            printf("%s grid %4d graph : %.2f us/kernel\n", mode ? "touch" : "empty", g, (t2 - t0) / (8 * 250) * 1e6);
            hipGraphExecDestroy(exec); hipGraphDestroy(graph);
        }
    }
    return 0;
}
