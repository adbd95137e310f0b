// Probe: which physical CUs does a stream created with hipExtStreamCreateWithCUMask run on?
//   hipcc --offload-arch=gfx950 -O3 tools/cu_mask_probe.cpp -o tools/cu_mask_probe.bin
// For a few masks, launches 2048 small blocks on the masked stream and prints the set of (XCC_ID, SE/SH/CU bits of
// HW_ID) the blocks reported, as a count per XCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_where(unsigned* out, int spin) {
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
    if (x == 1.2345f) out[0] = 0;
}

static int probe(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int nb = 4096;
    unsigned* d; CK(hipMalloc(&d, nb * 8));
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * nb);
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < nb; ++b) per_xcc[h[2 * b + 1] & 0xf].insert(h[2 * b] & 0xff00);
    int total = 0;
    printf("%-28s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  -> %d CUs\n", total);
    if (per_xcc.size() && per_xcc.begin()->second.size() <= 12) {
        printf("    xcc%u cu ids:", per_xcc.begin()->first);
        for (unsigned v : per_xcc.begin()->second) printf(" %04x", v);
        printf("\n");
    }
    CK(hipFree(d)); CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    std::vector<uint32_t> full(8, 0xffffffffu);
    if (probe("all 256 bits", full)) return 1;
    std::vector<uint32_t> m(8, 0u);
    m[0] = 0xffffffffu; m[1] = 0xffffffffu;
    if (probe("bits 0..63", m)) return 1;
    for (auto& w : m) w = 0x000000ffu;
    if (probe("low byte of every word", m)) return 1;
    for (auto& w : m) w = 0x11111111u;
    if (probe("every 4th bit", m)) return 1;
    for (auto& w : m) w = 0x03030303u;
    if (probe("bits 0,1 of every byte", m)) return 1;
    for (auto& w : m) w = 0xfcfcfcfcu;
    if (probe("bits 2..7 of every byte", m)) return 1;
    std::vector<uint32_t> one(8, 0u); one[0] = 0xffu;
    if (probe("bits 0..7", one)) return 1;
    return 0;
}
