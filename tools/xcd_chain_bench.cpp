// Probe + bench of the XCD-local recurrence (csrc/lstm_xcd.hip), all in one binary so that one GPU call answers:
//   1  the register layout and the cbsz / abid broadcast semantics of v_mfma_f32_4x4x1_16B_f32 (checked element by element)
//   2  its issue rate with 1 / 2 / 4 accumulator chains (cycles per instruction, one wave per SIMD)
//   3  block -> XCD placement (XCC id of block b) and the one-way hand-off latency between two blocks of the SAME XCD
//      vs two blocks of DIFFERENT XCDs, for write-through (sc1) and plain (L2-resident) 16-byte stores, sc1 loads
//   4  k_lstm_fwd_xcd / k_lstm_bwd_xcd against a double-precision CPU recurrence (B = 45 and B = 100), and their time
//      per step at T = 128 next to the column-split persistent kernels of lstm_step.hip; VARIANT=<XCD_* bits> selects the
//      variant the correctness / timing / profile legs run (the variant sweep line always covers 16 / 32 / 48)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifew-shot-music-generation_amd/csrc -Iinclude -c tools/xcd_chain_bench.cpp -o /tmp/xcb.o
//        hipcc --offload-arch=gfx950 /tmp/xcb.o few-shot-music-generation_amd/build/exp/lstm_xcd.o few-shot-music-generation_amd/build/exp/lstm_step.o few-shot-music-generation_amd/build/exp/gemm.o -o tools/xcd_chain_bench.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <chrono>
#include "fsmg_kernels.h"

using namespace fsmg;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- 1: layout
template <int ABID>
__global__ void k_layout(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + lane] = c[i];
}
__global__ void k_layout_nobcast(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + lane] = c[i];
}

static int check_layout() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + l; }
    float *a, *b, *d;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
    int bad = 0;
    hipLaunchKernelGGL(k_layout_nobcast, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {      // assumed: D[i][j] of block c in register i, lane 4c + j = A[4c + i] * B[4c + j]
        const float want = ha[4 * (l / 4) + i] * hb[l];
        if (hd[i * 64 + l] != want) { if (bad < 4) printf("  layout(no broadcast) reg %d lane %d: got %g want %g\n", i, l, hd[i * 64 + l], want); ++bad; }
    }
    printf("[1] 4x4x1_16B D layout (reg = row, lane = 4*block + col): %s\n", bad ? "MISMATCH" : "ok");
    int bad2 = 0;
    hipLaunchKernelGGL(k_layout<5>, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
        const float want = ha[4 * 5 + i] * hb[l];
        if (hd[i * 64 + l] != want) { if (bad2 < 4) printf("  cbsz=4 abid=5 reg %d lane %d: got %g want %g\n", i, l, hd[i * 64 + l], want); ++bad2; }
    }
    hipLaunchKernelGGL(k_layout<15>, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) if (hd[i * 64 + l] != ha[60 + i] * hb[l]) ++bad2;
    printf("[1] cbsz = 4, abid = b broadcasts A of block b to all 16 blocks: %s\n", bad2 ? "MISMATCH" : "ok");
    hipFree(a); hipFree(b); hipFree(d);
    return bad + bad2;
}

// ---------------------------------------------------------------- 2: issue rate
template <int CH>
__global__ __launch_bounds__(256, 1) void k_rate(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64 / CH; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 3, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
static void rate_one(float* out, unsigned long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL((k_rate<CH>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_rate<CH>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    const double n = 64.0 * iters;
    printf("[2] %d chain(s): %.2f s_memtime ticks per MFMA 4x4x1 (mean over blocks), kernel %.1f us -> %.1f TFLOP/s chip-wide\n", CH,
           mean / n, ms * 1e3, 512.0 * n * 1024 / (ms * 1e-3) / 1e12);
}

// ---------------------------------------------------------------- 3: placement + hand-off latency
__global__ void k_xcc(int* out) {
    if (threadIdx.x == 0) { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); out[blockIdx.x] = v & 0xF; }
}

__device__ __forceinline__ f32x4 ld_sc1(const f32x4* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sc1(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_plain(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }

// blocks `pa` and `pb` play ping-pong over two 1 KiB mailboxes (one wave each, 64 x 16 B), every other block exits.
// rounds of: A writes seq to box0; B polls box0 for seq, writes seq to box1; A polls box1.  result: ticks per round trip.
template <bool PLAIN>
__global__ void k_pingpong(f32x4* box, int pa, int pb, int rounds, unsigned long long* ticks, int* fails, int spin_limit) {
    const int b = blockIdx.x;
    if (b != pa && b != pb) return;
    const int lane = threadIdx.x;
    f32x4* box0 = box + lane;
    f32x4* box1 = box + 64 + lane;
    int failed = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 1; r <= rounds && !failed; ++r) {
        const float tag = (float)r;
        const f32x4 v = {tag, tag, tag, tag};
        if (b == pa) {
            if (PLAIN) st_plain(box0, v); else st_sc1(box0, v);
            int spins = 0;
            for (;;) { const f32x4 g = ld_sc1(box1); if (__all(g[0] == tag && g[3] == tag)) break; if (++spins > spin_limit) { failed = 1; break; } }
        } else {
            int spins = 0;
            for (;;) { const f32x4 g = ld_sc1(box0); if (__all(g[0] == tag && g[3] == tag)) break; if (++spins > spin_limit) { failed = 1; break; } }
            if (PLAIN) st_plain(box1, v); else st_sc1(box1, v);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { if (b == pa) ticks[0] = t1 - t0; if (failed) atomicAdd(fails, 1); }
}

static void handoff_probe() {
    int* d_x; CK(hipMalloc(&d_x, 4 * 256));
    hipLaunchKernelGGL(k_xcc, dim3(256), dim3(64), 0, 0, d_x);
    std::vector<int> x(256); CK(hipMemcpy(x.data(), d_x, 4 * 256, hipMemcpyDeviceToHost));
    int mism = 0; for (int b = 0; b < 256; ++b) mism += (x[b] != b % 8);
    printf("[3] XCC id of block b == b %% 8 for %d of 256 blocks (first 16:", 256 - mism);
    for (int b = 0; b < 16; ++b) printf(" %d", x[b]);
    printf(")\n");
    f32x4* box; unsigned long long* ticks; int* fails;
    CK(hipMalloc(&box, 4096)); CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&fails, 4));
    const int rounds = 2000;
    // pairs: (0, 8) same XCD if b % 8 holds; (0, 1) different XCDs
    int same_b = -1, diff_b = -1;
    for (int b = 1; b < 256 && (same_b < 0 || diff_b < 0); ++b) { if (x[b] == x[0] && same_b < 0) same_b = b; if (x[b] != x[0] && diff_b < 0) diff_b = b; }
    struct Case { const char* name; bool plain; int pb; } cases[] = {
        {"same XCD , sc1 (write-through) stores", false, same_b}, {"cross XCD, sc1 (write-through) stores", false, diff_b},
        {"same XCD , plain (L2-resident) stores ", true, same_b}, {"cross XCD, plain (L2-resident) stores ", true, diff_b}};
    for (auto& c : cases) {
        if (c.pb < 0) continue;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(box, 0, 4096)); CK(hipMemset(fails, 0, 4)); CK(hipMemset(ticks, 0, 8));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (c.plain) hipLaunchKernelGGL((k_pingpong<true>), dim3(256), dim3(64), 0, 0, box, 0, c.pb, rounds, ticks, fails, 200000);
            else hipLaunchKernelGGL((k_pingpong<false>), dim3(256), dim3(64), 0, 0, box, 0, c.pb, rounds, ticks, fails, 200000);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t; int f;
            CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, fails, 4, hipMemcpyDeviceToHost));
            if (rep) printf("[3] %s blocks (0,%3d): one-way %.3f us (%.0f ticks), %s\n", c.name, c.pb, ms * 1e3 / rounds / 2, (double)t / rounds / 2,
                            f ? "TIMED OUT (stale: not a valid hand-off for this placement)" : "ok");
        }
    }
    hipFree(d_x); hipFree(box); hipFree(ticks); hipFree(fails);
}

// ---------------------------------------------------------------- 4: the kernels
static inline int pcol(int u, int g) { return 16 * (u >> 2) + 4 * g + (u & 3); }
static inline double sigm(double x) { return 1.0 / (1.0 + std::exp(-x)); }

struct Problem {
    int B, T, H = 512, G4 = 2048;
    std::vector<float> Kh, Zin, dH;                 // Kh [H][G4] packed; Zin [T][B][G4]; dH [T][B][H]
    std::vector<double> hs, cs, gates, dz, dhrec;   // CPU results
};

static void cpu_forward(Problem& p) {
    const int B = p.B, T = p.T, H = p.H, G4 = p.G4;
    p.hs.assign((size_t)(T + 1) * B * H, 0.0); p.cs.assign((size_t)(T + 1) * B * H, 0.0); p.gates.assign((size_t)T * B * G4, 0.0);
    std::vector<double> z(G4);
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < B; ++b) {
            const double* h = &p.hs[((size_t)t * B + b) * H];
            for (int c = 0; c < G4; ++c) z[c] = p.Zin[((size_t)t * B + b) * G4 + c];
            for (int k = 0; k < H; ++k) { const double hk = h[k]; if (hk == 0.0) continue; const float* kr = &p.Kh[(size_t)k * G4]; for (int c = 0; c < G4; ++c) z[c] += hk * kr[c]; }
            for (int u = 0; u < H; ++u) {
                const double si = sigm(z[pcol(u, 0)]), tj = std::tanh(z[pcol(u, 1)]), sf = sigm(z[pcol(u, 2)] + 1.0), so = sigm(z[pcol(u, 3)]);
                const double c = p.cs[((size_t)t * B + b) * H + u] * sf + si * tj;
                p.cs[((size_t)(t + 1) * B + b) * H + u] = c;
                p.hs[((size_t)(t + 1) * B + b) * H + u] = std::tanh(c) * so;
                double* g = &p.gates[((size_t)t * B + b) * G4];
                g[pcol(u, 0)] = si; g[pcol(u, 1)] = tj; g[pcol(u, 2)] = sf; g[pcol(u, 3)] = so;
            }
        }
}

static void cpu_backward(Problem& p) {       // from the CPU forward's gates / cs and dH: dz [T][B][G4]
    const int B = p.B, T = p.T, H = p.H, G4 = p.G4;
    p.dz.assign((size_t)T * B * G4, 0.0);
    std::vector<double> dc((size_t)B * H, 0.0), dhr((size_t)B * H, 0.0);
    for (int t = T - 1; t >= 0; --t) {
        for (int b = 0; b < B; ++b)
            for (int u = 0; u < H; ++u) {
                const double* g = &p.gates[((size_t)t * B + b) * G4];
                const double si = g[pcol(u, 0)], tj = g[pcol(u, 1)], sf = g[pcol(u, 2)], so = g[pcol(u, 3)];
                const double ct = p.cs[((size_t)(t + 1) * B + b) * H + u], cp = p.cs[((size_t)t * B + b) * H + u];
                const double dh = p.dH[((size_t)t * B + b) * H + u] + dhr[(size_t)b * H + u];
                const double tc = std::tanh(ct);
                const double d = dc[(size_t)b * H + u] + dh * so * (1 - tc * tc);
                double* o = &p.dz[((size_t)t * B + b) * G4];
                o[pcol(u, 0)] = d * tj * si * (1 - si); o[pcol(u, 1)] = d * si * (1 - tj * tj);
                o[pcol(u, 2)] = d * cp * sf * (1 - sf); o[pcol(u, 3)] = dh * tc * so * (1 - so);
                dc[(size_t)b * H + u] = d * sf;
            }
        for (int b = 0; b < B; ++b) {
            const double* o = &p.dz[((size_t)t * B + b) * G4];
            for (int u = 0; u < H; ++u) { double s = 0; const float* kr = &p.Kh[(size_t)u * G4]; for (int c = 0; c < G4; ++c) s += o[c] * kr[c]; dhr[(size_t)b * H + u] = s; }
        }
    }
}

static double relmax(const float* got, const double* want, size_t n) {
    double e = 0, m = 1e-30;
    for (size_t i = 0; i < n; ++i) { e = std::max(e, std::fabs((double)got[i] - want[i])); m = std::max(m, std::fabs(want[i])); }
    return e / m;
}

struct Dev {
    float *Kh, *KhXf, *KhXb, *KhF, *HX, *inboxX, *Z, *Zsave, *Cs, *Hs, *dC, *dH, *HF, *inbox; int *tickets, *err;
};

static int g_variant = 0;
static bool g_bx3 = false;
static int g_rpx = 0;          // RPX=n: rows per XCD (packs the batch on the first ceil(B / n) XCDs); buffers sized for 128 rows then
static int run_case(int B, int Tcheck, int Ttime, int H) {
    const int G4 = 4 * H;
    Problem p; p.B = B; p.T = Tcheck; p.H = H; p.G4 = G4;
    std::mt19937 rng(1234 + B);
    const float kl = H == 512 ? 0.048f : 0.034f;        // Glorot-uniform limit of a [H, 4H] block
    std::uniform_real_distribution<float> uk(-kl, kl), uz(-1.5f, 1.5f), ud(-1e-3f, 1e-3f);
    p.Kh.resize((size_t)H * G4); for (auto& v : p.Kh) v = uk(rng);
    const int Tmax = std::max(Tcheck, Ttime);
    std::vector<float> Zin((size_t)Tmax * B * G4), dH((size_t)Tmax * B * H);
    for (auto& v : Zin) v = uz(rng);
    for (auto& v : dH) v = ud(rng);
    p.Zin.assign(Zin.begin(), Zin.begin() + (size_t)Tcheck * B * G4);
    p.dH.assign(dH.begin(), dH.begin() + (size_t)Tcheck * B * H);
    cpu_forward(p); cpu_backward(p);

    Dev d;
    const size_t Bp16 = (size_t)(B + 15) / 16 * 16;
    CK(hipMalloc(&d.Kh, 4ull * H * G4)); CK(hipMalloc(&d.KhXf, 4ull * lstm_xcd_weight_floats(H, g_bx3))); CK(hipMalloc(&d.KhXb, 4ull * lstm_xcd_weight_floats(H, g_bx3))); CK(hipMalloc(&d.KhF, 8ull * H * G4));
    CK(hipMalloc(&d.HX, 4ull * lstm_xcd_hx_floats((g_rpx && H == 512) ? 128 : B, Tmax, H, g_bx3))); CK(hipMalloc(&d.inboxX, 4ull * lstm_xcd_inbox_floats((g_rpx && H == 512) ? 128 : B, H)));
    CK(hipMalloc(&d.Z, 4ull * Tmax * B * G4)); CK(hipMalloc(&d.Zsave, 4ull * Tmax * B * G4));
    CK(hipMalloc(&d.Cs, 4ull * (Tmax + 1) * B * H)); CK(hipMalloc(&d.Hs, 4ull * (Tmax + 1) * B * H));
    CK(hipMalloc(&d.dC, 4ull * B * H)); CK(hipMalloc(&d.dH, 4ull * Tmax * B * H));
    CK(hipMalloc(&d.HF, 4ull * (Tmax + 1) * Bp16 * H));
    const bool old_rs = lstm_bwd_rs_supported(B, H);
    CK(hipMalloc(&d.inbox, 4ull * std::max<long long>(old_rs ? lstm_bwd_rs_inbox_floats(B, H) : 64, 64)));
    CK(hipMalloc(&d.tickets, 64 * 4)); CK(hipMalloc(&d.err, 256));
    CK(hipMemcpy(d.Kh, p.Kh.data(), 4ull * H * G4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.dH, dH.data(), 4ull * Tmax * B * H, hipMemcpyHostToDevice));
    CK(hipMemset(d.err, 0, 256));
    hipStream_t s; CK(hipStreamCreate(&s));
    CK(launch_repack_kh_xcd(s, d.Kh, d.KhXf, d.KhXb, H, g_bx3));
    CK(launch_repack_kh(s, d.Kh, d.KhF, d.KhF + (size_t)H * G4, H));
    int rc = 0;

    auto fwd_xcd = [&](int T, int nchunk) {
        CK(hipMemcpyAsync(d.Z, Zin.data(), 4ull * T * B * G4, hipMemcpyHostToDevice, s));
        CK(hipMemsetAsync(d.Cs, 0, 4ull * B * H, s)); CK(hipMemsetAsync(d.Hs, 0, 4ull * B * H, s));
        const size_t step_f = (size_t)lstm_xcd_hx_floats((g_rpx && H == 512) ? 128 : B, 0, H, g_bx3);
        CK(hipMemsetAsync(d.HX, 0, 4 * step_f, s));
        CK(hipMemsetAsync(d.HX + step_f, 0xFF, 4 * step_f * T, s));
        CK(hipMemsetAsync(d.tickets, 0, 64 * 4, s));
        for (int c = 0; c < nchunk; ++c) {
            LstmFwdXcdArgs a{};
            a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets + 8 * c; a.err_flag = d.err;
            a.B = B; a.T = T; a.t0 = (int)((long long)c * T / nchunk); a.t1 = (int)((long long)(c + 1) * T / nchunk); a.spin_limit = 1 << 18; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
            CK(launch_lstm_fwd_xcd(s, a));
        }
    };
    auto bwd_xcd = [&](int T, int nchunk) {
        CK(hipMemsetAsync(d.dC, 0, 4ull * B * H, s));
        CK(hipMemsetAsync(d.inboxX, 0xFF, 4ull * lstm_xcd_inbox_floats((g_rpx && H == 512) ? 128 : B, H), s));
        CK(hipMemsetAsync(d.tickets, 0, 64 * 4, s));
        for (int c = nchunk - 1; c >= 0; --c) {
            LstmBwdXcdArgs a{};
            a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets + 8 * c; a.err_flag = d.err;
            a.B = B; a.T = T; a.t0 = (int)((long long)c * T / nchunk); a.t1 = (int)((long long)(c + 1) * T / nchunk); a.spin_limit = 1 << 18; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
            CK(launch_lstm_bwd_xcd(s, a));
        }
    };
    auto read_err = [&]() { int e; CK(hipStreamSynchronize(s)); CK(hipMemcpy(&e, d.err, 4, hipMemcpyDeviceToHost)); return e; };

    // ---- correctness (2 chunks so that the cross-launch hand-off is exercised too)
    {
        const int T = Tcheck;
        fwd_xcd(T, 2);
        int e = read_err();
        std::vector<float> hs((size_t)(T + 1) * B * H), cs((size_t)(T + 1) * B * H), g((size_t)T * B * G4);
        CK(hipMemcpy(hs.data(), d.Hs, hs.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cs.data(), d.Cs, cs.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(g.data(), d.Z, g.size() * 4, hipMemcpyDeviceToHost));
        const double eh = relmax(hs.data(), p.hs.data(), hs.size()), ec = relmax(cs.data(), p.cs.data(), cs.size()), eg = relmax(g.data(), p.gates.data(), g.size());
        const bool ok = e == 0 && eh < 2e-5 && ec < 2e-5 && eg < 2e-5;
        printf("[4] B=%d H=%d fwd_xcd  vs CPU fp64 (T=%d, 2 launches): err_flag %d, h %.2e c %.2e gates %.2e  %s\n", B, H, T, e, eh, ec, eg, ok ? "ok" : "FAIL");
        rc += !ok;
        // backward on the GPU's own forward state
        bwd_xcd(T, 2);
        e = read_err();
        std::vector<float> dz((size_t)T * B * G4);
        CK(hipMemcpy(dz.data(), d.Z, dz.size() * 4, hipMemcpyDeviceToHost));
        const double ed = relmax(dz.data(), p.dz.data(), dz.size());
        const bool okb = e == 0 && ed < 1e-4;
        printf("[4] B=%d H=%d bwd_xcd  vs CPU fp64 (T=%d, 2 launches): err_flag %d, dz %.2e  %s\n", B, H, T, e, ed, okb ? "ok" : "FAIL");
        rc += !okb;
        CK(hipMemset(d.err, 0, 4));
    }
    // ---- timing at T = Ttime, 1 and 4 launches per chain, new vs old kernels
    {
        const int T = Ttime;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double mflop = 2.0 * B * H * G4 / 1e6 * 1e3;   // so that mflop * T / ms / 1e6 = TFLOP/s
        for (int nchunk : {1, 4}) {
            float best_f = 1e9f, best_b = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                fwd_xcd(T, 0);            // resets only
                CK(hipEventRecord(e0, s));
                for (int c = 0; c < nchunk; ++c) {
                    LstmFwdXcdArgs a{};
                    a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets + 8 * c; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = (int)((long long)c * T / nchunk); a.t1 = (int)((long long)(c + 1) * T / nchunk); a.spin_limit = 1 << 18; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    CK(launch_lstm_fwd_xcd(s, a));
                }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_f = std::min(best_f, ms);
                bwd_xcd(T, 0);
                CK(hipEventRecord(e0, s));
                for (int c = nchunk - 1; c >= 0; --c) {
                    LstmBwdXcdArgs a{};
                    a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets + 8 * c; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = (int)((long long)c * T / nchunk); a.t1 = (int)((long long)(c + 1) * T / nchunk); a.spin_limit = 1 << 18; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    CK(launch_lstm_bwd_xcd(s, a));
                }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                CK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
            }
            const int e = read_err();
            printf("[4] B=%d H=%d xcd-local, %d launch(es) per chain, T=%d: fwd %.3f ms = %.2f us/step = %.1f TFLOP/s (%.1f%% of 157.3) | bwd %.3f ms = %.2f us/step = %.1f TFLOP/s (%.1f%%)  err_flag %d\n",
                   B, H, nchunk, T, best_f, best_f * 1e3 / T, mflop * T / best_f / 1e6, mflop * T / best_f / 1e6 / 157.3 * 100, best_b, best_b * 1e3 / T,
                   mflop * T / best_b / 1e6, mflop * T / best_b / 1e6 / 157.3 * 100, e);
            CK(hipMemset(d.err, 0, 4));
        }
        if (getenv("CONC") && atoi(getenv("CONC")) && g_bx3 && g_rpx && H == 512 && B == 45) {
            // CONC=1 (with BX3=1 RPX=15): the backward chain on XCDs 0-2 BESIDE cfg-B's dW GEMM (256 x 256-tile queue kernel,
            // K split 6) confined to XCDs 3-7 on a second stream -- what the packed schedule of DESIGN.md section 4 would run
            const int M = 512, N = 10016, K = 5760, S = 6;
            float *A, *Bm, *slabs; int* ctl;
            CK(hipMalloc(&A, 4ull * K * M)); CK(hipMalloc(&Bm, 4ull * K * N)); CK(hipMalloc(&slabs, 4ull * std::max<size_t>((size_t)S * M * N, (size_t)5760 * N) /* dW's slabs, or the projection's output */)); CK(hipMalloc(&ctl, 4 * 65536));
            CK(hipMemset(A, 0, 4ull * K * M)); CK(hipMemset(Bm, 0, 4ull * K * N));
            hipStream_t s2; CK(hipStreamCreate(&s2));
            hipEvent_t g0, g1, c0, c1; CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
            GemmArgs g{};
            g.A = A; g.lda = M; g.B = Bm; g.ldb = N; g.C = slabs; g.ldc = N; g.M = M; g.N = N; g.K = K; g.ksplit = S; g.c_slab = (long long)M * N; g.bx3 = 3;
            g.work = ctl; g.stop = ctl + 2; g.claim = ctl + 4; g.work_limit = 1 << 30;
            auto chain = [&]() {
                LstmBwdXcdArgs a{};
                a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets; a.err_flag = d.err;
                a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.variant = 32; a.Hp = H; a.bx3 = 1; a.rpx = g_rpx;
                CK(launch_lstm_bwd_xcd(s, a));
            };
            for (int mode = 0; mode < 3; ++mode) {      // 0: chain alone, 1: confined GEMM alone, 2: both
                float best_c = 1e9f, best_g = 1e9f, best_w = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    fwd_xcd(T, 1); bwd_xcd(T, 0);      // fresh forward state and resets
                    CK(hipMemsetAsync(ctl, 0, 4 * 65536, s)); CK(hipStreamSynchronize(s));
                    const auto w0 = std::chrono::steady_clock::now();
                    if (mode != 1) { CK(hipEventRecord(c0, s)); chain(); CK(hipEventRecord(c1, s)); }
                    if (mode != 0) {
                        GemmArgs r1 = g; r1.xcd_first = 3;
                        CK(hipEventRecord(g0, s2)); CK(launch_gemm(s2, OP_XC, OP_XC, r1, 0)); CK(hipEventRecord(g1, s2));
                    }
                    CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
                    const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
                    float ms;
                    if (mode != 1) { CK(hipEventElapsedTime(&ms, c0, c1)); best_c = std::min(best_c, ms); }
                    if (mode != 0) { CK(hipEventElapsedTime(&ms, g0, g1)); best_g = std::min(best_g, ms); }
                    best_w = std::min(best_w, wall);
                }
                int claimed = 0; { std::vector<int> hc(65536); CK(hipMemcpy(hc.data(), ctl, 4 * 65536, hipMemcpyDeviceToHost)); for (int i = 4; i < 4 + 80 * S; ++i) claimed += hc[i] != 0; }
                printf("[5] CONC mode %d (%s): chain %.3f ms (%.2f us/step)  confined dW (K split %d, %d of %d items) %.3f ms  wall %.3f ms  err_flag %d\n", mode,
                       mode == 0 ? "chain alone" : mode == 1 ? "GEMM alone on XCDs 3-7" : "both", mode != 1 ? best_c : 0.f, mode != 1 ? best_c * 1e3 / T : 0.f, S, claimed, 80 * S,
                       mode != 0 ? best_g : 0.f, best_w, read_err());
                CK(hipMemset(d.err, 0, 4));
            }
            // the forward side: the packed forward chain beside the projection (M = T B rows, K = 512, one slab) confined to the
            // free XCDs -- every tile on offer at once, i.e. interference and confined rate, not the dependency on h_t
            {
                GemmArgs p2{};
                p2.A = A; p2.lda = 512; p2.B = Bm; p2.ldb = N; p2.C = slabs; p2.ldc = N; p2.M = 5760; p2.N = N; p2.K = 512; p2.ksplit = 1; p2.bx3 = 3; p2.nt_store = 1;
                p2.work = ctl; p2.stop = ctl + 2; p2.claim = ctl + 4; p2.work_limit = 1 << 30;
                for (int mode = 0; mode < 3; ++mode) {
                    float best_c = 1e9f, best_g = 1e9f, best_w = 1e9f;
                    for (int rep = 0; rep < 4; ++rep) {
                        fwd_xcd(T, 0);                 // resets only
                        CK(hipMemsetAsync(ctl, 0, 4 * 65536, s)); CK(hipStreamSynchronize(s));
                        const auto w0 = std::chrono::steady_clock::now();
                        if (mode != 1) {
                            LstmFwdXcdArgs a{};
                            a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets; a.err_flag = d.err;
                            a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.variant = 32; a.Hp = H; a.bx3 = 1; a.rpx = g_rpx;
                            CK(hipEventRecord(c0, s)); CK(launch_lstm_fwd_xcd(s, a)); CK(hipEventRecord(c1, s));
                        }
                        if (mode != 0) {
                            GemmArgs r1 = p2; r1.xcd_first = 3;
                            CK(hipEventRecord(g0, s2)); CK(launch_gemm(s2, OP_KC, OP_XC, r1, 0)); CK(hipEventRecord(g1, s2));
                        }
                        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
                        const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
                        float ms;
                        if (mode != 1) { CK(hipEventElapsedTime(&ms, c0, c1)); best_c = std::min(best_c, ms); }
                        if (mode != 0) { CK(hipEventElapsedTime(&ms, g0, g1)); best_g = std::min(best_g, ms); }
                        best_w = std::min(best_w, wall);
                    }
                    printf("[5] CONC forward mode %d (%s): chain %.3f ms (%.2f us/step)  confined projection (920 tiles) %.3f ms  wall %.3f ms  err_flag %d\n", mode,
                           mode == 0 ? "chain alone" : mode == 1 ? "GEMM alone on XCDs 3-7" : "both", mode != 1 ? best_c : 0.f, mode != 1 ? best_c * 1e3 / T : 0.f,
                           mode != 0 ? best_g : 0.f, best_w, read_err());
                    CK(hipMemset(d.err, 0, 4));
                }
            }
            hipFree(A); hipFree(Bm); hipFree(slabs); hipFree(ctl);
        }
        {                    // variants (same results): 16 = XCD_DEFER_OUTPUTS, 32 = XCD_NO_POLL_SLEEP
            for (int dbg : {32, 160, 672, 688, 33456, 288, 1312, 2080}) {
                if ((dbg & (64 | 256 | 1024)) && H != 1024) continue;
                if ((dbg & (128 | 512 | 2048)) && H == 512 && !g_bx3) continue;          // (the bf16-split kernels' variants)          // XCD_CHAINS: hidden 1024 only
                float best_f = 1e9f, best_b = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    fwd_xcd(T, 0);
                    CK(hipEventRecord(e0, s));
                    { LstmFwdXcdArgs a{}; a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets; a.err_flag = d.err;
                      a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.variant = dbg; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0; CK(launch_lstm_fwd_xcd(s, a)); }
                    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_f = std::min(best_f, ms);
                    bwd_xcd(T, 0);
                    CK(hipEventRecord(e0, s));
                    { LstmBwdXcdArgs a{}; a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets; a.err_flag = d.err;
                      a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.variant = dbg; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0; CK(launch_lstm_bwd_xcd(s, a)); }
                    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                    CK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
                }
                printf("[4] B=%d H=%d variant %d (%s%s): fwd %.2f us/step | bwd %.2f us/step  err_flag %d\n", B, H, dbg, (dbg & 16) ? "deferred stores " : "", (dbg & 64) ? "row-group chains, no poll sleep" : (dbg & 32) ? "no poll sleep" : "",
                       best_f * 1e3 / T, best_b * 1e3 / T, read_err());
                CK(hipMemset(d.err, 0, 4));
            }
        }
        if (H == 1024 && !g_bx3) {     // chain kernels: how many of the early-issued polls were NOT ready when their phase started
            unsigned long long* prof; CK(hipMalloc(&prof, 8ull * 64));
            unsigned long long hp[16];
            for (int dir = 0; dir < 2; ++dir) {
                CK(hipMemset(prof, 0, 8ull * 64));
                if (dir == 0) {
                    fwd_xcd(T, 0);
                    LstmFwdXcdArgs a{}; a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.prof = prof; a.variant = 96; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    if (launch_lstm_fwd_xcd(s, a) != hipSuccess) continue;
                } else {
                    bwd_xcd(T, 0);
                    LstmBwdXcdArgs a{}; a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.prof = prof; a.variant = 96; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    if (launch_lstm_bwd_xcd(s, a) != hipSuccess) continue;
                }
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(hp, prof, 8 * 8, hipMemcpyDeviceToHost));
                printf("[4] B=%d H=1024 chains %s: early polls checked / not ready per wave: %llu/%llu %llu/%llu %llu/%llu %llu/%llu\n", B, dir ? "bwd" : "fwd",
                       hp[0], hp[4], hp[1], hp[5], hp[2], hp[6], hp[3], hp[7]);
            }
            hipFree(prof);
            CK(hipMemset(d.err, 0, 4));
        }
        if ((B == 45 && H == 512) || (B == 45 && H == 1024 && g_bx3)) {       // phase profile of the instrumented build (RG = 2; hidden 1024 bf16-split: RG = 3)
            unsigned long long* prof; CK(hipMalloc(&prof, 8ull * 256 * 4 * 8));
            std::vector<unsigned long long> hp(256 * 4 * 8);
            for (int dir = 0; dir < 2; ++dir) {
                CK(hipMemset(prof, 0, 8ull * 256 * 4 * 8));
                if (dir == 0) {
                    fwd_xcd(T, 0);
                    LstmFwdXcdArgs a{};
                    a.KhX = d.KhXf; a.HX = d.HX; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.tickets = d.tickets; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.prof = prof; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    CK(hipEventRecord(e0, s)); CK(launch_lstm_fwd_xcd(s, a)); CK(hipEventRecord(e1, s));
                } else {
                    bwd_xcd(T, 0);
                    LstmBwdXcdArgs a{};
                    a.KhXb = d.KhXb; a.inbox = d.inboxX; a.Z = d.Z; a.Cs = d.Cs; a.dc = d.dC; a.dH = d.dH; a.tickets = d.tickets; a.err_flag = d.err;
                    a.B = B; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18; a.prof = prof; a.variant = g_variant; a.Hp = H; a.bx3 = (g_bx3) ? 1 : 0; a.rpx = (H == 512) ? g_rpx : 0;
                    CK(hipEventRecord(e0, s)); CK(launch_lstm_bwd_xcd(s, a)); CK(hipEventRecord(e1, s));
                }
                CK(hipStreamSynchronize(s));
                float ms_prof = 0.0f; CK(hipEventElapsedTime(&ms_prof, e0, e1));
                CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
                const char* names_f0[5] = {"wait h_t", "MFMA", "LDS+barrier", "cell->store", "rest"};
                const char* names_b0[5] = {"wait inbox", "psum+barrier", "cell+dzA+barrier", "LDS read+MFMA", "drain+stores+rest"};
                const char** names_f = names_f0;
                const char** names_b = names_b0;
                if (H == 1024) for (int w = 0; w < 4; ++w) {       // per wave: phases + probe / full poll rounds per step
                    double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int b = 0; b < 256; ++b) for (int i = 0; i < 8; ++i) m[i] += (double)hp[((size_t)b * 4 + w) * 8 + i];
                    printf("[4] %s wave %d ticks per step: %.0f %.0f %.0f %.0f %.0f | probe rounds %.2f full rounds %.2f probe ticks %.0f\n", dir ? "bwd" : "fwd", w,
                           m[0] / 256 / T, m[1] / 256 / T, m[2] / 256 / T, m[3] / 256 / T, m[4] / 256 / T, m[5] / 256 / T, m[6] / 256 / T, m[7] / 256 / T);
                }
                for (int wc = 0; wc < 2; ++wc) {        // cell waves (0,1) vs the others (2,3)
                    double m[5] = {0, 0, 0, 0, 0};
                    for (int b = 0; b < 256; ++b) for (int w = 2 * wc; w < 2 * wc + 2; ++w) for (int i = 0; i < 5; ++i) m[i] += (double)hp[((size_t)b * 4 + w) * 8 + i];
                    printf("[4] %s phase ticks per step, waves %d-%d:", dir ? "bwd" : "fwd", 2 * wc, 2 * wc + 1);
                    double tot = 0;
                    for (int i = 0; i < 5; ++i) { printf("  %s %.0f", dir ? names_b[i] : names_f[i], m[i] / 512 / T); tot += m[i] / 512 / T; }
                    printf("  | total %.0f  (stamped launch: %.3f us per step -> %.3f GHz shader clock)\n", tot, 1e3 * ms_prof / T, tot / (1e3 * ms_prof / T) / 1e3);
                }
            }
            hipFree(prof);
            CK(hipMemset(d.err, 0, 4));
        }
        if (lstm_fwd_chain_supported(B, H)) {
            float best_f = 1e9f, best_b = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemcpyAsync(d.Z, Zin.data(), 4ull * T * B * G4, hipMemcpyHostToDevice, s));
                CK(hipMemsetAsync(d.Cs, 0, 4ull * B * H, s)); CK(hipMemsetAsync(d.Hs, 0, 4ull * B * H, s));
                CK(hipMemsetAsync(d.HF, 0, 4ull * Bp16 * H, s)); CK(hipMemsetAsync(d.HF + Bp16 * H, 0xFF, 4ull * T * Bp16 * H, s));
                LstmFwdChainArgs a{};
                a.KhF = d.KhF; a.HF = d.HF; a.Z = d.Z; a.Cs = d.Cs; a.Hs = d.Hs; a.err_flag = d.err; a.B = B; a.Hp = H; a.T = T; a.t0 = 0; a.t1 = T; a.spin_limit = 1 << 18;
                CK(hipEventRecord(e0, s)); CK(launch_lstm_fwd_chain(s, a)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_f = std::min(best_f, ms);
                if (old_rs) {
                    CK(hipMemsetAsync(d.dC, 0, 4ull * B * H, s)); CK(hipMemsetAsync(d.inbox, 0xFF, 4ull * lstm_bwd_rs_inbox_floats(B, H), s));
                    LstmBwdRsArgs r{};
                    r.KhF = d.KhF + (size_t)H * G4; r.inbox = d.inbox; r.Z = d.Z; r.Cs = d.Cs; r.dc = d.dC; r.dH = d.dH; r.err_flag = d.err; r.B = B; r.Hp = H; r.T = T; r.t0 = 0; r.t1 = T; r.spin_limit = 1 << 18;
                    CK(hipEventRecord(e0, s)); CK(launch_lstm_bwd_rs(s, r)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                    CK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
                }
            }
            printf("[4] B=%d column-split (round 1) kernels, 1 launch, T=%d: fwd %.2f us/step (%.1f%%) | bwd %.2f us/step (%.1f%%)  err_flag %d\n", B, T,
                   best_f * 1e3 / T, mflop * T / best_f / 1e6 / 157.3 * 100, best_b * 1e3 / T, mflop * T / best_b / 1e6 / 157.3 * 100, read_err());
        }
    }
    hipFree(d.Kh); hipFree(d.KhXf); hipFree(d.KhXb); hipFree(d.KhF); hipFree(d.HX); hipFree(d.inboxX); hipFree(d.Z); hipFree(d.Zsave); hipFree(d.Cs);
    hipFree(d.Hs); hipFree(d.dC); hipFree(d.dH); hipFree(d.HF); hipFree(d.inbox); hipFree(d.tickets); hipFree(d.err);
    return rc;
}

int main(int argc, char** argv) {
    int rc = check_layout();
    {
        float* out; unsigned long long* cyc;
        CK(hipMalloc(&out, 4 * 65536)); CK(hipMalloc(&cyc, 8 * 256));
        rate_one<1>(out, cyc); rate_one<2>(out, cyc); rate_one<4>(out, cyc);
        hipFree(out); hipFree(cyc);
    }
    handoff_probe();
    if (rc) { printf("MFMA layout assumption wrong: kernels not run\n"); return 1; }
    const int only_pipe = getenv("PIPE") ? atoi(getenv("PIPE")) : -1;
    g_variant = getenv("VARIANT") ? atoi(getenv("VARIANT")) : 0;
    g_bx3 = getenv("BX3") && atoi(getenv("BX3")) != 0;
    g_rpx = getenv("RPX") ? atoi(getenv("RPX")) : 0;
    if (g_rpx && (!g_bx3 || g_rpx < 13 || g_rpx > 16)) { printf("RPX needs BX3=1 and 13..16 rows per XCD (the buffers are laid out for four row groups then)\n"); return 2; }      // hidden 512: the bf16-split kernels (k_lstm_*_xcd16)
    (void)only_pipe;
    const int only_h = getenv("HID") ? atoi(getenv("HID")) : 0;
    if (only_h != 1024) {
        rc += run_case(45, 6, 128, 512);
        rc += run_case(100, 4, 128, 512);
        rc += run_case(20, 4, 128, 512);
    }
    if (only_h != 512) {             // hidden 1024: one copy of K_h per XCD pair (cfg-C / cfg-E shapes: 45 / 25 / 20 rows, T = 50)
        rc += run_case(45, 5, 50, 1024);
        rc += run_case(25, 4, 50, 1024);
        rc += run_case(20, 4, 50, 1024);
        rc += run_case(64, 3, 50, 1024);
    }
    printf(rc ? "FAILED\n" : "ALL OK\n");
    return rc;
}
