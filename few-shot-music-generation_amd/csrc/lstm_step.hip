// Fused LSTM cell steps for gfx950: recurrent contraction on v_mfma_f32_16x16x4_f32 with the
// gate nonlinearities / Hadamard products (forward) and the gate-gradient arithmetic
// (backward) fused behind it.  One launch per time step; the launches of a sequence are
// replayed as a hipGraph by the caller.  Cell semantics: BasicLSTMCell(forget_bias=1),
// gate order i, j, f, o (reference src/models/lstm_baseline.py:44-55).
//
// Gate columns are "packed": column pc = 16*(u/4) + 4*gate + (u%4) holds gate `gate` of
// hidden unit u, so the 16 columns of one MFMA N-tile are the four gates of four units and
// the cell update for those units is local to the block.
//
// K-order trick: an MFMA 16x16x4 step consumes k = 4 values, lane slot q = lane>>4 supplying
// A[m][k_q] and B[k_q][n].  Any bijection slot->k is valid as long as A and B agree, so within
// a group of 16 k's slot q takes k = 16g + 4q + s at sub-step s: every lane then reads its
// four k's as ONE 16-byte load from the row-major activations instead of four strided dwords.
#include "fsmg_kernels.h"

namespace fsmg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------- forward
// grid (4Hp/16, ceil(B/48)); 256 threads = 4 waves, wave w owns a quarter of the K = Hp range.
constexpr int FWD_ROWS = 48;
__global__ __launch_bounds__(256) void k_lstm_fwd_step(const LstmFwdArgs a) {
    __shared__ float red[4][FWD_ROWS][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int nb = blockIdx.x;           // unit block: units 4nb..4nb+3, packed cols 16nb..16nb+15
    const int m0 = blockIdx.y * FWD_ROWS;
    const int Hp = a.Hp, G4 = 4 * a.Hp;

    const int ngroups = Hp >> 4;
    const int g_beg = (wave * ngroups) >> 2, g_end = ((wave + 1) * ngroups) >> 2;

    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* hrow[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = m0 + 16 * i + l15;
        hrow[i] = (row < a.B) ? a.h_prev + (long long)row * Hp + 4 * q : nullptr;
    }
    const float* kcol = a.Kh + (long long)(4 * q) * G4 + 16 * nb + l15;

    for (int g = g_beg; g < g_end; ++g) {
        float4 av[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            av[i] = hrow[i] ? *reinterpret_cast<const float4*>(hrow[i] + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* kp = kcol + (long long)(16 * g) * G4;
        const float b0 = kp[0], b1 = kp[G4], b2 = kp[2 * (long long)G4], b3 = kp[3 * (long long)G4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, b0, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, b1, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, b2, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, b3, acc[i], 0, 0, 0);
        }
    }
    // C/D layout 16x16: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][16 * i + 4 * q + r][l15] = acc[i][r];
    __syncthreads();

    if (tid < FWD_ROWS * 4) {
        const int row = tid >> 2, uu = tid & 3;
        const int b = m0 + row;
        if (b < a.B) {
            float zg[4];
            float* zp = a.z + (long long)b * G4 + 16 * nb + uu;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int c = 4 * gi + uu;
                zg[gi] = zp[4 * gi] + ((red[0][row][c] + red[1][row][c]) + (red[2][row][c] + red[3][row][c]));
            }
            const int u = 4 * nb + uu;
            const float si = sigmoidf_(zg[0]);
            const float tj = tanhf(zg[1]);
            const float sf = sigmoidf_(zg[2] + 1.0f);          // forget_bias = 1 added at run time
            const float so = sigmoidf_(zg[3]);
            const float cp = a.c_prev[(long long)b * Hp + u];
            const float cn = cp * sf + si * tj;
            a.c_next[(long long)b * Hp + u] = cn;
            a.h_next[(long long)b * Hp + u] = tanhf(cn) * so;
            zp[0] = si; zp[4] = tj; zp[8] = sf; zp[12] = so;  // activated gates kept for BPTT
        }
    }
}

// ---------------------------------------------------------------- backward
// grid (Hp/16, ceil(B/16)); 512 threads = 8 waves splitting K = 4Hp (packed gate columns).
// dh_rec[b][u] = sum_pc dz_next[b][pc] * Kh[u][pc]; then the gate gradients of step t for the
// block's 16 rows x 16 units.
__global__ __launch_bounds__(512) void k_lstm_bwd_step(const LstmBwdArgs a) {
    __shared__ float red[8][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int u0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int Hp = a.Hp, G4 = 4 * a.Hp;

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.dz_next != nullptr) {
        const int ngroups = G4 >> 4;
        const int g_beg = (wave * ngroups) >> 3, g_end = ((wave + 1) * ngroups) >> 3;
        const int row = m0 + l15;
        const float* ap = (row < a.B) ? a.dz_next + (long long)row * G4 + 4 * q : nullptr;
        const float* bp = a.Kh + (long long)(u0 + l15) * G4 + 4 * q;
        for (int g = g_beg; g < g_end; ++g) {
            const float4 av = ap ? *reinterpret_cast<const float4*>(ap + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bv = *reinterpret_cast<const float4*>(bp + 16 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * q + r][l15] = acc[r];
    __syncthreads();

    if (tid < 256) {
        const int row = tid >> 4, un = tid & 15;
        const int b = m0 + row, u = u0 + un;
        if (b < a.B) {
            float dh_rec = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) dh_rec += red[w][row][un];
            const long long hi = (long long)b * Hp + u;
            float* gp = a.gates + (long long)b * G4 + 16 * (u >> 2) + (u & 3);
            const float si = gp[0], tj = gp[4], sf = gp[8], so = gp[12];
            const float ct = a.c_t[hi], cp = a.c_prev[hi];
            const float tc = tanhf(ct);
            const float dh = a.dh_top[hi] + dh_rec;
            const float dc = a.dc[hi] + dh * so * (1.0f - tc * tc);
            gp[0] = dc * tj * si * (1.0f - si);                 // di
            gp[4] = dc * si * (1.0f - tj * tj);                 // dj
            gp[8] = dc * cp * sf * (1.0f - sf);                 // df
            gp[12] = dh * tc * so * (1.0f - so);                // do
            a.dc[hi] = dc * sf;
        }
    }
}

}  // namespace

hipError_t launch_lstm_fwd_step(hipStream_t s, const LstmFwdArgs& a) {
    dim3 grid((4 * a.Hp) / 16, (a.B + FWD_ROWS - 1) / FWD_ROWS);
    hipLaunchKernelGGL(k_lstm_fwd_step, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_lstm_bwd_step(hipStream_t s, const LstmBwdArgs& a) {
    dim3 grid(a.Hp / 16, (a.B + 15) / 16);
    hipLaunchKernelGGL(k_lstm_bwd_step, grid, dim3(512), 0, s, a);
    return hipGetLastError();
}

}  // namespace fsmg
