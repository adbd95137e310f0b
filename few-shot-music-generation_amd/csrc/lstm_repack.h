// The fragment-ordered copies of K_h the column-split recurrent kernels (lstm_step.hip) read: shared by k_repack_kh and by the
// one-launch repack of every layer and layout (lstm_xcd.hip: k_repack_kh_all).
#pragma once
#include <hip/hip_runtime.h>

namespace fsmg {

// block b of nb (256 threads each) of the grid-stride copy; layout as documented at the top of lstm_step.hip
__device__ __forceinline__ void repack_kh_chunked(const float* __restrict__ Kh, float* __restrict__ fwd, float* __restrict__ bwd,
                                                  int Hp, int b, int nb) {
    const int G4 = 4 * Hp;
    const long long total = (long long)Hp * G4 / 4;            // float4 slots per copy
    for (long long i = (long long)b * blockDim.x + threadIdx.x; i < total; i += (long long)nb * blockDim.x) {
        const int lane = (int)(i & 63), q = lane >> 4, n = lane & 15;
        {   // forward copy
            const int ng = Hp >> 4;
            const int g = (int)((i >> 6) % ng), nbk = (int)((i >> 6) / ng);
            float4 v;
            const float* src = Kh + (long long)(16 * g + 4 * q) * G4 + 16 * nbk + n;
            v.x = src[0]; v.y = src[G4]; v.z = src[2LL * G4]; v.w = src[3LL * G4];
            reinterpret_cast<float4*>(fwd)[i] = v;
        }
        {   // backward copy
            const int ng = G4 >> 4;
            const int g = (int)((i >> 6) % ng), ug = (int)((i >> 6) / ng);
            reinterpret_cast<float4*>(bwd)[i] =
                *reinterpret_cast<const float4*>(Kh + (long long)(16 * ug + n) * G4 + 16 * g + 4 * q);
        }
    }
}

}  // namespace fsmg
