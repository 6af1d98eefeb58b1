// fp32 MFMA 32x32x2 helpers (weight-gradient tiles of K6, MFMA self-test).  gfx950 only.
//
// v_mfma_f32_32x32x2_f32 (exact fp32 inputs and accumulation, 64 cycles / SIMD):
//   A: lane l supplies A[row = l & 31][k = l >> 5];  B: lane l supplies B[k = l >> 5][col = l & 31];
//   C/D: acc[r] is C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
#pragma once
#include "erl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
