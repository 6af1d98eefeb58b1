// Workgroup-level building blocks of the fused actor/critic MLP kernels (K1, K2, K6).  gfx950 only.
//
// Formulation.  Activations live in LDS *feature-major*: actT[feature][row] with leading dimension
// LD = M + 1 (M = rows per tile, 32 or 64).  A layer is computed transposed,
//        outT (Hout x M) = W (Hout x K) . inT (K x M),
// with the nn.Linear weight (row-major [out][in], straight from the flat parameter buffer in global
// memory / L2) as the MFMA A operand and the LDS activations as the B operand.  With an odd LD every
// LDS access pattern used here (lanes along rows m, or lanes along features with stride LD) is
// bank-conflict free for ds_read_b32 / ds_write_b32.
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32 inputs and accumulation, 64 cycles / SIMD).
//   A: lane l supplies A[row = l & 31][k = l >> 5];  B: lane l supplies B[k = l >> 5][col = l & 31];
//   C/D: acc[r] is C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
// A k-block of 8 is consumed by 4 MFMAs: lane (., hi) feeds k = k0 + 4 hi + j for j = 0..3, so a
// lane's 4 A values are contiguous in the weight row (one 16-byte load when aligned).
#pragma once
#include "erl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ERL_KSTEPS (ERL_MAX_HIDDEN / 8)  // max k-blocks of 8 per layer (K <= 128)

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 4 consecutive weights W[o][kb .. kb+3] with bounds (k < K) and alignment handling.
__device__ __forceinline__ float4 load_w4(const float *__restrict__ Wrow, int kb, int K, bool vec_ok)
{
    if (vec_ok && kb + 3 < K) return *reinterpret_cast<const float4 *>(Wrow + kb);
    float4 w;
    w.x = (kb + 0 < K) ? Wrow[kb + 0] : 0.f;
    w.y = (kb + 1 < K) ? Wrow[kb + 1] : 0.f;
    w.z = (kb + 2 < K) ? Wrow[kb + 2] : 0.f;
    w.w = (kb + 3 < K) ? Wrow[kb + 3] : 0.f;
    return w;
}

// ---------------------------------------------------------------------------------------------
// forward layer: outT[o][m] = f( sum_k W[o][k] inT[k][m] + bias[o] ),  f = GELU (and gradT = GELU')
// or identity.  W: global [Hout][K];  inT: LDS, rows k in [K, Kpad) must be zero;  Hout % 32 == 0.
// ---------------------------------------------------------------------------------------------
template <int M, int NW, bool ACT, bool KEEP_GRAD, int CH = ERL_KSTEPS>
__device__ __forceinline__ void layer_forward(const float *__restrict__ W, const float *__restrict__ bias, int Hout, int K,
                                              const float *inT, float *outT, float *gradT)
{
    constexpr int LD = M + 1, MT = M / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int Kpad = (K + 7) & ~7;
    const bool vec_ok = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    const int ntiles = (Hout >> 5) * MT;
    for (int tile = wave; tile < ntiles; tile += NW) {
        const int ot = tile / MT, mt = tile - ot * MT;
        const float *Wrow = W + (size_t)(ot * 32 + l31) * K;
        const float *bcol = inT + mt * 32 + l31;
        f32x16 acc = {0};
#pragma unroll
        for (int c0 = 0; c0 < ERL_KSTEPS; c0 += CH) {  // weights are fetched CH k-blocks at a time (CH*4 VGPRs)
            if (c0 * 8 < Kpad) {
                float4 wreg[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if ((c0 + i) * 8 < Kpad) wreg[i] = load_w4(Wrow, (c0 + i) * 8 + 4 * hi, K, vec_ok);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if ((c0 + i) * 8 < Kpad) {
                        const float *b = bcol + ((c0 + i) * 8 + 4 * hi) * LD;
                        const float b0 = b[0], b1 = b[LD], b2 = b[2 * LD], b3 = b[3 * LD];
                        acc = mfma32(wreg[i].x, b0, acc);
                        acc = mfma32(wreg[i].y, b1, acc);
                        acc = mfma32(wreg[i].z, b2, acc);
                        acc = mfma32(wreg[i].w, b3, acc);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ot * 32 + crow(r, hi);
            const float z = acc[r] + bias[o];
            const int idx = o * LD + mt * 32 + l31;
            if (ACT) {
                float y, g;
                gelu_and_grad(z, y, g);
                outT[idx] = y;
                if (KEEP_GRAD) gradT[idx] = g;
            } else {
                outT[idx] = z;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward through a layer's input: gT[i][m] <- gT[i][m] * sum_o W[o][i] dZT[o][m]
// (gT holds GELU'(z_in) on entry and dL/dz_in on exit).  W: global [Hout][Hin]; Hout, Hin % 32 == 0.
// ---------------------------------------------------------------------------------------------
template <int M, int NW, int CH = ERL_KSTEPS>
__device__ __forceinline__ void layer_backward_input(const float *__restrict__ W, int Hout, int Hin, const float *dZT, float *gT)
{
    constexpr int LD = M + 1, MT = M / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int ntiles = (Hin >> 5) * MT;
    for (int tile = wave; tile < ntiles; tile += NW) {
        const int it = tile / MT, mt = tile - it * MT;
        const float *Wcol = W + it * 32 + l31;  // A[row = i][k = o] = W[o][i]: lanes along i are contiguous
        const float *bcol = dZT + mt * 32 + l31;
        f32x16 acc = {0};
#pragma unroll
        for (int c0 = 0; c0 < ERL_KSTEPS; c0 += CH) {
            if (c0 * 8 < Hout) {
                float4 wreg[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if ((c0 + i) * 8 < Hout) {
                        const float *w = Wcol + (size_t)((c0 + i) * 8 + 4 * hi) * Hin;
                        wreg[i] = make_float4(w[0], w[Hin], w[2 * Hin], w[3 * Hin]);
                    }
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if ((c0 + i) * 8 < Hout) {
                        const float *b = bcol + ((c0 + i) * 8 + 4 * hi) * LD;
                        const float b0 = b[0], b1 = b[LD], b2 = b[2 * LD], b3 = b[3 * LD];
                        acc = mfma32(wreg[i].x, b0, acc);
                        acc = mfma32(wreg[i].y, b1, acc);
                        acc = mfma32(wreg[i].z, b2, acc);
                        acc = mfma32(wreg[i].w, b3, acc);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = (it * 32 + crow(r, hi)) * LD + mt * 32 + l31;
            gT[idx] = gT[idx] * acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient accumulation over one row tile: acc[j] (+)= dZT (Hout x M) . inT^T (M x Kc)
// for this wave's output tiles j (tile id = wave + j*NW over (Hout/32) x (Kc/32)).  The accumulators
// stay in registers across row tiles; store_weight_grad writes them out once at the end.
// ---------------------------------------------------------------------------------------------
template <int M, int NW, int NT>
__device__ __forceinline__ void weight_grad_accumulate(f32x16 (&acc)[NT], int Hout, int Kc /* padded to 32 */,
                                                       const float *dZT, const float *inT)
{
    constexpr int LD = M + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int kt_count = Kc >> 5, ntiles = (Hout >> 5) * kt_count;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tile = wave + j * NW;
        if (tile < ntiles) {
            const int ot = tile / kt_count, it = tile - ot * kt_count;
            const float *a = dZT + (ot * 32 + l31) * LD + 4 * hi;  // A[row = o][k = m]
            const float *b = inT + (it * 32 + l31) * LD + 4 * hi;  // B[k = m][col = i]
#pragma unroll
            for (int k0 = 0; k0 < M; k0 += 8) {
                acc[j] = mfma32(a[k0 + 0], b[k0 + 0], acc[j]);
                acc[j] = mfma32(a[k0 + 1], b[k0 + 1], acc[j]);
                acc[j] = mfma32(a[k0 + 2], b[k0 + 2], acc[j]);
                acc[j] = mfma32(a[k0 + 3], b[k0 + 3], acc[j]);
            }
        }
    }
}

template <int NW, int NT>
__device__ __forceinline__ void store_weight_grad(const f32x16 (&acc)[NT], int Hout, int Kc, int K /* real in-dim */,
                                                  float *__restrict__ dW /* [Hout][K] */)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int kt_count = Kc >> 5, ntiles = (Hout >> 5) * kt_count;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tile = wave + j * NW;
        if (tile < ntiles) {
            const int ot = tile / kt_count, it = tile - ot * kt_count;
            const int i = it * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ot * 32 + crow(r, hi);
                if (i < K) dW[(size_t)o * K + i] = acc[j][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gather + normalise one tile of states into XT (feature-major), zero-padding features to Sc rows and
// rows beyond `valid`.  row_of(m) gives the global state row of tile row m.  Optionally mirrors the raw
// rows to `copy_out` (rollout: states[t] = state).
// ---------------------------------------------------------------------------------------------
template <int M, typename RowFn>
__device__ __forceinline__ void gather_states(float *XT, int Sc, int S, const float *__restrict__ states,
                                              const float *__restrict__ avg, const float *__restrict__ sd, int valid,
                                              RowFn row_of, float *__restrict__ copy_out, int64_t copy_row0)
{
    constexpr int LD = M + 1;
    for (int e = threadIdx.x; e < M * Sc; e += blockDim.x) {
        const int m = e / Sc, c = e - m * Sc;
        float x = 0.f;
        if (m < valid && c < S) {
            const int64_t row = row_of(m);
            const float raw = states[row * S + c];
            if (copy_out) copy_out[(copy_row0 + m) * S + c] = raw;
            x = (raw - avg[c]) / (sd[c] + 1e-4f);  // (s - avg) / (std + 1e-4), AgentPPO.py:360-361
        }
        XT[c * LD + m] = x;
    }
}

// small dense output layer on the vector ALU: YT[a][m] = sum_i W3[a][i] HT[i][m] + b3[a]
template <int M>
__device__ __forceinline__ void output_layer(const float *__restrict__ W3, const float *__restrict__ b3, int out, int h,
                                             const float *HT, float *YT)
{
    constexpr int LD = M + 1;
    for (int e = threadIdx.x; e < out * M; e += blockDim.x) {
        const int a = e / M, m = e - a * M;
        const float *w = W3 + (size_t)a * h;
        float s0 = 0.f, s1 = 0.f;
        for (int i = 0; i < h; i += 2) {
            s0 = fmaf(w[i], HT[i * LD + m], s0);
            s1 = fmaf(w[i + 1], HT[(i + 1) * LD + m], s1);
        }
        YT[a * LD + m] = (s0 + s1) + b3[a];
    }
}
