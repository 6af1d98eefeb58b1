// Register-chained MLP building blocks shared by K1 / K2 / K6 (gfx950, fp32 MFMA 16x16x4).  See ppo_step.hip for the
// formulation: every layer is computed transposed on 16-sample tiles, the result tile of one layer is the B operand
// of the next without leaving the register file, weights are served from zero-padded LDS copies.
#pragma once
#include "mlp_tiles.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PCH
#define PCH 2          // k-tiles (of 16) per software-pipelined weight chunk: 2 x PCH x 4 VGPRs of operand buffer
#endif

namespace {

// flat parameter block of build_mlp([S, h1, h2, out]) (+ action_std_log): offsets in floats (include/erl_hip.h)
struct Dims {
    int S, h1, h2, out;
    __host__ __device__ int64_t oW1() const { return 0; }
    __host__ __device__ int64_t ob1() const { return (int64_t)h1 * S; }
    __host__ __device__ int64_t oW2() const { return ob1() + h1; }
    __host__ __device__ int64_t ob2() const { return oW2() + (int64_t)h2 * h1; }
    __host__ __device__ int64_t oW3() const { return ob2() + h2; }
    __host__ __device__ int64_t ob3() const { return oW3() + (int64_t)out * h2; }
    __host__ __device__ int64_t oStd() const { return ob3() + out; }
    __host__ __device__ int64_t count(bool with_std) const { return oStd() + (with_std ? out : 0); }
};


inline bool mlp_dims_ok(int S, int h1, int h2, int out)
{
    return S >= 1 && S <= ERL_MAX_STATE_DIM && h1 >= 32 && h1 <= ERL_MAX_HIDDEN && (h1 % 32) == 0 && h2 >= 32 &&
           h2 <= ERL_MAX_HIDDEN && (h2 % 32) == 0 && out >= 1 && out <= ERL_MAX_ACTION_DIM;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory
// counter, so in-flight global loads (prefetches) and the gradient-slab stores keep streaming across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

constexpr float kLogSqrt2PiF = 0.91893853320467274178f;  // log(sqrt(2 pi))

// tanh(x) = sign(x) (1 - e) / (1 + e), e = exp(-2 |x|) in (0, 1]: no overflow, no cancellation in 1 + e;
// 1 - e loses nothing below |x| ~ 1e-4 that the result's own fp32 ulp would show (abs err < 2e-7).
__device__ __forceinline__ float fast_tanh(float x)
{
    const float e = __expf(-2.f * fabsf(x));
    return copysignf(__fdividef(1.f - e, 1.f + e), x);
}

// exact-erf GELU and its derivative.  erf through Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, the size of an fp32
// ulp of the result), sharing exp(-z^2/2) with the Gaussian density of the derivative: ~20 VALU ops instead of ~70.
__device__ __forceinline__ void gelu_and_grad_fast(float z, float &y, float &gd)
{
    const float x = fabsf(z) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float u = __expf(-(x * x));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float erfa = fmaf(-p, u, 1.0f);
    const float cdf = 0.5f * (1.0f + copysignf(erfa, z));
    y = z * cdf;
    gd = fmaf(z * u, 0.39894228040143267794f, cdf);
}

// 4 consecutive floats row[k0 .. k0+3] of a row of length K, zeros beyond K.  Branch-free: the address is clamped
// into the row and the result selected, so the loads can be hoisted and pipelined freely.
template <bool VEC>
__device__ __forceinline__ float4 load4(const float *__restrict__ row, int k0, int K)
{
    if (VEC) {   // K % 4 == 0 and 16-byte aligned rows: a 4-group is either fully inside or fully outside
        const int kc = min(k0, K - 4);
        const float4 v = *reinterpret_cast<const float4 *>(row + kc);
        return k0 < K ? v : zero4();
    }
    float x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = row[min(k0 + c, K - 1)];
        x[c] = (k0 + c < K) ? v : 0.f;
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

// row stride of an LDS weight copy with `cols` columns: the smallest 4 * odd >= cols + 1 (see the file header)
__host__ __device__ constexpr int lds_ld(int cols) { return 4 * (2 * ((cols + 7) / 8) + 1); }

// cooperative copy of a row-major [rows][cols] matrix into LDS [rows_pad][ld], zero padded, in two halves so that
// the global round trip overlaps other work: copy_load issues up to MAXV float4 loads per thread (rows_pad *
// cols_pad / 4 <= MAXV * NT, NT = threads per workgroup), copy_store writes them to LDS.
template <bool VEC, int MAXV, int NT>
__device__ __forceinline__ void copy_load(float4 (&v)[MAXV], const float *__restrict__ src, int rows, int cols, int rows_pad,
                                          int cols_pad, int tid)
{
    const int vpr = cols_pad >> 2, total = rows_pad * vpr;   // cols_pad % 4 == 0
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int e = min(tid + u * NT, total - 1);
        const int i = e / vpr, j4 = e - i * vpr;
        v[u] = load4<VEC>(src + (size_t)min(i, rows - 1) * cols, 4 * j4, cols);
        if (i >= rows) v[u] = zero4();
    }
}

template <int MAXV, int NT>
__device__ __forceinline__ void copy_store(const float4 (&v)[MAXV], float *dst, int ld, int rows_pad, int cols_pad, int tid)
{
    const int vpr = cols_pad >> 2, total = rows_pad * vpr;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int e = tid + u * NT;
        if (e < total) {
            const int i = e / vpr, j4 = e - i * vpr;
            *reinterpret_cast<float4 *>(dst + i * ld + 4 * j4) = v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward layer on registers:  out[ot] (16 features x 16 samples) = act( W[16 ot .. +15][:] . in + bias )
// W, bias: zero-padded LDS copies (row stride ldw).  KT = k-tiles of the input (compile time; 0 = use `kt`).
// Weight chunks of 4 k-tiles (16 VGPRs) are software-pipelined one chunk ahead of the MFMAs.
// ---------------------------------------------------------------------------------------------------------
template <bool ACT, int KT, bool KEEPG = true>
__device__ __forceinline__ void forward_layer(const float *W, int ldw, const float *bias, int kt_rt, int nout,
                                              const f32x4 (&in)[8], f32x4 (&outH)[8], f32x4 (&outG)[8], int l15, int q)
{
    constexpr int NCH = KT ? (KT + PCH - 1) / PCH : 8 / PCH;           // chunks per output tile
    constexpr int NC = 8 * NCH;
    const int kt = KT ? KT : kt_rt;
    float4 wq[2][PCH];
    auto issue = [&](int c, float4(&dst)[PCH]) {
        const int ot = c / NCH, th = c % NCH;
#pragma unroll
        for (int j = 0; j < PCH; ++j) {
            const int t = PCH * th + j;
            if (ot < nout && t < kt) dst[j] = *reinterpret_cast<const float4 *>(W + (16 * ot + l15) * ldw + 16 * t + 4 * q);
        }
    };
    issue(0, wq[0]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ot = c / NCH, th = c % NCH;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ot < nout) {
            if (th == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < PCH; ++j) {
                const int t = PCH * th + j;
                if (t < kt) {
                    const float4 wv = wq[c & 1][j];
                    acc = mfma16(wv.x, in[t][0], acc);
                    acc = mfma16(wv.y, in[t][1], acc);
                    acc = mfma16(wv.z, in[t][2], acc);
                    acc = mfma16(wv.w, in[t][3], acc);
                }
            }
            if (th == NCH - 1) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bias + 16 * ot + 4 * q);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[r] + bb[r];
                    if (ACT) {
                        float y, gd;
                        gelu_and_grad_fast(z, y, gd);
                        outH[ot][r] = y;
                        if (KEEPG) outG[ot][r] = gd;
                    } else {
                        outH[ot][r] = z;
                    }
                }
            }
        }
    }
}


}  // namespace
