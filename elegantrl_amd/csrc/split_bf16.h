// Three-way bf16 splitting of fp32 MFMA operands and the LDS image layout that goes with it (the PPO minibatch kernel,
// ppo_step_s3_impl.h).  x = h + m + l exactly, each part a bf16 (round to nearest); a
// product is accumulated in fp32 from the six partial products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16 (every bf16 x bf16
// product is exact in fp32, the three dropped terms are <= 2^-23 |a b|): as close to fp64 as v_mfma_f32_32x32x2_f32
// (tools/split_mfma_probe.hip, tests/test_kernels_gpu.py::test_ppo_step_split_arith).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "mlp_tiles.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

typedef float f32x2_sb __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)      // v_cvt_pk_bf16_f32 (round to nearest even): a in the low half
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_sb{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// The residual x - (one half of the packed pair p), the step between two levels of a split: unpack (v_lshlrev_b32 / v_and_b32) +
// v_sub_f32.  ERL_SPLIT_DOT2=1 forms it with ONE instruction instead, v_dot2_f32_bf16 D = A.lo B.lo + A.hi B.hi + C with B = {-1, 0} /
// {0, -1} from a scalar register (4 vector instructions less per pair of a three-way split, 11 -> 7).  Round 6 built and measured it, and
// it is OFF: the bits are the same (tools/dot2_split_probe.hip: 0 mismatches over 8.2 M pairs of every class, profiles/r06_dot2_split_probe.json)
// but the instruction is no cheaper than the two it replaces beside the bf16 MFMAs -- the minibatch kernel's backward phase went 13.8k ->
// 15.4k cycles, the kernel 36.0 -> 35.4-36.2 us, the step 2.067 -> 2.089 ms (profiles/r06_dot2_split_ab.txt: same box, alternating
// processes) -- and as inline assembly it is invisible to the compiler's hazard recogniser (a transcendental's or an MFMA's result read
// too early: the rollout kernels' actions came out wrong, tests/test_agent_gpu.py::test_explore_env_reproduces_reference_rollout).
#ifndef ERL_SPLIT_DOT2
#define ERL_SPLIT_DOT2 0
#endif
__device__ __forceinline__ float sub_bf_lo(float x, uint32_t p)
{
#if ERL_SPLIT_DOT2
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "s"(0x0000bf80u), "v"(x));
    return r;
#else
    return x - bf_lo(p);
#endif
}
__device__ __forceinline__ float sub_bf_hi(float x, uint32_t p)
{
#if ERL_SPLIT_DOT2
    float r;
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "s"(0xbf800000u), "v"(x));
    return r;
#else
    return x - bf_hi(p);
#endif
}

// (x0, x1) -> three packed bf16 pairs with h + m + l == x exactly (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|, the last residual has
// <= 8 significant bits left)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pk_bf16(x0, x1);
    const float r0 = sub_bf_lo(x0, h), r1 = sub_bf_hi(x1, h);
    m = pk_bf16(r0, r1);
    const float q0 = sub_bf_lo(r0, m), q1 = sub_bf_hi(r1, m);
    l = pk_bf16(q0, q1);
}

struct Parts {          // one MFMA operand (8 k-values of one row / column) in its three parts
    u32x4 h, m, l;
};

// acc[8 a .. 8 a + 7] of a tile -> the operand of k-step (tile, a)
__device__ __forceinline__ Parts split8(const f32x16 &t, int a)
{
    Parts p;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        uint32_t h, m, l;
        split2(t[8 * a + 2 * e], t[8 * a + 2 * e + 1], h, m, l);
        p.h[e] = h; p.m[e] = m; p.l[e] = l;
    }
    return p;
}

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// the six partial products, smallest first
__device__ __forceinline__ void mma6(const Parts &a, const Parts &b, f32x16 &acc)
{
    acc = mfma_bf(a.m, b.m, acc);
    acc = mfma_bf(a.l, b.h, acc);
    acc = mfma_bf(a.h, b.l, acc);
    acc = mfma_bf(a.m, b.h, acc);
    acc = mfma_bf(a.h, b.m, acc);
    acc = mfma_bf(a.h, b.h, acc);
}

// ---- LDS images: [row][3 parts][CP chunks of 16 bytes], CP = K / 8 in {4, 8, 16}; chunk c of a row sits at c ^ swz(row).
// Conditions (banks: 64 dwords for reads, 32 for writes; lane groups per /opt/skills/guides/MI355X_MICROARCH.md): for a fixed
// chunk, (a) the 16 rows of a ds_read_b128 lane group -- four aligned quads with distinct (row >> 2) & 3 -- and (c) 8 consecutive
// rows of a ds_write_b128 group must land in distinct 16-byte bank groups; (b) a transposing read's 32 lanes cover 4 aligned
// rows x 4 aligned chunks.  With r0..r3 the row's low bits:
template <int CP>
__device__ __forceinline__ int swz(int r)
{
    const int r0 = r & 1, r1 = (r >> 1) & 1, r2 = (r >> 2) & 1, r3 = (r >> 3) & 1;
    if (CP == 16) return ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2;
    if (CP == 8) return (r1 << 2) | (r2 << 1) | (r0 ^ r3);
    return (r2 << 1) | (r1 ^ r3);
}

typedef unsigned char u8;

__device__ __forceinline__ u32x2 lds_tr(const u8 *p)
{
    typedef __attribute__((address_space(3))) s16x4_t *lptr;
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(uint32_t)(uintptr_t)p));
}

__device__ __forceinline__ Parts parts_of(const u32x2 (&r)[6])
{
    Parts a;
    a.h = u32x4{r[0].x, r[0].y, r[1].x, r[1].y};
    a.m = u32x4{r[2].x, r[2].y, r[3].x, r[3].y};
    a.l = u32x4{r[4].x, r[4].y, r[5].x, r[5].y};
    return a;
}

// one operand (feature tile `tile`, 16 samples of k-step ks) of a weight gradient from a sample-major image
template <int CP>
struct TrOperand {
    const u8 *b0, *b1;
    int x0, x1;
    __device__ __forceinline__ TrOperand(const u8 *S, int lane)
    {
        const int q = lane >> 4, kb = q >> 1, half = q & 1, t = lane & 15, rr = t >> 2, u = t & 3;
        const int ccl = 2 * half + (u >> 1), sub = 8 * (u & 1);
        const int rl0 = 8 * kb + rr, rl1 = rl0 + 4;
        b0 = S + rl0 * (48 * CP) + sub;
        b1 = S + rl1 * (48 * CP) + sub;
        x0 = 16 * (ccl ^ swz<CP>(rl0));
        x1 = 16 * (ccl ^ swz<CP>(rl1));
    }
    __device__ __forceinline__ void issue(int tile, int ks, u32x2 (&dst)[6]) const
    {
        const u8 *p0 = b0 + 16 * ks * (48 * CP) + ((64 * tile) ^ x0), *p1 = b1 + 16 * ks * (48 * CP) + ((64 * tile) ^ x1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            dst[2 * pl] = lds_tr(p0 + pl * 16 * CP);
            dst[2 * pl + 1] = lds_tr(p1 + pl * 16 * CP);
        }
    }
};

}  // namespace
