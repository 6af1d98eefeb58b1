// Hidden layers of the latency-form rollout kernels (mlp.hip rollout_split_kernel, rollout_fused.hip) on the bf16 matrix pipe.
//
// ActorPPO.get_action / CriticPPO.forward (elegantrl/agents/AgentPPO.py:368-376, :435-441) in the rollout are two 128-wide layers on a
// 16-env tile per workgroup -- a dependent chain whose length is MFMA issue time.  v_mfma_f32_16x16x4_f32 gives 256 FLOP / clk / CU;
// the same product from three-way bf16 splits of both operands (split_bf16.h: x = h + m + l exactly, six partial products of weight
// >= 2^-16, fp32 accumulation: as close to fp64 as the fp32 instruction) on v_mfma_f32_16x16x32_bf16 issues 6 x 16 cycles per
// 16 x 16 x 32 block against 8 x 32: 2.7x less matrix-pipe time for the same tile.
//
//   A (weights): lane (m = lane & 15, q = lane >> 4) holds row m of the wave's 16 output features, k = 32 ks + 8 q .. + 7, as three
//                packed-bf16 register quadruples (`Parts`), split ONCE per launch;
//   B (activations): lane (n = lane & 15, q) holds sample n, the same k -- read as 16 bytes per part from a sample-major LDS tile of
//                three bf16 planes (state tile: 64 + 8 columns per row; hidden tile: 128 + 8: rows 36 / 68 dwords apart, so the 16
//                samples of a ds_read_b128 lane group hit 16 distinct 4-bank groups), written by whoever produces the value;
//   D: acc[r] = feature 4 q + r of sample n -- the layout of the fp32 instruction, so everything downstream is unchanged.
//
// BOTH kernels call rb_mma6() / rb_sum() with the same operands in the same order: the per-step and the persistent rollout stay bit-identical
// (tests/test_rollout_fused_gpu.py).
#pragma once
#include "mlp_chain.h"
#include "split_bf16.h"

namespace {

constexpr int RB_XLD = 144;                  // bytes per sample row of one part plane of a state tile (64 bf16 + 16 bytes)
constexpr int RB_TLD = 272;                  // ... of a hidden tile (128 bf16 + 16 bytes)
constexpr int RB_XBYTES = 3 * 16 * RB_XLD;   // 6912
constexpr int RB_TBYTES = 3 * 16 * RB_TLD;   // 13056

__device__ __forceinline__ f32x4 mfma_bf16k32(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ Parts rb_split8(const float4 &u, const float4 &v)
{
    Parts p;
    uint32_t h, m, l;
    split2(u.x, u.y, h, m, l); p.h[0] = h; p.m[0] = m; p.l[0] = l;
    split2(u.z, u.w, h, m, l); p.h[1] = h; p.m[1] = m; p.l[1] = l;
    split2(v.x, v.y, h, m, l); p.h[2] = h; p.m[2] = m; p.l[2] = l;
    split2(v.z, v.w, h, m, l); p.h[3] = h; p.m[3] = m; p.l[3] = l;
    return p;
}

// row[32 ks + 8 q .. + 7] of a weight row of length K (zeros beyond K), split
template <bool VEC>
__device__ __forceinline__ Parts rb_load_w(const float *__restrict__ row, int ks, int q, int K)
{
    const float4 u = load4<VEC>(row, 32 * ks + 8 * q, K), v = load4<VEC>(row, 32 * ks + 8 * q + 4, K);
    return rb_split8(u, v);
}

// four values of sample `n` at columns c0 .. c0 + 3 -> the three planes of a tile (ld = RB_XLD / RB_TLD)
__device__ __forceinline__ void rb_tile_put(u8 *T, int ld, int n, int c0, float x0, float x1, float x2, float x3)
{
    uint32_t h0, m0, l0, h1, m1, l1;
    split2(x0, x1, h0, m0, l0);
    split2(x2, x3, h1, m1, l1);
    u8 *p = T + n * ld + 2 * c0;
    *reinterpret_cast<u32x2 *>(p) = u32x2{h0, h1};
    *reinterpret_cast<u32x2 *>(p + 16 * ld) = u32x2{m0, m1};
    *reinterpret_cast<u32x2 *>(p + 32 * ld) = u32x2{l0, l1};
}

// two values at columns c0, c0 + 1 (c0 even)
__device__ __forceinline__ void rb_tile_put2(u8 *T, int ld, int n, int c0, float x0, float x1)
{
    uint32_t h, m, l;
    split2(x0, x1, h, m, l);
    u8 *p = T + n * ld + 2 * c0;
    *reinterpret_cast<uint32_t *>(p) = h;
    *reinterpret_cast<uint32_t *>(p + 16 * ld) = m;
    *reinterpret_cast<uint32_t *>(p + 32 * ld) = l;
}

__device__ __forceinline__ Parts rb_tile_get(const u8 *T, int ld, int n, int ks, int q)
{
    const u8 *p = T + n * ld + 64 * ks + 16 * q;
    Parts b;
    b.h = *reinterpret_cast<const u32x4 *>(p);
    b.m = *reinterpret_cast<const u32x4 *>(p + 16 * ld);
    b.l = *reinterpret_cast<const u32x4 *>(p + 32 * ld);
    return b;
}

// Two accumulators per dot product: the three large partial products (h h, h m, m h) and the three small ones (m m, h l, l h)
// alternate, so an MFMA depends on the one issued two slots before it (a 4-pass instruction: no stall), and the small terms meet the
// large ones once, in rb_sum.
struct RbAcc {
    f32x4 big, small;
    __device__ __forceinline__ RbAcc() { big = small = f32x4{0.f, 0.f, 0.f, 0.f}; }
};

__device__ __forceinline__ void rb_mma6(const Parts &a, const Parts &b, RbAcc &c)
{
    c.small = mfma_bf16k32(a.m, b.m, c.small);
    c.big = mfma_bf16k32(a.m, b.h, c.big);
    c.small = mfma_bf16k32(a.l, b.h, c.small);
    c.big = mfma_bf16k32(a.h, b.m, c.big);
    c.small = mfma_bf16k32(a.h, b.l, c.small);
    c.big = mfma_bf16k32(a.h, b.h, c.big);
}

__device__ __forceinline__ float rb_sum(const RbAcc &c, int r) { return c.big[r] + c.small[r]; }

}  // namespace
