// K6 for the reference's wider demo networks: one PPO minibatch (gather + actor & critic forward + objective + full backward) for
// net_dims = (256, h2), h2 in {64, 128}, S <= 64, A <= 8 -- examples/demo_A2C_PPO.py:117 trains (256, 128) -- on the bf16 matrix pipe with
// the fp32-equivalent split arithmetic of ppo_step_s3_impl.h (three bf16 parts per operand, six partial products, fp32 accumulation).
// Same contract as the [128,128] kernels (AgentPPO.update_objectives up to the optimizer steps, elegantrl/agents/AgentPPO.py:173-204;
// ActorPPO.get_logprob_entropy :378-386; same slabs, same logged sums), same mapping (grid (ceil(B / 128), 2 networks), four waves,
// one per SIMD, 32 samples each).  What a 256-wide first layer changes:
//
//   * registers: H1 of a wave's 32 samples alone is 128 registers per lane.  It stays (the second layer and the dW2 staging read it);
//     GELU'(z1) (128 more) leaves for a per-workgroup scratch block in global memory as each tile of the first layer finishes and
//     comes back tile by tile as the gate of the backward pass; H2 takes the same trip between the output layer and dW3
//     (lane-contiguous 16-byte accesses, 192 KB per workgroup and network, written once and read once by the same CU).
//   * LDS: the W2 image (h2 x 3 x 256 bf16 = 196 KB at h2 = 128) does not fit.  The optimiser keeps it as four COLUMN-QUARTER images
//     [h2][3][64 bf16] (48 KB each, s3_image.h's layout with K = 64); they stream through two 48 KB slots by LDS-DMA, issued piece by
//     piece behind the MFMAs of the quarter before: the second layer accumulates z2 quarter by quarter (K split), the backward pass
//     produces dZ1 quarter by quarter (64 of its 256 features), and each quarter of dZ1 is staged into the slot its W2 quarter just
//     left and contracted with the staged input (dW1) before the next one is formed -- dZ1 never exists as a whole.  dW2 walks H1 in
//     quarters the same way (re-split from the fp32 registers).  Three slots of 48 KB: [A: W1 image / X image / H1 quarter]
//     [X][Y: W2 quarters / dZ1 quarter / H2^T / dZ2 image].
//
// Everything else -- operand layouts, the phi row permutation, swizzled images, transposing reads for the weight gradients, the fp32
// output layer -- is ppo_step_s3_impl.h's, whose helpers this file uses.
#pragma once
#include "ppo_step_s3_impl.h"
#include "ppo_step_wd_args.h"

namespace {

// ERL_WD_LAST = 1: the second layer's GELU rides the last quarter's MFMAs (fwd_acc_wd<LAST>) instead of running on its own on packed
// fp32 -- measured slower (quarter 3: 4.6k -> 10.5k cycles for 3.3k of GELU saved; 17 scalar-fp32 instructions per element against
// 12.5 packed, and six MFMAs hide four or five of them): off.  ERL_WD_GATE_EARLY = 1: the first gate tiles of the backward pass are
// requested before the output layer (dZ1's first quarter: 8.1k -> 5.3k cycles).
#ifndef ERL_WD_LAST
#define ERL_WD_LAST 0
#endif
#ifndef ERL_WD_GATE_EARLY
#define ERL_WD_GATE_EARLY 1
#endif
#ifndef ERL_WD_DBG
#define ERL_WD_DBG 0
#endif
constexpr int kWdSlot = 49152;
constexpr int kWdSmall = (256 + 128 + 16 + 64 + 64 + 16 + 128) * 4;
constexpr size_t kWdLdsBytes = (size_t)3 * kWdSlot + kS3W3 + kWdSmall;
static_assert(kWdLdsBytes <= 160 * 1024, "LDS budget");
static_assert(2 * kWdSlot >= 128 * PLD * 4, "H2^T (fp32, feature-major) spans two slots");

// one 1 KB piece by LDS-DMA: lane l copies 16 bytes from src + voff (voff = 16 l) to LDS byte lds + 16 l.  Inline assembly the
// compiler does not see (ppo_step_s3_impl.h explains why): the issuing wave waits by hand (s_waitcnt vmcnt(0)) before the barrier that
// publishes the bytes.  The address is a wave-uniform base in scalar registers plus ONE 32-bit lane offset shared by every piece: with
// per-lane 64-bit addresses the compiler computed all of a phase's piece addresses ahead, spilled the register pairs to scratch memory
// and reloaded one in front of every piece -- and a scratch reload returns behind the pieces already in flight (memory operations
// return in order): 7k cycles per wave in one quarter of the backward pass (tools/wide_phase_profile.py, first version).
__device__ __forceinline__ void wd_dma1(const u8 *src, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ void wd_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the lane index, computed where it is asked for (the compiler cannot merge two of these): the thread's indices are re-derived at every
// phase of the kernel instead of living in registers from entry to exit -- long-lived and rarely used, they were what the register
// allocator spilled to scratch memory first, and a scratch reload returns behind every LDS-DMA piece, gate tile and gradient store in
// flight (all of the first version's reloads were `tid & 63` and its relatives)
__device__ __forceinline__ int wd_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// ---------------------------------------------------------------------------------------------------------
// first layer: fwd_s3 (ppo_step_s3_impl.h) with two differences -- GELU' is handed to `done(tile, k-step, values)` as each k-step's
// share of a tile is finished instead of being kept, and NO may be 8.  NK in {2, 4}.
// ---------------------------------------------------------------------------------------------------------
// TROT: the image's row tiles sit rotated in LDS (tile To at position (To + TROT) % NO: a 96 KB image whose first half was streamed into
// the upper slot).
template <int NK, int NO, int CP, int TROT = 0, typename Side, typename Done>
__device__ __forceinline__ void fwd_wd(const u8 *img, const float *bias, Parts (&inP)[NK], const f32x16 (&inH)[(NK + 1) / 2],
                                       f32x16 (&outH)[NO], int m, int hi, const Side &side, const Done &done)
{
    constexpr int ROWB = 48 * CP, PBY = 16 * CP, NC = NO * NK;
    constexpr int EP = 16 / NK;
    static_assert(EP * NK == 16 && EP >= 2 && NK >= 2, "k-steps per tile");
    const int row = phi(m);
    const u8 *base = img + row * ROWB;
    const int x16 = 16 * (swz<CP>(row) ^ hi);
    Parts aq[2];
    auto issue = [&](int c, Parts &dst) {
        const int To = c / NK, ks = c % NK;
        const u8 *p = base + 32 * ((To + TROT) % NO) * ROWB + ((32 * ks) ^ x16);
        dst.h = *reinterpret_cast<const u32x4 *>(p);
        dst.m = *reinterpret_cast<const u32x4 *>(p + PBY);
        dst.l = *reinterpret_cast<const u32x4 *>(p + 2 * PBY);
    };
    f32x16 prev, prev1;
    float gdt[EP];
    constexpr float kC = 0.84932180028801904272f;
    constexpr float kP = 0.3275911f * 0.70710678118654752440f / kC;
    float z[EP], xa[EP], tt[EP], uu[EP], pp[EP];
    auto stage = [&](int Tp, int ks, int s, bool fence = true) {
#pragma unroll
        for (int i = 0; i < EP; ++i) {
            const int e = EP * ks + i;
            if (s == 0) {
                z[i] = prev[e] + prev1[e];
                xa[i] = fabsf(z[i]) * kC;
                tt[i] = fmaf(xa[i], kP, 1.0f);
            } else if (s == 1) {
                tt[i] = __builtin_amdgcn_rcpf(tt[i]);
                uu[i] = __builtin_amdgcn_exp2f(-(xa[i] * xa[i]));
            } else if (s == 2) {
                pp[i] = fmaf(tt[i], 0.5f * 1.061405429f, 0.5f * -1.453152027f);
                pp[i] = fmaf(tt[i], pp[i], 0.5f * 1.421413741f);
                pp[i] = fmaf(tt[i], pp[i], 0.5f * -0.284496736f);
            } else if (s == 3) {
                pp[i] = fmaf(tt[i], pp[i], 0.5f * 0.254829592f);
                pp[i] = pp[i] * tt[i];
                pp[i] = fmaf(-pp[i], uu[i], 0.5f);
            } else if (s == 4) {
                pp[i] = copysignf(pp[i], z[i]) + 0.5f;
                uu[i] = z[i] * uu[i];
            } else {
                float y = z[i] * pp[i];
                float gd = fmaf(uu[i], 0.39894228040143267794f, pp[i]);
                asm volatile("" : "+v"(y), "+v"(gd));
                gdt[i] = gd;
                outH[Tp][e] = y;
            }
        }
        if (s == 5) done(Tp, ks, gdt);        // GELU' of elements EP ks .. EP ks + EP - 1 leaves right away
        if (fence) __builtin_amdgcn_sched_barrier(0);
    };
    auto jit = [&](int ks, int s) {
        if (ks < NK && s < 4) {
            uint32_t h, mm, l;
            split2(inH[ks >> 1][8 * (ks & 1) + 2 * s], inH[ks >> 1][8 * (ks & 1) + 2 * s + 1], h, mm, l);
            asm volatile("" : "+v"(h), "+v"(mm), "+v"(l));
            inP[ks].h[s] = h; inP[ks].m[s] = mm; inP[ks].l[s] = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) jit(0, s);
    issue(0, aq[0]);
    f32x16 nb;
    auto load_bias = [&](int To) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4 b0 = *reinterpret_cast<const float4 *>(bias + 32 * To + 16 * a + 8 * hi);
            const float4 b1 = *reinterpret_cast<const float4 *>(bias + 32 * To + 16 * a + 8 * hi + 4);
            nb[8 * a + 0] = b0.x; nb[8 * a + 1] = b0.y; nb[8 * a + 2] = b0.z; nb[8 * a + 3] = b0.w;
            nb[8 * a + 4] = b1.x; nb[8 * a + 5] = b1.y; nb[8 * a + 6] = b1.z; nb[8 * a + 7] = b1.w;
        }
    };
    load_bias(0);
#pragma unroll
    for (int To = 0; To < NO; ++To) {
        f32x16 acc = nb, acc1 = {0};
        if (To + 1 < NO) load_bias(To + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = To * NK + ks;
            if (c + 1 < NC) issue(c + 1, aq[(c + 1) & 1]);
            const Parts &a = aq[c & 1], &b = inP[ks];
            auto fill = [&](int s) {
                if (To > 0) stage(To - 1, ks, s);
                else jit(ks + 1, s);
            };
            acc = mfma_bf(a.m, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
            side(c);
            __builtin_amdgcn_sched_barrier(0);
            fill(0);
            acc1 = mfma_bf(a.l, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(1);
            acc = mfma_bf(a.h, b.l, acc);
            __builtin_amdgcn_sched_barrier(0);
            fill(2);
            acc1 = mfma_bf(a.m, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(3);
            acc = mfma_bf(a.h, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
            fill(4);
            acc1 = mfma_bf(a.h, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(5);
        }
        prev = acc;
        prev1 = acc1;
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int s = 0; s < 6; ++s) stage(NO - 1, ks, s, false);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// one K quarter of the second layer: Z[To] += Wq[32 To + .][64 inputs] . in, Wq a column-quarter image (CP = 8), in = the two fp32
// tiles H[2 Q], H[2 Q + 1] (split on the way, behind tile 0's MFMAs).  Two accumulators alternate as in fwd_s3; their sum is folded into
// Z behind the next tile's MFMAs.
// ---------------------------------------------------------------------------------------------------------
// LAST: this is the layer's last quarter -- behind tile To's MFMAs the finished sum of tile To - 1 goes through the GELU stages of
// fwd_wd (outH, outG instead of Z).
template <int Q, int NO, int CP, bool LAST, int NH, typename Side, typename Keep>
__device__ __forceinline__ void fwd_acc_wd(const u8 *img, const f32x16 (&H)[NH], f32x16 (&Z)[NO], int m, int hi, const Side &side, const Keep &keep,
                                           f32x16 *outH = nullptr, f32x16 *outG = nullptr)
{
    static_assert(2 * Q + 1 < NH, "quarter outside the input");
    constexpr int NK = 4, ROWB = 48 * CP, PBY = 16 * CP, NC = NO * NK;
    const int row = phi(m);
    const u8 *base = img + row * ROWB;
    const int x16 = 16 * (swz<CP>(row) ^ hi);
    Parts aq[2], inP[NK];
    auto issue = [&](int c, Parts &dst) {
        const int To = c / NK, ks = c % NK;
        const u8 *p = base + 32 * To * ROWB + ((32 * ks) ^ x16);
        dst.h = *reinterpret_cast<const u32x4 *>(p);
        dst.m = *reinterpret_cast<const u32x4 *>(p + PBY);
        dst.l = *reinterpret_cast<const u32x4 *>(p + 2 * PBY);
    };
    auto jit = [&](int ks, int s) {
        if (ks < NK && s < 4) {
            uint32_t h, mm, l;
            split2(H[2 * Q + (ks >> 1)][8 * (ks & 1) + 2 * s], H[2 * Q + (ks >> 1)][8 * (ks & 1) + 2 * s + 1], h, mm, l);
            asm volatile("" : "+v"(h), "+v"(mm), "+v"(l));
            inP[ks].h[s] = h; inP[ks].m[s] = mm; inP[ks].l[s] = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 prev, prev1;
    auto merge = [&](int Tp, int ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = prev[4 * ks + i] + prev1[4 * ks + i];
            asm volatile("" : "+v"(v));
            Z[Tp][4 * ks + i] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // LAST: exact-erf GELU and its derivative of elements 4 ks .. 4 ks + 3 of tile Tp, stage s of 6 (fwd_wd's)
    constexpr float kC = 0.84932180028801904272f;
    constexpr float kP = 0.3275911f * 0.70710678118654752440f / kC;
    float z[4], xa[4], tt[4], uu[4], pp[4];
    auto stage = [&](int Tp, int ks, int s, bool fence = true) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * ks + i;
            if (s == 0) {
                z[i] = prev[e] + prev1[e];
                xa[i] = fabsf(z[i]) * kC;
                tt[i] = fmaf(xa[i], kP, 1.0f);
            } else if (s == 1) {
                tt[i] = __builtin_amdgcn_rcpf(tt[i]);
                uu[i] = __builtin_amdgcn_exp2f(-(xa[i] * xa[i]));
            } else if (s == 2) {
                pp[i] = fmaf(tt[i], 0.5f * 1.061405429f, 0.5f * -1.453152027f);
                pp[i] = fmaf(tt[i], pp[i], 0.5f * 1.421413741f);
                pp[i] = fmaf(tt[i], pp[i], 0.5f * -0.284496736f);
            } else if (s == 3) {
                pp[i] = fmaf(tt[i], pp[i], 0.5f * 0.254829592f);
                pp[i] = pp[i] * tt[i];
                pp[i] = fmaf(-pp[i], uu[i], 0.5f);
            } else if (s == 4) {
                pp[i] = copysignf(pp[i], z[i]) + 0.5f;
                uu[i] = z[i] * uu[i];
            } else {
                float y = z[i] * pp[i];
                float gd = fmaf(uu[i], 0.39894228040143267794f, pp[i]);
                asm volatile("" : "+v"(y), "+v"(gd));
                outG[Tp][e] = gd;
                outH[Tp][e] = y;
            }
        }
        if (fence) __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) jit(0, s);
    issue(0, aq[0]);
#pragma unroll
    for (int To = 0; To < NO; ++To) {
        f32x16 acc = Z[To], acc1 = {0};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = To * NK + ks;
            if (c + 1 < NC) issue(c + 1, aq[(c + 1) & 1]);
            const Parts &a = aq[c & 1], &b = inP[ks];
            auto fill = [&](int s) {
                if (To == 0) jit(ks + 1, s);
                else if (LAST) stage(To - 1, ks, s);
                else if (s == 0) merge(To - 1, ks);
            };
            acc = mfma_bf(a.m, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
            side(c);
            __builtin_amdgcn_sched_barrier(0);
            fill(0);
            acc1 = mfma_bf(a.l, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(1);
            acc = mfma_bf(a.h, b.l, acc);
            __builtin_amdgcn_sched_barrier(0);
            fill(2);
            acc1 = mfma_bf(a.m, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(3);
            acc = mfma_bf(a.h, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
            if (LAST && To > 0) fill(4);
            if (To == 1) {                  // (this k-step's operand is complete since tile 0; its copy for dW2 leaves behind tile 1)
                keep(ks, inP[ks]);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc1 = mfma_bf(a.h, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (LAST && To > 0) fill(5);
        }
        prev = acc;
        prev1 = acc1;
    }
    if (LAST) {
        // the last tile has no MFMAs to ride behind: unfenced, so that its elements' chains interleave
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
            for (int s = 0; s < 6; ++s) stage(NO - 1, ks, s, false);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) Z[NO - 1][e] = prev[e] + prev1[e];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// one quarter of the backward pass through W2: bwd_s3 (ppo_step_s3_impl.h) on a column-quarter image (NO = 2 tiles = 64 of the first
// layer's features), with the split of dz skipped when an earlier quarter has done it (PRESPLIT) and a per-k-step hook.
// ---------------------------------------------------------------------------------------------------------
// KROT: the image's rows sit rotated by 16 KROT in LDS (see fwd_wd).
template <int NK, int NO, int CP, bool PRESPLIT, int KROT = 0, typename Side>
__device__ __forceinline__ void bwd_wd(const u8 *img, Parts (&dzP)[NK], const f32x16 (&dzH)[NK / 2], const f32x16 (&gate)[NO],
                                       Parts (&outP)[2 * NO], int lane, const Side &side)
{
    constexpr int ROWB = 48 * CP, PBY = 16 * CP, NC = NO * NK;
    constexpr int EP = 16 / NK;
    static_assert(EP * NK == 16 && EP % 2 == 0, "k-steps per tile");
    constexpr int NPR = EP / 2;
    const int q = lane >> 4, kb = q >> 1, half = q & 1, t = lane & 15, rr = t >> 2, u = t & 3;
    const int su = ((u & 1) << 1) | (u >> 1);
    const int ccl = 2 * half + (su >> 1), sub = 8 * (su & 1);
    const int rl0 = 8 * kb + rr, rl1 = rl0 + 4;
    const u8 *b0 = img + rl0 * ROWB + sub, *b1 = img + rl1 * ROWB + sub;
    const int x0 = 16 * (ccl ^ swz<CP>(rl0)), x1 = 16 * (ccl ^ swz<CP>(rl1));
    u32x2 rq[2][6];
    auto issue = [&](int c, u32x2(&dst)[6]) {
        const int To = c / NK, ks = c % NK;
        const u8 *p0 = b0 + 16 * ((ks + KROT) % NK) * ROWB + ((64 * To) ^ x0), *p1 = b1 + 16 * ((ks + KROT) % NK) * ROWB + ((64 * To) ^ x1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            dst[2 * pl] = lds_tr(p0 + pl * PBY);
            dst[2 * pl + 1] = lds_tr(p1 + pl * PBY);
        }
    };
    auto jit = [&](int ks, int s) {
        if (!PRESPLIT && ks < NK && s < 4) {
            uint32_t h, mm, l;
            split2(dzH[ks >> 1][8 * (ks & 1) + 2 * s], dzH[ks >> 1][8 * (ks & 1) + 2 * s + 1], h, mm, l);
            asm volatile("" : "+v"(h), "+v"(mm), "+v"(l));
            dzP[ks].h[s] = h; dzP[ks].m[s] = mm; dzP[ks].l[s] = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 prev, prev1;
    float v0[NPR], v1[NPR];
    uint32_t sh[NPR], sm[NPR];
    auto gstage = [&](int Tp, int ks, int s, bool fence = true) {
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            const int e = EP * ks + 2 * i;
            if (s == 0) {
                v0[i] = gate[Tp][e] * (prev[e] + prev1[e]);
                v1[i] = gate[Tp][e + 1] * (prev[e + 1] + prev1[e + 1]);
                asm volatile("" : "+v"(v0[i]), "+v"(v1[i]));
            } else if (s == 1) {
                sh[i] = pk_bf16(v0[i], v1[i]);
                v0[i] = sub_bf_lo(v0[i], sh[i]);
                v1[i] = sub_bf_hi(v1[i], sh[i]);
                asm volatile("" : "+v"(sh[i]), "+v"(v0[i]), "+v"(v1[i]));
            } else if (s == 2) {
                sm[i] = pk_bf16(v0[i], v1[i]);
                v0[i] = sub_bf_lo(v0[i], sm[i]);
                v1[i] = sub_bf_hi(v1[i], sm[i]);
                asm volatile("" : "+v"(sm[i]), "+v"(v0[i]), "+v"(v1[i]));
            } else if (s == 3) {
                uint32_t l = pk_bf16(v0[i], v1[i]);
                asm volatile("" : "+v"(l));
                Parts &o = outP[2 * Tp + (e >> 3)];
                o.h[(e & 7) >> 1] = sh[i]; o.m[(e & 7) >> 1] = sm[i]; o.l[(e & 7) >> 1] = l;
            }
        }
        if (fence) __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) jit(0, s);
    issue(0, rq[0]);
#pragma unroll
    for (int To = 0; To < NO; ++To) {
        f32x16 acc = {0}, acc1 = {0};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = To * NK + ks;
            if (c + 1 < NC) issue(c + 1, rq[(c + 1) & 1]);
            const Parts a = parts_of(rq[c & 1]);
            const Parts &b = dzP[ks];
            auto fill = [&](int s) {
                if (To > 0) gstage(To - 1, ks, s);
                else jit(ks + 1, s);
            };
            acc = mfma_bf(a.m, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
            side(c);
            __builtin_amdgcn_sched_barrier(0);
            fill(0);
            acc1 = mfma_bf(a.l, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(1);
            acc = mfma_bf(a.h, b.l, acc);
            __builtin_amdgcn_sched_barrier(0);
            fill(2);
            acc1 = mfma_bf(a.m, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
            fill(3);
            acc = mfma_bf(a.h, b.m, acc);
            acc1 = mfma_bf(a.h, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        prev = acc;
        prev1 = acc1;
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int s = 0; s < 4; ++s) gstage(NO - 1, ks, s, false);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// N3 > 0: a third hidden layer of 32 N3 features (net_dims (256, 128, 64 | 128): examples/demo_A2C_PPO.py:171, :224); N2 = 4 then
// where the 16 bytes at LDS position 1024 i + lane16 of a wave's 32 rows of a sample-major image (CP chunks per part) sit in the wave's
// lane-ordered scratch block [k-step][part][64 lanes]: row r, part, chunk position cs hold the logical chunk cs ^ swz(r) = (k-step, lane half)
template <int CP>
__device__ __forceinline__ uint32_t wd_gather_off(int i, uint32_t lane16)
{
    const uint32_t o = 1024u * i + lane16, x = o / (16u * CP);      // x = 3 r + part
    const uint32_t r = (x * 171u) >> 9, part = x - 3u * r;           // (x / 3 for x < 256)
    const uint32_t c = ((o % (16u * CP)) >> 4) ^ (uint32_t)swz<CP>((int)r);
    return (((c >> 1) * 3u + part) * 64u + (c & 1u) * 32u + r) * 16u;
}

template <bool ACTOR, int KX, int N2, int N3, bool VEC>     // KX: input tiles of 32 (1: S <= 32, 2: S <= 64)
__device__ __forceinline__ void ppo_block_wd(const PpoWdArgs &args, u8 *smem, const int slab_ix)
{
    const Ppo2Args &g = args.g;
    constexpr int N1 = 8, h1 = 32 * N1, h2 = 32 * N2, h3 = 32 * N3;
    constexpr bool L3 = N3 > 0;
    constexpr int NL = L3 ? N3 : N2, hL = 32 * NL;          // the hidden layer that feeds the output layer
    static_assert(!L3 || N2 == 4, "three hidden layers: (256, 128, h3)");
    constexpr int NK1 = 2 * KX;                             // k-steps of 16 of the (zero-padded) input
    constexpr int CP1 = 4 * KX, CPQ = 8, CPH2 = 4 * N2;     // chunks per part: W1 / X images, W2 quarter / dZ1 quarter / H1 quarter images, dZ2 image
    constexpr int QB = h2 * 48 * CPQ;                       // bytes of one W2 column-quarter image
    constexpr int QPW = QB / 1024 / QNW;                    // its 1 KB pieces per wave (12 at h2 = 128, 6 at 64)
    constexpr int W1PW = h1 * 48 * CP1 / 1024 / QNW;        // W1 image pieces per wave (12 / 24)
    static_assert(QB % (1024 * QNW) == 0 && QB <= kWdSlot && h1 * 48 * CP1 <= 2 * kWdSlot, "image sizes");
    static_assert(QPW <= 4 * N2 && QPW <= 8 * NK1, "a quarter's pieces ride the k-steps of the phase before");
    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);      // wave index (scalar)
    const int wave = wave_u;
    int lane, m, hi, col, tid;                              // re-derived by refresh() at every phase (see wd_lane)
    uint32_t lane16;
    auto refresh = [&]() {
        lane = wd_lane();
        m = lane & 31;
        hi = lane >> 5;
        col = 32 * wave_u + m;                              // sample slot inside the workgroup
        tid = 64 * wave_u + lane;
        lane16 = 16u * (uint32_t)lane;
    };
    refresh();
    constexpr int net = ACTOR ? 0 : 1;
    const int S = g.S, OUT = ACTOR ? g.A : 1;
    // flat parameter block: W1 b1 W2 b2 [W3 b3] Wout bout [action_std_log]
    struct Offs {
        int64_t W1, b1, W2, b2, W3, b3, Wo, bo, sd;
    };
    const Offs d = [&]() {
        Offs o;
        o.W1 = 0; o.b1 = (int64_t)h1 * S; o.W2 = o.b1 + h1; o.b2 = o.W2 + (int64_t)h2 * h1;
        o.W3 = o.b2 + h2; o.b3 = o.W3 + (int64_t)h3 * h2;
        o.Wo = L3 ? o.b3 + h3 : o.W3; o.bo = o.Wo + (int64_t)OUT * hL; o.sd = o.bo + OUT;
        return o;
    }();
    const float *P = g.P[net];
    const float *std_log = P + d.sd;

    u8 *SLA = smem, *SLX = SLA + kWdSlot, *SLY = SLX + kWdSlot;
    float *RW3 = reinterpret_cast<float *>(SLY + kWdSlot);      // W3 copy [16][ld3] fp32 (rows >= OUT zero); later RC = dY^T [16][PLD]
    float *s_b1 = RW3 + kS3W3 / 4, *s_b2 = s_b1 + 256, *s_b3 = s_b2 + 128;
    float *s_nr = s_b3 + 16, *s_nn = s_nr + 64;
    float *s_red = s_nn + 64;
    float *s_b3h = s_red + 16;                                  // the third hidden layer's bias (128)
    constexpr int ld3 = lds_ld(128);
    float *scr0 = args.scratch + ((size_t)slab_ix * 2 + net) * wd_scratch_floats(NL, N3);
    // register tile T (0..7: GELU'(z1); 8..: H2), quad r of this thread: wave-uniform base + the thread's 16 bytes
    auto scr_tile = [&](int T, int r) -> float4 & { return reinterpret_cast<float4 *>(scr0 + (size_t)(4 * T + r) * QNT * 4)[tid]; };
    u8 *scrI = reinterpret_cast<u8 *>(scr0 + (8 + NL) * 16 * QNT);      // H1 quarter images
    u8 *scrI2 = scrI + 4 * kWdH1ImgBytes;                               // (three hidden layers) the H2 image, then the dZ3 image
    u8 *scrI3 = scrI2 + 128 * 768;

    // the wave's share of a DMA transfer: pieces [QPW wave, QPW (wave + 1)) of a quarter, [W1PW wave, ...) of the W1 image
    const uint32_t ldsXw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLX + QPW * 1024 * wave));
    const uint32_t ldsYw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLY + QPW * 1024 * wave));
    const uint32_t ldsAw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLA + W1PW * 1024 * wave));
    const u8 *w2src = g.w2img[net] + QPW * 1024 * wave_u;
    auto dma_q = [&](int q, uint32_t slot_w, int i) { wd_dma1(w2src + (size_t)q * QB + 1024 * i, lane16, slot_w + 1024u * i); };

    // ---- prologue: the sample id; the W1 image by LDS-DMA; biases, W3, normalisation constants (every load unconditional: see
    //      ppo_step_s3_impl.h)
    PROF(0);
    const int64_t bidx = (int64_t)slab_ix * PB + col;
    const bool valid = bidx < g.B;
    const int64_t id = g.ids[valid ? bidx : 0];
    const AdvNorm advn = adv_norm_consts(ACTOR ? g.adv_stats : nullptr);
    {
        const u8 *src1 = g.w1img[net] + W1PW * 1024 * wave_u;
#pragma unroll
        for (int i = 0; i < W1PW; ++i) wd_dma1(src1 + 1024 * i, lane16, ldsAw + 1024u * i);
    }
    // The prologue's loads AND stores are unconditional -- no block of it runs under a partial EXEC mask (a wave's whole mask can be
    // empty there: `tid < 64` in waves 1..3, the second W3 pass at h2 = 64).  hipcc put a live-range split of the thread index (a copy
    // into an AGPR, read back much later) at the end of such a block: lanes inactive at the copy read garbage back, and the logged sums
    // of the (8, 256, 64, 2) shape were summed over stale LDS (tests/test_ppo_wide_gpu.py caught it).  Guarded stores become stores of
    // selected values at wrapped / clamped indices (two threads may write the same value to the same place).
    constexpr int NU3 = 16 * (hL / 4) / QNT;                // float4 of the output layer's copy per thread: rows [16][hL], 1 (hL = 64) or 2
    static_assert(NU3 * QNT == 16 * (hL / 4), "output-layer copy passes");
    float4 c3[NU3];
#pragma unroll
    for (int u = 0; u < NU3; ++u) {                         // output-layer rows [16][hL] (rows >= OUT zeroed when stored)
        const int e = tid + u * QNT, i = e / (hL / 4), j4 = e - i * (hL / 4);
        c3[u] = load4<VEC>(P + d.Wo + (size_t)min(i, OUT - 1) * hL, 4 * j4, hL);
    }
    const float b3h_raw = L3 ? P[d.b3 + min(tid & 127, (L3 ? h3 : 1) - 1)] : 0.f;
    const float b1_raw = P[d.b1 + tid];
    const float b2_raw = P[d.b2 + min(tid & 127, h2 - 1)];
    const float b3_raw = P[d.bo + min(tid & 15, OUT - 1)];
    const float sd_raw = g.sd[net][min(tid & 63, S - 1)], avg_raw = g.avg[net][min(tid & 63, S - 1)];

    // ---- id -> (t = id % H, n = id // H) -> buffer row t*N + n  (AgentPPO.py:179-187) and its data
    int64_t n_, t_;
    if (g.H * g.N <= 0x7fffffffLL) {
        const uint32_t i32 = (uint32_t)id, h32 = (uint32_t)g.H, n32 = i32 / h32;
        n_ = n32;
        t_ = i32 - n32 * h32;
    } else {
        n_ = id / g.H;
        t_ = id - n_ * g.H;
    }
    const int64_t row = valid ? t_ * g.N + n_ : 0;          // padding slots read row 0 (finite data) and carry zero weight
    float4 XR[NK1][2];
    {
        const float *xrow = g.states + row * S;
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) {
            XR[ks][0] = load4<VEC>(xrow, 16 * ks + 8 * hi, S);
            XR[ks][1] = load4<VEC>(xrow, 16 * ks + 8 * hi + 4, S);
        }
    }
    const uint8_t um_raw = g.unmasks[row];
    const float um = (valid && um_raw) ? 1.f : 0.f;
    const float xa = ACTOR ? g.logprobs[row] : g.reward_sums[row];
    const float xb = ACTOR ? g.advantages[row] : 0.f;
    float act_pre[4] = {0.f, 0.f, 0.f, 0.f}, sl_pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACTOR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ac = min(4 * hi + j, OUT - 1);
            act_pre[j] = g.actions[row * OUT + ac];
            sl_pre[j] = std_log[ac];
        }
    }
#pragma unroll
    for (int u = 0; u < (kS3W3 / 16 + QNT - 1) / QNT; ++u) reinterpret_cast<float4 *>(RW3)[min(tid + u * QNT, kS3W3 / 16 - 1)] = zero4();
    s_b1[tid] = b1_raw;
    s_b2[tid & 127] = (tid & 127) < h2 ? b2_raw : 0.f;
    s_b3[tid & 15] = (tid & 15) < OUT ? b3_raw : 0.f;
    if constexpr (L3) s_b3h[tid & 127] = (tid & 127) < h3 ? b3h_raw : 0.f;
    {
        const float nr = __builtin_amdgcn_rcpf(sd_raw + 1e-4f);                  // (x - avg) / (std + 1e-4)  (AgentPPO.py:360-361)
        s_nr[tid & 63] = (tid & 63) < S ? nr : 0.f;
        s_nn[tid & 63] = (tid & 63) < S ? -(avg_raw * nr) : 0.f;
    }
    wd_wait_dma();                                                   // this wave's pieces of the W1 image (and every load above)
    lds_barrier();
    refresh();                                                   // (0a) W1 image, biases, constants visible; RW3 zeroed
    PROF_NV(1);
#pragma unroll
    for (int u = 0; u < NU3; ++u) {                                  // W3 copy [16][ld3], rows >= OUT zero; visible after (0b)
        const int e = tid + u * QNT, i = e / (hL / 4), j4 = e - i * (hL / 4);
        *reinterpret_cast<float4 *>(RW3 + i * ld3 + 4 * j4) = i < OUT ? c3[u] : zero4();
    }
    // ---- normalise the own row; its split rides behind the first output tile's MFMAs of the first layer
    Parts Xp[NK1];
    f32x16 XH[(NK1 + 1) / 2];
#pragma unroll
    for (int ks = 0; ks < NK1; ++ks) {
        const float *nr = s_nr + 16 * ks + 8 * hi, *nn = s_nn + 16 * ks + 8 * hi;
        const float4 r0 = *reinterpret_cast<const float4 *>(nr), r1 = *reinterpret_cast<const float4 *>(nr + 4);
        const float4 n0 = *reinterpret_cast<const float4 *>(nn), n1 = *reinterpret_cast<const float4 *>(nn + 4);
        f32x16 &t = XH[ks >> 1];
        const int o = 8 * (ks & 1);
        t[o + 0] = fmaf(XR[ks][0].x, r0.x, n0.x); t[o + 1] = fmaf(XR[ks][0].y, r0.y, n0.y);
        t[o + 2] = fmaf(XR[ks][0].z, r0.z, n0.z); t[o + 3] = fmaf(XR[ks][0].w, r0.w, n0.w);
        t[o + 4] = fmaf(XR[ks][1].x, r1.x, n1.x); t[o + 5] = fmaf(XR[ks][1].y, r1.y, n1.y);
        t[o + 6] = fmaf(XR[ks][1].z, r1.z, n1.z); t[o + 7] = fmaf(XR[ks][1].w, r1.w, n1.w);
    }

    // ---- first layer (W1 image in slot A [+ X]); GELU' leaves for the scratch block tile by tile; W2 quarter 0 streams into slot Y
    f32x16 H1[N1];
    {
        auto side = [&](int c) { if (c < QPW) dma_q(0, ldsYw, c); };
        constexpr int EP1 = 16 / NK1;                                 // elements of a tile finished per k-step: one or two quads
        auto done = [&](int Tp, int ks, const float (&gd)[EP1]) {
#pragma unroll
            for (int r = 0; r < EP1 / 4; ++r) scr_tile(Tp, ks * (EP1 / 4) + r) = make_float4(gd[4 * r], gd[4 * r + 1], gd[4 * r + 2], gd[4 * r + 3]);
        };
        fwd_wd<NK1, N1, CP1>(SLA, s_b1, Xp, XH, H1, m, hi, side, done);
    }
    PROF_NV(2);
    wd_wait_dma();
    lds_barrier();
    refresh();                                                   // (0b) quarter 0, W3 copy visible; every wave is done with the W1 image
    PROF_NV(3);
    {
        Parts Xs[2 * KX];
#pragma unroll
        for (int ks = 0; ks < 2 * KX; ++ks) Xs[ks] = Xp[ks];
        stage_s3<2 * KX, CP1, 0>(SLA, Xs, col, hi);                  // the input's image for dW1 (visible after the next barrier)
    }

    // ---- second layer, K split in quarters: q0 (Y) | q1 (X) | q2 (Y) | q3 (X); quarter q + 1 streams in behind quarter q's MFMAs
    f32x16 Z2[N2], H2[N2], G2[N2], H3[NL], G3[NL], Gq[2];            // (H3 / G3: three hidden layers only)
    auto load_gate = [&](int q) {                                    // GELU'(z1) tiles 2 q, 2 q + 1 back from the scratch block
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 v = scr_tile(2 * q + t, r);
                Gq[t][4 * r] = v.x; Gq[t][4 * r + 1] = v.y; Gq[t][4 * r + 2] = v.z; Gq[t][4 * r + 3] = v.w;
            }
        }
    };
#pragma unroll
    for (int To = 0; To < N2; ++To) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float4 b0 = *reinterpret_cast<const float4 *>(s_b2 + 32 * To + 16 * a + 8 * hi);
            const float4 b1 = *reinterpret_cast<const float4 *>(s_b2 + 32 * To + 16 * a + 8 * hi + 4);
            Z2[To][8 * a + 0] = b0.x; Z2[To][8 * a + 1] = b0.y; Z2[To][8 * a + 2] = b0.z; Z2[To][8 * a + 3] = b0.w;
            Z2[To][8 * a + 4] = b1.x; Z2[To][8 * a + 5] = b1.y; Z2[To][8 * a + 6] = b1.z; Z2[To][8 * a + 7] = b1.w;
        }
    }
    {
        // the split H1 operand of every k-step also leaves for the scratch block, in the layout of the dW2 operand image (stage_s3's)
        // (in the order the lanes hold it: block [wave][k-step][part] of 64 lanes x 16 bytes, one whole 1 KB line group per store --
        // stored in image order, 32 rows of 384 bytes per instruction, a quarter's twelve stores cost ~2.4k cycles; the DMA that brings
        // the image back gathers instead: every lane of a piece fetches the 16 bytes that belong at its LDS position, h1_gather below)
        auto keep_q = [&](int q) {
            return [=](int ks, const Parts &p) {
                u8 *ub = scrI + q * kWdH1ImgBytes + wave_u * (kWdH1ImgBytes / QNW) + ks * 3072;     // wave-uniform
                *reinterpret_cast<u32x4 *>(ub + lane16) = p.h;
                *reinterpret_cast<u32x4 *>(ub + 1024 + lane16) = p.m;
                *reinterpret_cast<u32x4 *>(ub + 2048 + lane16) = p.l;
            };
        };
        auto s1 = [&](int c) { if (c < QPW) dma_q(1, ldsXw, c); };
        fwd_acc_wd<0, N2, CPQ, false>(SLY, H1, Z2, m, hi, s1, keep_q(0));
        wd_wait_dma();
        lds_barrier();
        refresh();
    refresh();                                               // quarter 1 visible; every wave is done with quarter 0
        PROF_NV(4);
        auto s2 = [&](int c) { if (c < QPW) dma_q(2, ldsYw, c); };
        fwd_acc_wd<1, N2, CPQ, false>(SLX, H1, Z2, m, hi, s2, keep_q(1));
        wd_wait_dma();
        lds_barrier();
        refresh();
        PROF_NV(5);
        auto s3 = [&](int c) { if (c < QPW) dma_q(3, ldsXw, c); };
        fwd_acc_wd<2, N2, CPQ, false>(SLY, H1, Z2, m, hi, s3, keep_q(2));
        wd_wait_dma();
        lds_barrier();
        refresh();
        PROF_NV(6);
        // quarter 2 stays in Y, quarter 3 in X: the backward pass of a two-layer net starts there; with a third hidden layer the first
        // half of its weight image (rows 0 .. 63) streams into Y behind quarter 3's MFMAs
        if constexpr (L3) {
            const u8 *w3src = args.w3img[net] + 12 * 1024 * wave_u;
            const uint32_t ldsY3 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLY + 12 * 1024 * wave));
            auto s4 = [&](int c) { if (c < 12) wd_dma1(w3src + 1024 * c, lane16, ldsY3 + 1024u * c); };
            fwd_acc_wd<3, N2, CPQ, false>(SLX, H1, Z2, m, hi, s4, keep_q(3));
        } else {
            fwd_acc_wd<3, N2, CPQ, false>(SLX, H1, Z2, m, hi, NoSide(), keep_q(3));
        }
#pragma unroll
        for (int To = 0; To < N2; ++To) {
            gelu_tile(Z2[To], H2[To], G2[To]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PROF_NV(7);
    }
    refresh();
    if constexpr (L3) {
        // ---- third hidden layer: its weight image [h3][3][128 bf16] in Y (h3 = 64) or in X + Y with its halves swapped (rows 64 .. 127
        //      come into X once every wave is done with quarter 3: TROT / KROT); H2's split operand leaves for the scratch block (the
        //      dW3 operand image, lane order), GELU'(z3) stays
        wd_wait_dma();
        lds_barrier();
        refresh();
        if constexpr (N3 == 4) {
            const u8 *w3src = args.w3img[net] + 49152 + 12 * 1024 * wave_u;
            const uint32_t ldsX3 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLX + 12 * 1024 * wave));
#pragma unroll
            for (int i = 0; i < 12; ++i) wd_dma1(w3src + 1024 * i, lane16, ldsX3 + 1024u * i);
            wd_wait_dma();
            lds_barrier();
            refresh();
        }
        Parts H2p[2 * N2];
        auto done3 = [&](int Tp, int ks, const float (&gd)[2]) { G3[Tp][2 * ks] = gd[0]; G3[Tp][2 * ks + 1] = gd[1]; };
        fwd_wd<2 * N2, NL, 4 * N2, (N3 == 4 ? 2 : 0)>(N3 == 4 ? SLX : SLY, s_b3h, H2p, H2, H3, m, hi, NoSide(), done3);
        u8 *ub2 = scrI2 + wave_u * (2 * N2 * 3072);
#pragma unroll
        for (int ks = 0; ks < 2 * N2; ++ks) {
            *reinterpret_cast<u32x4 *>(ub2 + ks * 3072 + lane16) = H2p[ks].h;
            *reinterpret_cast<u32x4 *>(ub2 + ks * 3072 + 1024 + lane16) = H2p[ks].m;
            *reinterpret_cast<u32x4 *>(ub2 + ks * 3072 + 2048 + lane16) = H2p[ks].l;
        }
        refresh();
    }
    // HL / GL: the hidden layer that feeds the output layer and its GELU'
    auto &HL = [&]() -> f32x16(&)[NL] { if constexpr (L3) return H3; else return H2; }();
    auto &GL = [&]() -> f32x16(&)[NL] { if constexpr (L3) return G3; else return G2; }();
#pragma unroll
    for (int To = 0; To < NL; ++To) {
#pragma unroll
        for (int r = 0; r < 4; ++r) scr_tile(8 + To, r) = make_float4(HL[To][4 * r], HL[To][4 * r + 1], HL[To][4 * r + 2], HL[To][4 * r + 3]);
    }
    // the first gate tiles of the backward pass (GELU'(z1), features 192..255) are requested now: they arrive under the output layer
#if ERL_WD_GATE_EARLY
    load_gate(3);
#endif

    PROF_NV(8);
    refresh();
    // ---- output layer (fp32, as in ppo_step_s3_impl.h: HL[T][4 gq + j] is feature 32 T + 16 (gq >> 1) + 8 hi + 4 (gq & 1) + j)
    float Y[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACTOR) {
        f32x4 ya[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) ya[q >> 1][q & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *w3a = RW3 + (lane & 3) * ld3 + 8 * hi;
#pragma unroll
        for (int c = 0; c < 4 * NL; ++c) {
            const int T = c >> 2, gq = c & 3;
            const float4 w0 = *reinterpret_cast<const float4 *>(w3a + 32 * T + 16 * (gq >> 1) + 4 * (gq & 1));
            const float4 w1 = *reinterpret_cast<const float4 *>(w3a + 4 * ld3 + 32 * T + 16 * (gq >> 1) + 4 * (gq & 1));
            const float a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ya[0][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[j], HL[T][4 * gq + j], ya[0][j & 1], 0, 0, 0);
                ya[1][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], HL[T][4 * gq + j], ya[1][j & 1], 0, 0, 0);
            }
        }
        const float4 b4 = *reinterpret_cast<const float4 *>(s_b3 + 4 * hi);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = ya[0][0][j] + ya[0][1][j], hi_ = ya[1][0][j] + ya[1][1][j];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi_), false, false);
            Y[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]) + bb[j];   // lanes < 32: output j; lanes >= 32: output 4 + j
        }
    } else {
        f32x2 yp = {0.f, 0.f}, yq = {0.f, 0.f};
        const float *w3 = RW3 + 8 * hi;
#pragma unroll
        for (int c = 0; c < 4 * NL; ++c) {
            const int T = c >> 2, gq = c & 3;
            const float4 wv = *reinterpret_cast<const float4 *>(w3 + 32 * T + 16 * (gq >> 1) + 4 * (gq & 1));
            yp = f32x2{wv.x, wv.y} * f32x2{HL[T][4 * gq + 0], HL[T][4 * gq + 1]} + yp;
            yq = f32x2{wv.z, wv.w} * f32x2{HL[T][4 * gq + 2], HL[T][4 * gq + 3]} + yq;
        }
        const float s = (yp.x + yp.y) + (yq.x + yq.y);
        Y[0] = s + __shfl_xor(s, 32, 64) + s_b3[0];
    }

    refresh();
    // ---- objective and dL/dY for this lane's outputs a = 4 hi + j   (AgentPPO.py:189-204)
    float dY[4] = {0.f, 0.f, 0.f, 0.f};
    float loss0 = 0.f, loss1 = 0.f;
    float dsl[4] = {0.f, 0.f, 0.f, 0.f};
    if (!ACTOR) {
        const float diff = Y[0] - xa;
        const bool head = hi == 0;
        loss0 = head ? diff * diff * um : 0.f;
        dY[0] = head ? 2.f * diff * um * g.inv_batch : 0.f;
    } else {
        float diffv[4], ivar[4];
        float lp = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = 4 * hi + j;
            const float sl = sl_pre[j];
            const float diff = act_pre[j] - Y[j];
            const bool on = a < OUT;
            ivar[j] = __expf(-2.f * sl);
            diffv[j] = on ? diff : 0.f;
            const float term = -(diff * diff) * (0.5f * ivar[j]) - sl - kLogSqrt2PiF;
            lp += on ? term : 0.f;
        }
        lp += __shfl_xor(lp, 32, 64);
        const PpoActorTerms o = ppo_actor_terms(g.objective, adv_normalized(xb, advn), lp, xa, g.ratio_clip, g.lambda_entropy, um, OUT, true);
        if (hi == 0) {
            loss0 = valid ? o.logged : 0.f;
            loss1 = valid ? o.ent_mask : 0.f;
        }
        const float dlp = (valid ? o.dlp : 0.f) * g.inv_batch;
        const float ent_term = (valid ? o.ent_w : 0.f) * g.inv_batch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = 4 * hi + j < OUT;
            dY[j] = on ? dlp * (diffv[j] * ivar[j]) : 0.f;
            dsl[j] = on ? dlp * (diffv[j] * diffv[j] * ivar[j] - 1.f) + ent_term : 0.f;
        }
    }

    refresh();
    // ---- dZ2 = (W3^T dY) * GELU'(z2)  (K = 8 outputs: four fp32 k-pairs; A-row m carries feature 32 To + phi(m))
    {
        const int pm = phi(m);
        float w3[NL][4];
#pragma unroll
        for (int To = 0; To < NL; ++To) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w3[To][j] = RW3[(4 * hi + j) * ld3 + 32 * To + pm];
        }
#pragma unroll
        for (int To = 0; To < NL; ++To) {
            f32x16 acc = {0};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma32(w3[To][j], dY[j], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) GL[To][r] *= acc[r];
        }
    }

    PROF_NV(9);
    // ---- dZ1 = (W2^T dZ2) * GELU'(z1), quarter by quarter (3, 2 resident; 1, 0 streamed back), each contracted with the input
    //      (dW1, db1) as soon as it is staged
    float *slab = g.slabs + (size_t)slab_ix * g.stride + (ACTOR ? 0 : g.Pa);
    float *RC = RW3;
    Parts dZ2p[2 * N2], dZ1q[4];
    // the gate tiles requested a phase ago are IN their registers as far as the compiler is concerned (called right after a
    // s_waitcnt vmcnt(0)): a compiler-placed wait for them inside the next quarter would also wait for every LDS-DMA piece issued
    // there before it (memory operations return in order) -- 7k cycles per wave in the first version of this kernel
    auto gate_arrived = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(Gq[t][e]));
        }
    };
    auto dw1 = [&](int q, const u8 *slot) {
        const int it = wave & 1, jc = wave >> 1;             // row tile of the quarter's two, column tile of the input's KX
        if (jc < KX) {
            Parts A[8];
            grad_a_load<CPQ>(slot, it, A, lane);
            grad_tiles<CP1, 1, 2>(A, SLA, it, jc, 0, slab + d.W1 + (size_t)(64 * q) * S, S, S, lane);
            if (jc == 0) grad_bias(A, slab + d.b1 + 64 * q, it, lane);
        }
    };
    if constexpr (L3) {
        // ---- back through the third layer: dZ2 = (W3^T dZ3) * GELU'(z2) leaves split (dZ2p); dZ3's split operand goes to the scratch
        //      block (the dW3 operand image, lane order); then W2's quarters come back: 2 into Y, 3 into X where W3 overwrote it
        Parts dZ3p[2 * NL];
        refresh();
        bwd_wd<2 * NL, N2, 4 * N2, false, (N3 == 4 ? 4 : 0)>(N3 == 4 ? SLX : SLY, dZ3p, G3, G2, dZ2p, lane, NoSide());
        u8 *ub3 = scrI3 + wave_u * (2 * NL * 3072);
#pragma unroll
        for (int ks = 0; ks < 2 * NL; ++ks) {
            *reinterpret_cast<u32x4 *>(ub3 + ks * 3072 + lane16) = dZ3p[ks].h;
            *reinterpret_cast<u32x4 *>(ub3 + ks * 3072 + 1024 + lane16) = dZ3p[ks].m;
            *reinterpret_cast<u32x4 *>(ub3 + ks * 3072 + 2048 + lane16) = dZ3p[ks].l;
        }
        lds_barrier();
        refresh();                                               // every wave is done with the W3 image
        if constexpr (N3 == 4) {
#pragma unroll
            for (int i = 0; i < QPW; ++i) dma_q(3, ldsXw, i);
        }
#pragma unroll
        for (int i = 0; i < QPW; ++i) dma_q(2, ldsYw, i);
        wd_wait_dma();
        gate_arrived();
        lds_barrier();
        refresh();
    }
    refresh();
#if !ERL_WD_GATE_EARLY
    load_gate(3);
#endif
    bwd_wd<2 * N2, 2, CPQ, L3>(SLX, dZ2p, G2, Gq, dZ1q, lane, NoSide());        // (two hidden layers: splits dZ2 into dZ2p on the way)
    PROF_NV(10);
    lds_barrier();
    refresh();                                                   // (1) every wave is done with quarter 3 (X) and with W3
    stage_s3<4, CPQ, 0>(SLX, dZ1q, col, hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        RC[(4 * hi + j) * PLD + col] = dY[j];
        RC[(8 + 4 * hi + j) * PLD + col] = dsl[j];       // rows 8..15: per-sample dL/dstd_log (zero for the critic)
    }
    load_gate(2);
    lds_barrier();
    refresh();                                                   // (2) dZ1 quarter 3 (and, long since, the X image) visible
    PROF_NV(11);
    dw1(3, SLX);
    refresh();
    PROF_NV(12);
    bwd_wd<2 * N2, 2, CPQ, true>(SLY, dZ2p, G2, Gq, dZ1q, lane, NoSide());
    PROF_NV(13);
    lds_barrier();
    refresh();                                                   // (3) quarter 2 (Y) and the dZ1 image in X consumed
    stage_s3<4, CPQ, 0>(SLY, dZ1q, col, hi);
#pragma unroll
    for (int i = 0; i < QPW; ++i) dma_q(1, ldsXw, i);                // quarter 1 comes back into X
    load_gate(1);
    lds_barrier();
    refresh();                                                   // (4)
    PROF_NV(14);
    dw1(2, SLY);
    wd_wait_dma();
    gate_arrived();
    lds_barrier();
    refresh();                                                   // (5) quarter 1 visible; the dZ1 image in Y consumed
    PROF_NV(15);
    {
        auto s0 = [&](int c) { if (c < QPW) dma_q(0, ldsYw, c); };   // quarter 0 comes back into Y behind quarter 1's MFMAs
        bwd_wd<2 * N2, 2, CPQ, true>(SLX, dZ2p, G2, Gq, dZ1q, lane, s0);
    }
    PROF_NV(16);
    lds_barrier();
    refresh();                                                   // (6)
    stage_s3<4, CPQ, 0>(SLX, dZ1q, col, hi);
    load_gate(0);
    lds_barrier();
    refresh();                                                   // (7)
    PROF_NV(17);
    dw1(1, SLX);
    wd_wait_dma();
    gate_arrived();
    lds_barrier();
    refresh();                                                   // (8) quarter 0 visible; the dZ1 image in X consumed
    PROF_NV(18);
    bwd_wd<2 * N2, 2, CPQ, true>(SLY, dZ2p, G2, Gq, dZ1q, lane, NoSide());
    PROF_NV(19);
    lds_barrier();
    refresh();                                                   // (9)
    stage_s3<4, CPQ, 0>(SLY, dZ1q, col, hi);
    lds_barrier();
    refresh();                                                   // (10)
    PROF_NV(20);
    float4 h2v[NL][4];                                               // the last hidden layer comes back under dW1's last quarter
#pragma unroll
    for (int t = 0; t < NL; ++t) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) h2v[t][r4] = scr_tile(8 + t, r4);
    }
    dw1(0, SLY);
    lds_barrier();
    refresh();                                                   // (11) X, Y, the X image in A consumed
    PROF_NV(21);
    // H1's quarter images come back from the scratch block by LDS-DMA (every wave its share of the 1 KB pieces): quarter 0 into A now
    // (the X image was consumed before (11)), 1 and 2 into X and Y once the dZ2 image is in registers, 3 into A behind quarter 0
    constexpr int HPW = kWdH1ImgBytes / 1024 / QNW;                  // pieces per wave (12)
    static_assert(HPW * 1024 == 32 * 48 * CPQ, "a wave's pieces are exactly its own 32 sample rows");
    const u8 *isrc = scrI + HPW * 1024 * wave_u;
    const uint32_t ldsAh = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLA + HPW * 1024 * wave));
    const uint32_t ldsXh = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLX + HPW * 1024 * wave));
    const uint32_t ldsYh = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLY + HPW * 1024 * wave));
    // where the 16 bytes at LDS position 1024 i + 16 lane of this wave's 32 image rows sit in the wave's scratch block (keep_q's order):
    // row r, part, chunk position cs hold the logical chunk cs ^ swz(r) = (k-step, lane half) of sample r
    uint32_t h1_gather[kWdH1ImgBytes / 1024 / QNW];
#pragma unroll
    for (int i = 0; i < kWdH1ImgBytes / 1024 / QNW; ++i) h1_gather[i] = wd_gather_off<CPQ>(i, lane16);
    auto dma_h1 = [&](int q, uint32_t slot_w) {
#pragma unroll
        for (int i = 0; i < HPW; ++i) wd_dma1(isrc + (size_t)q * kWdH1ImgBytes, h1_gather[i], slot_w + 1024u * i);
    };
    // (H2, requested before dW1's last quarter, is in its registers as far as the compiler is concerned before the pieces go out:
    // see gate_arrived)
    wd_wait_dma();
#pragma unroll
    for (int t = 0; t < NL; ++t) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) asm volatile("" : "+v"(h2v[t][r4].x), "+v"(h2v[t][r4].y), "+v"(h2v[t][r4].z), "+v"(h2v[t][r4].w));
    }
    dma_h1(0, ldsAh);                                                // (A is free: the X image was consumed before (11))

    // ---- output layer: dW3 (16 x h2) = dY^T . H2 on 16x16x4 fp32 MFMA (H2 back from the scratch block, staged feature-major in X + Y)
    {
        float *T2 = reinterpret_cast<float *>(SLX);
#pragma unroll
        for (int t = 0; t < NL; ++t) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = h2v[t][r4];
                const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * r4 + j;
                    T2[(32 * t + 16 * (r >> 3) + 8 * hi + (r & 7)) * PLD + col] = hv[j];
                }
            }
        }
    }
    lds_barrier();
    refresh();                                                   // (12)
    PROF_NV(22);
    {
        const float *T2 = reinterpret_cast<const float *>(SLX);
        const int l15 = lane & 15, q = lane >> 4;
        f32x2 hs = {0.f, 0.f};
#pragma unroll
        for (int rep = 0; rep < (2 * NL + QNW - 1) / QNW; ++rep) {
            const int it = wave + QNW * rep;                            // 16-column tile of dW3 (wave-uniform)
            if (it >= 2 * NL) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *a = RC + l15 * PLD + 4 * q;
            const float *b = T2 + (16 * it + l15) * PLD + 4 * q;
#pragma unroll
            for (int j = 0; j < PB / 16; ++j) {
                const float4 av = *reinterpret_cast<const float4 *>(a + 16 * j), bv = *reinterpret_cast<const float4 *>(b + 16 * j);
                acc = mfma16(av.x, bv.x, acc);
                acc = mfma16(av.y, bv.y, acc);
                acc = mfma16(av.z, bv.z, acc);
                acc = mfma16(av.w, bv.w, acc);
                if (rep == 0) {
                    hs += f32x2{av.x, av.y};
                    hs += f32x2{av.z, av.w};
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a_ = 4 * q + r;
                if (a_ < OUT) slab_store(acc[r], slab + d.Wo + (size_t)a_ * hL + 16 * it + l15);
            }
        }
        float s = hs.x + hs.y;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (wave == 0 && q == 0) {
            if (l15 < OUT) slab[d.bo + l15] = s;
            else if (ACTOR && l15 >= 8 && l15 - 8 < OUT) slab[d.sd + l15 - 8] = s;
        }
    }
    lds_barrier();
    refresh();                                                   // (13) H2^T consumed
    PROF_NV(23);
    stage_s3<2 * N2, CPH2, 0>(SLX, dZ2p, col, hi);                   // the dZ2 image spans X + Y
    wd_wait_dma();
    lds_barrier();
    refresh();                                                   // (14)
    PROF_NV(24);

    // ---- layer 2: dW2 = dZ2^T . H1, db2  (wave w: row tile w % N2 of dZ2^T, read once into registers; H1 passes by in quarters)
    {
        constexpr int CS = 4 / N2, NBW = 2 / CS;
        const int it = wave % N2, jc = wave / N2;
        Parts A[8];
        grad_a_load<CPH2>(SLX, it, A, lane);
        lds_barrier();
        refresh();
    refresh();                                               // (15) the dZ2 image is in registers: X, Y are free
        PROF_NV(25);
        dma_h1(1, ldsXh);
        dma_h1(2, ldsYh);
        grad_tiles<CPQ, NBW, CS>(A, SLA, it, jc, 0, slab + d.W2, h1, h1, lane);
        wd_wait_dma();
        lds_barrier();
        refresh();
    refresh();                                               // (16) quarters 1, 2 visible; quarter 0 (A) consumed
        PROF_NV(26);
        dma_h1(3, ldsAh);
        grad_tiles<CPQ, NBW, CS>(A, SLX, it, jc, 2, slab + d.W2, h1, h1, lane);
        grad_tiles<CPQ, NBW, CS>(A, SLY, it, jc, 4, slab + d.W2, h1, h1, lane);
        wd_wait_dma();
        lds_barrier();
        refresh();
    refresh();                                               // (17)
        PROF_NV(27);
        grad_tiles<CPQ, NBW, CS>(A, SLA, it, jc, 6, slab + d.W2, h1, h1, lane);
        if (jc == 0) grad_bias(A, slab + d.b2, it, lane);
    }
    PROF(28);
    if constexpr (L3) {
        // ---- layer 3: dW3 = dZ3^T . H2, db3 -- both operand images come back from the scratch block by gathering LDS-DMA: H2
        //      ([128][3][128], 96 KB) into X + Y, dZ3 in halves of 64 features ([128][3][64]) into A
        lds_barrier();
        refresh();                                               // every wave is done with dW2's last quarter (A)
        constexpr int P2W = 32 * 48 * 16 / 1024;                 // H2 image pieces per wave (24); a dZ3 half: HPW (12)
        const u8 *s2 = scrI2 + wave_u * (P2W * 1024);
        const uint32_t lX = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(SLX + P2W * 1024 * wave));
#pragma unroll
        for (int i = 0; i < P2W; ++i) wd_dma1(s2, wd_gather_off<16>(i, lane16), lX + 1024u * i);
        const int it = wave & 1, jc = wave >> 1;                 // row tile of the half's two; column tiles jc, jc + 2 of H2's four
#pragma unroll
        for (int hf = 0; hf < N3 / 2; ++hf) {
            const u8 *s3 = scrI3 + wave_u * (2 * N3 * 3072) + hf * (4 * 3072);
#pragma unroll
            for (int i = 0; i < HPW; ++i) wd_dma1(s3, h1_gather[i], ldsAh + 1024u * i);     // (the same gather as an H1 quarter: CP = 8)
            wd_wait_dma();
            lds_barrier();
            refresh();
            Parts A[8];
            grad_a_load<CPQ>(SLA, it, A, lane);
            grad_tiles<16, 2, 2>(A, SLX, it, jc, 0, slab + d.W3 + (size_t)(64 * hf) * h2, h2, h2, lane);
            if (jc == 0) grad_bias(A, slab + d.b3 + 64 * hf, it, lane);
            if (hf + 1 < N3 / 2) {
                lds_barrier();
                refresh();                                       // the first half of the dZ3 image is consumed
            }
        }
    }

    // ---- objective partial sums (scaled by 1/B so that the slab reduction yields the means)
    // (sums over the workgroup with the kernel's own wave / lane indices)
    auto wg_sum = [&](float v, float *red) {
        v = wave_sum(v);
        lds_barrier();
        if (lane == 0) red[wave_u] = v;
        lds_barrier();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < QNW; ++w) t += red[w];
        return t;
    };
    refresh();
    const float t0 = wg_sum(loss0, s_red);
    const float t1 = wg_sum(loss1, s_red + 8);
    if (tid == 0) {
        float *logs = g.slabs + (size_t)slab_ix * g.stride + g.Pa + g.Pc;
        if (ACTOR) {
            float ent = 0.f;
            for (int a = 0; a < OUT; ++a) ent += 1.4189385332046727418f + logf(expf(std_log[a]));
            logs[1] = t0 * g.inv_batch;
            logs[2] = ent * t1 * g.inv_batch;
        } else {
            logs[0] = t0 * g.inv_batch;
            logs[3] = 0.f;
            for (int64_t e = g.Pa + g.Pc + 4; e < g.stride; ++e) logs[e - (g.Pa + g.Pc)] = 0.f;
        }
    }
}

template <int KX, int N2, int N3, bool VEC>
__global__ __launch_bounds__(QNT) void ppo_step_wd_kernel(PpoWdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_wd[];
    const SpanT t_span = span_enter(a.g);
    const K6Wg wg = k6_wg_map(a.g);
    if (wg.actor) ppo_block_wd<true, KX, N2, N3, VEC>(a, smem_wd, wg.slab);
    else ppo_block_wd<false, KX, N2, N3, VEC>(a, smem_wd, wg.slab);
    span_exit(a.g, t_span);
}

template <int KX, int N2, int N3, bool VEC>
int launch_wd(const PpoWdArgs &a, int n_slabs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)ppo_step_wd_kernel<KX, N2, N3, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)kWdLdsBytes),
                                "hipFuncSetAttribute(ppo_step_wd_kernel)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((ppo_step_wd_kernel<KX, N2, N3, VEC>), dim3(n_slabs, 2), dim3(QNT), kWdLdsBytes, stream, a);
    return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32 (wide)");
}

}  // namespace
