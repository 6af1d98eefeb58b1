// K6 on the bf16 matrix pipe with fp32-equivalent arithmetic: every fp32 operand x of the five big products of a PPO minibatch
// (two forward layers, the backward through W2, dW1, dW2) is split into three bf16 parts x = h + m + l (round-to-nearest each,
// h + m + l == x exactly), and a product a . b is accumulated in fp32 from the six partial products whose weight is >= 2^-16
//     ah bh + (ah bm + am bh) + (ah bl + al bh + am bm)
// on v_mfma_f32_32x32x16_bf16 (every bf16 x bf16 product is exact in fp32; the three dropped terms are <= 2^-23 |a b|, one or
// two fp32 roundings' worth -- measured against fp64 the result is as close as v_mfma_f32_32x32x2_f32's:
// tools/split_mfma_probe.hip, profiles/r03_split_mfma_probe.txt).  Six bf16 MFMAs of K = 16 (192 cycles) replace eight fp32 MFMAs
// of K = 2 (512 cycles), and the bf16 matrix pipe runs beside the vector ALUs instead of on them.
//
// Same contract as ppo_step_w4_impl.h (AgentPPO.update_objectives up to the optimizer steps, elegantrl/agents/AgentPPO.py:173-204;
// ActorPPO.get_logprob_entropy :378-386; same slabs, same logged sums), same mapping (grid (ceil(B / 128), 2), four waves, one
// per SIMD, 32 samples each, every activation of a sample in its lane's registers).  What differs:
//   * result-row permutation: A-row i of every forward / backward tile carries feature phi(i) = i with bits 2 and 3 swapped, so
//     that lane (sample, hi) ends up with features 16 a + 8 hi + 0..7 (a = 0, 1) of a 32-feature tile in acc[8 a .. 8 a + 7]: the
//     B operand of the next layer's k-step (tile, a) is those eight values in natural order (four v_cvt_pk_bf16_f32 per part), its
//     A operand one 16-byte LDS read per part, and a staging store 16 bytes per part;
//   * weights live in LDS as images [row][3 parts][K bf16], rows 6 K bytes, no padding: 16-byte chunks are XOR-swizzled with a
//     function of the row index (swz<>) chosen so that the three access patterns are bank-conflict free at once -- 16-byte reads
//     by the lane groups of ds_read_b128, the [4 rows][64 bytes] blocks of ds_read_b64_tr_b16 (the backward's W2^T operand and
//     both operands of the weight gradients are transposing reads), and 16-byte stores by 8 consecutive rows;
//   * the weight gradients contract over samples, which sit in lanes: both operands go through LDS as sample-major images
//     S[sample][part][feature] (16-byte stores straight from the split registers) and come back transposed.  A wave's A operand
//     (its 32 features of dZ^T, all 128 samples) is read once into 96 registers; the bias gradient is three more MFMAs per k-step
//     against a constant ones operand.
//   * the 8-row output layer, dZ2 = W3^T dY and dW3 stay on the fp32 instructions of ppo_step_w4_impl.h (K <= 8 or 16 rows).
#pragma once
#include "ppo_step_w4_impl.h"
#include "split_bf16.h"
#include "s3_image.h"
#include <type_traits>

namespace {

__device__ __forceinline__ int phi(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }   // bits 2 and 3 swapped

template <int CP>
__device__ __forceinline__ u8 *img_at(u8 *img, int row, int part, int chunk)
{
    return img + row * (48 * CP) + part * (16 * CP) + 16 * (chunk ^ swz<CP>(row));
}

// a float4 (columns 4 j4 .. 4 j4 + 3 of `row`) into an image: three 8-byte stores
template <int CP>
__device__ __forceinline__ void img_store4(u8 *img, int row, int j4, float4 v)
{
    uint32_t h0, m0, l0, h1, m1, l1;
    split2(v.x, v.y, h0, m0, l0);
    split2(v.z, v.w, h1, m1, l1);
    u8 *p = img_at<CP>(img, row, 0, j4 >> 1) + 8 * (j4 & 1);
    *reinterpret_cast<u32x2 *>(p) = u32x2{h0, h1};
    *reinterpret_cast<u32x2 *>(p + 16 * CP) = u32x2{m0, m1};
    *reinterpret_cast<u32x2 *>(p + 32 * CP) = u32x2{l0, l1};
}

// the registers of copy_load (row-major [rows_pad][4 * CP * 2 ... ] see mlp_chain.h) into an image with K = 8 CP columns
template <int MAXV, int CP>
__device__ __forceinline__ void img_store(const float4 (&v)[MAXV], u8 *img, int rows_pad, int tid)
{
    constexpr int vpr = 2 * CP;                       // float4 per row
    const int total = rows_pad * vpr;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int e = tid + u * QNT;
        if (e < total) img_store4<CP>(img, e / vpr, e % vpr, v[u]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward layer: tile To of the result (A-row i <-> feature 32 To + phi(i)) = GELU(bias + W . in), W an image with CP chunks
// per part.  Leaves H (fp32) and GELU'.  The input comes as fp32 tiles (inH) and is split on the way -- k-step ks + 1's operand
// behind the MFMAs of k-step ks of tile 0 -- into inP, which the caller keeps.
//
// The bf16 matrix pipe runs beside the vector ALUs (32 cycles per MFMA, room for ~6 other instructions each), and a wave issues
// in order: the vector work is cut into six stages per k-step, one behind each MFMA, pinned there by scheduling fences (left to
// itself hipcc bunches it): the operand split under tile 0, the GELU of tile To - 1 under tile To.  Consecutive MFMAs alternate
// between two accumulators: an instruction between two MFMAs on the SAME accumulator costs ~43 cycles (the back-to-back
// forwarding path is lost).
// ---------------------------------------------------------------------------------------------------------
struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

// `side(c)` is called once per k-step c = To NK + ks, between its first MFMA and its first vector stage (the caller's LDS-DMA
// pieces ride there: a vector-memory instruction among MFMAs costs its issue slot, a burst of them 60-180 cycles apiece)
template <int NK, int NO, int CP, typename Side = NoSide>
__device__ __forceinline__ void fwd_s3(const u8 *img, const float *bias, Parts (&inP)[NK], const f32x16 (&inH)[(NK + 1) / 2],
                                       f32x16 (&outH)[NO], f32x16 (&outG)[NO], int m, int hi, const Side &side = Side())
{
    constexpr int ROWB = 48 * CP, PBY = 16 * CP, NC = NO * NK;
    constexpr int EP = 16 / NK;                          // elements of the previous tile finished per k-step (NK in {1, 2, 4, 8})
    static_assert(EP * NK == 16 && EP >= 2, "k-steps per tile");
    const int row = phi(m);
    const u8 *base = img + row * ROWB;
    const int x16 = 16 * (swz<CP>(row) ^ hi);            // chunk 2 ks + hi, swizzled: (32 ks) ^ x16
    Parts aq[2];
    auto issue = [&](int c, Parts &dst) {
        const int To = c / NK, ks = c % NK;
        const u8 *p = base + 32 * To * ROWB + ((32 * ks) ^ x16);
        dst.h = *reinterpret_cast<const u32x4 *>(p);
        dst.m = *reinterpret_cast<const u32x4 *>(p + PBY);
        dst.l = *reinterpret_cast<const u32x4 *>(p + 2 * PBY);
    };
    f32x16 prev, prev1;
    // exact-erf GELU (Abramowitz-Stegun 7.1.26 as in gelu2: exp(-z^2 / 2) as one v_exp_f32, half-scaled coefficients) and its
    // derivative for elements EP ks .. EP ks + EP - 1 of tile Tp, stage s of 6 (3 instructions per element and stage).  Scalar
    // fp32 instructions: beside bf16 MFMAs a packed-fp32 instruction costs more than the two it replaces.
    constexpr float kC = 0.84932180028801904272f;              // sqrt(log2(e) / 2)
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
                pp[i] = fmaf(-pp[i], uu[i], 0.5f);                  // 0.5 erf(|z| / sqrt 2)
            } else if (s == 4) {
                pp[i] = copysignf(pp[i], z[i]) + 0.5f;              // the normal cdf
                uu[i] = z[i] * uu[i];
            } else {
                float y = z[i] * pp[i];
                float gd = fmaf(uu[i], 0.39894228040143267794f, pp[i]);
                asm volatile("" : "+v"(y), "+v"(gd));      // (LLVM sinks pure arithmetic towards its first use: see ERL_PIN4)
                outG[Tp][e] = gd;
                outH[Tp][e] = y;
            }
        }
        if (fence) __builtin_amdgcn_sched_barrier(0);
    };
    // pair s (of 4) of the operand of k-step ks from the fp32 input
    auto jit = [&](int ks, int s) {
        if (ks < NK && s < 4) {
            uint32_t h, mm, l;
            split2(inH[ks >> 1][8 * (ks & 1) + 2 * s], inH[ks >> 1][8 * (ks & 1) + 2 * s + 1], h, mm, l);
            asm volatile("" : "+v"(h), "+v"(mm), "+v"(l));
            inP[ks].h[s] = h; inP[ks].m[s] = mm; inP[ks].l[s] = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the same split of k-step ks's four pairs spread over ALL SIX gaps, independent pairs side by side (round 6: one pair per gap was a
    // chain of eleven dependent instructions behind each of four MFMAs and nothing behind the other two; second layer -0.3k cycles at
    // [128,128], -0.37k at [128,64].  The GELU cut the same way -- four elements over two k-steps, twelve pieces -- measured +0.24k and is
    // not kept: profiles/r06_k6_slice_ab.txt)
    float ja[4], jb[4];
    auto jit6 = [&](int ks, int gp) {
        if (ks < NK) {
            const f32x16 &t = inH[ks >> 1];
            const int o = 8 * (ks & 1);
            Parts &q = inP[ks];
            auto P0 = [&](int j) { uint32_t h = pk_bf16(t[o + 2 * j], t[o + 2 * j + 1]); asm volatile("" : "+v"(h)); q.h[j] = h; };
            auto P1 = [&](int j) { ja[j] = sub_bf_lo(t[o + 2 * j], q.h[j]); jb[j] = sub_bf_hi(t[o + 2 * j + 1], q.h[j]); asm volatile("" : "+v"(ja[j]), "+v"(jb[j])); };
            auto P2 = [&](int j) { uint32_t mm = pk_bf16(ja[j], jb[j]); asm volatile("" : "+v"(mm)); q.m[j] = mm; };
            auto P3 = [&](int j) { ja[j] = sub_bf_lo(ja[j], q.m[j]); jb[j] = sub_bf_hi(jb[j], q.m[j]); asm volatile("" : "+v"(ja[j]), "+v"(jb[j])); };
            auto P4 = [&](int j) { uint32_t l = pk_bf16(ja[j], jb[j]); asm volatile("" : "+v"(l)); q.l[j] = l; };
            if (gp == 0) { P0(0); P0(1); P0(2); P0(3); }
            else if (gp == 1) { P1(0); P1(1); }
            else if (gp == 2) { P1(2); P1(3); }
            else if (gp == 3) { P2(0); P2(1); P2(2); P2(3); P3(0); }
            else if (gp == 4) { P3(1); P3(2); }
            else { P3(3); P4(0); P4(1); P4(2); P4(3); }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) jit(0, s);
    issue(0, aq[0]);
    if constexpr (NK == 1) {
        // one k-step per tile (S <= 8): six MFMAs cannot carry sixteen GELUs -- the epilogue runs on its own, on packed fp32
        // (gelu_tile of ppo_step_w4_impl.h: 12.5 instructions per element instead of 17)
#pragma unroll
        for (int To = 0; To < NO; ++To) {
            f32x16 acc;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float4 b0 = *reinterpret_cast<const float4 *>(bias + 32 * To + 16 * a + 8 * hi);
                const float4 b1 = *reinterpret_cast<const float4 *>(bias + 32 * To + 16 * a + 8 * hi + 4);
                acc[8 * a + 0] = b0.x; acc[8 * a + 1] = b0.y; acc[8 * a + 2] = b0.z; acc[8 * a + 3] = b0.w;
                acc[8 * a + 4] = b1.x; acc[8 * a + 5] = b1.y; acc[8 * a + 6] = b1.z; acc[8 * a + 7] = b1.w;
            }
            if (To + 1 < NO) issue(To + 1, aq[(To + 1) & 1]);
            mma6(aq[To & 1], inP[0], acc);
            __builtin_amdgcn_sched_barrier(0);
            gelu_tile(acc, outH[To], outG[To]);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // the accumulator starts from the bias, read a tile ahead (acc[8 a + e] <-> feature 32 To + 16 a + 8 hi + e)
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
                else if constexpr ((ERL_K6_SLICE & 1) != 0) jit6(ks + 1, s);
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
#if defined(ERL_PROFILE) && defined(ERL_PROFILE_FINE)
    side(NC);                                            // (fine phase stamps: the MFMA loop is over)
#endif
    // the last tile has no MFMAs to ride behind: unfenced, so that all 16 elements' chains interleave (two dependent chains
    // alone stall on each other: tools/split_mfma_probe.hip, part E)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int s = 0; s < 6; ++s) stage(NO - 1, ks, s, false);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// backward through a layer's input: dzin[To] = gate[To] * (W^T . dz), W the image of the layer (rows = its outputs, CP chunks
// per part = its inputs / 8).  dz comes as fp32 tiles (dzH) and is split on the way into dzP (kept by the caller) behind tile
// 0's MFMAs; behind tile To's MFMAs the gate product of tile To - 1 is formed and split (outP), so that the weight-gradient
// staging finds registers to store.  The A operand W^T comes through transposing reads: lane (q = lane >> 4: kb = q >> 1,
// half = q & 1; t = lane & 15: rr = t >> 2, u = t & 3) addresses row 16 ks + 8 kb + 4 ridx + rr, columns 32 To + 16 half +
// 4 sigma(u) .. + 3 (sigma swaps 1 and 2), and receives rows .. + 0..3 of column 32 To + phi(lane & 31): the result rows are in
// the gate's order.
// ---------------------------------------------------------------------------------------------------------
template <int NK, int NO, int CP, typename Side = NoSide>
__device__ __forceinline__ void bwd_s3(const u8 *img, Parts (&dzP)[NK], const f32x16 (&dzH)[NK / 2], const f32x16 (&gate)[NO],
                                       Parts (&outP)[2 * NO], int lane, const Side &side = Side())
{
    constexpr int ROWB = 48 * CP, PBY = 16 * CP, NC = NO * NK;
    constexpr int EP = 16 / NK;
    static_assert(EP * NK == 16 && EP % 2 == 0, "k-steps per tile");
    constexpr int NPR = EP / 2;                               // element pairs of the previous tile finished per k-step
    const int q = lane >> 4, kb = q >> 1, half = q & 1, t = lane & 15, rr = t >> 2, u = t & 3;
    const int su = ((u & 1) << 1) | (u >> 1);                 // sigma(u)
    const int ccl = 2 * half + (su >> 1), sub = 8 * (su & 1);
    const int rl0 = 8 * kb + rr, rl1 = rl0 + 4;               // row within the 16 of a k-step, ridx = 0 / 1
    const u8 *b0 = img + rl0 * ROWB + sub, *b1 = img + rl1 * ROWB + sub;
    const int x0 = 16 * (ccl ^ swz<CP>(rl0)), x1 = 16 * (ccl ^ swz<CP>(rl1));
    u32x2 rq[2][6];
    auto issue = [&](int c, u32x2(&dst)[6]) {
        const int To = c / NK, ks = c % NK;
        const u8 *p0 = b0 + 16 * ks * ROWB + ((64 * To) ^ x0), *p1 = b1 + 16 * ks * ROWB + ((64 * To) ^ x1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            dst[2 * pl] = lds_tr(p0 + pl * PBY);
            dst[2 * pl + 1] = lds_tr(p1 + pl * PBY);
        }
    };
    auto jit = [&](int ks, int s) {
        if (ks < NK && s < 4) {
            uint32_t h, mm, l;
            split2(dzH[ks >> 1][8 * (ks & 1) + 2 * s], dzH[ks >> 1][8 * (ks & 1) + 2 * s + 1], h, mm, l);
            asm volatile("" : "+v"(h), "+v"(mm), "+v"(l));
            dzP[ks].h[s] = h; dzP[ks].m[s] = mm; dzP[ks].l[s] = l;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    float ja[4], jb[4];
    auto jit6 = [&](int ks, int gp) {                         // (fwd_s3's: the four pairs of a k-step spread over the gaps, side by side)
        if (ks < NK) {
            const f32x16 &t = dzH[ks >> 1];
            const int o = 8 * (ks & 1);
            Parts &q = dzP[ks];
            auto P0 = [&](int j) { uint32_t h = pk_bf16(t[o + 2 * j], t[o + 2 * j + 1]); asm volatile("" : "+v"(h)); q.h[j] = h; };
            auto P1 = [&](int j) { ja[j] = sub_bf_lo(t[o + 2 * j], q.h[j]); jb[j] = sub_bf_hi(t[o + 2 * j + 1], q.h[j]); asm volatile("" : "+v"(ja[j]), "+v"(jb[j])); };
            auto P2 = [&](int j) { uint32_t mm = pk_bf16(ja[j], jb[j]); asm volatile("" : "+v"(mm)); q.m[j] = mm; };
            auto P3 = [&](int j) { ja[j] = sub_bf_lo(ja[j], q.m[j]); jb[j] = sub_bf_hi(jb[j], q.m[j]); asm volatile("" : "+v"(ja[j]), "+v"(jb[j])); };
            auto P4 = [&](int j) { uint32_t l = pk_bf16(ja[j], jb[j]); asm volatile("" : "+v"(l)); q.l[j] = l; };
            // (this loop has five gaps: the last two MFMAs of a k-step are back to back)
            if (gp == 0) { P0(0); P0(1); P0(2); P0(3); P1(0); }
            else if (gp == 1) { P1(1); P1(2); }
            else if (gp == 2) { P1(3); P2(0); P2(1); P2(2); P2(3); }
            else if (gp == 3) { P3(0); P3(1); }
            else { P3(2); P3(3); P4(0); P4(1); P4(2); P4(3); }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 prev, prev1;
    float v0[NPR], v1[NPR];
    uint32_t sh[NPR], sm[NPR];
    // elements EP ks .. EP ks + EP - 1 of tile Tp: the gate products (stage 0) and their three-way split, one part per stage
    // (stages 1..3: five, five and one instruction per pair -- one clump of eleven behind a single MFMA left the pipe idle)
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
                else if constexpr ((ERL_K6_SLICE & 1) != 0) jit6(ks + 1, s);
                else jit(ks + 1, s);
            };
            acc = mfma_bf(a.m, b.m, acc);
            __builtin_amdgcn_sched_barrier(0);
#if defined(ERL_PROFILE) && defined(ERL_PROFILE_FINE)
            side(c);
#endif
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
            if constexpr ((ERL_K6_SLICE & 1) != 0) {
                if (To == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    jit6(ks + 1, 4);
                }
            }
            acc1 = mfma_bf(a.h, b.h, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        prev = acc;
        prev1 = acc1;
    }
#if defined(ERL_PROFILE) && defined(ERL_PROFILE_FINE)
    side(NC);
#endif
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int s = 0; s < 4; ++s) gstage(NO - 1, ks, s, false);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// split registers of an activation (NKS k-steps = 16 NKS features) of this lane's sample -> the sample-major image
template <int NKS, int CP, int KS0, int N>      // k-steps KS0 .. KS0 + NKS - 1 of p become features 0 .. 16 NKS - 1 of the image
__device__ __forceinline__ void stage_s3(u8 *S, const Parts (&p)[N], int srow, int hi)
{
    static_assert(KS0 + NKS <= N && 2 * NKS <= CP, "image too narrow");
    constexpr int PBY = 16 * CP;
    u8 *base = S + srow * (48 * CP);
    const int sw = swz<CP>(srow);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        u8 *d = base + 16 * ((2 * ks + hi) ^ sw);
        *reinterpret_cast<u32x4 *>(d) = p[KS0 + ks].h;
        *reinterpret_cast<u32x4 *>(d + PBY) = p[KS0 + ks].m;
        *reinterpret_cast<u32x4 *>(d + 2 * PBY) = p[KS0 + ks].l;
    }
}

// The A operand of a weight gradient: row tile `it` of dZ^T over the 128 staged samples, 8 k-steps x 3 parts (96 registers).
// row_sums: the bias gradient (sum over samples of every part, exact products with 1.0 accumulated in fp32) by three MFMAs per
// k-step against a ones operand; every column of the result tile is the same vector.
template <int CP>
__device__ __forceinline__ void grad_a_load(const u8 *SA, int it, Parts (&A)[8], int lane)
{
    const TrOperand<CP> ta(SA, lane);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        u32x2 r[6];
        ta.issue(it, ks, r);
        A[ks] = parts_of(r);
    }
}

__device__ __forceinline__ void grad_bias(const Parts (&A)[8], float *__restrict__ db, int it, int lane)
{
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        acc = mfma_bf(A[ks].l, ones, acc);
        acc = mfma_bf(A[ks].m, ones, acc);
        acc = mfma_bf(A[ks].h, ones, acc);
    }
    const int hi = lane >> 5;
    if ((lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) db[32 * it + crow(r, hi)] = acc[r];
    }
}

// dW tiles (it, jt0 + CS k) = A . B^T, B read from the sample-major image SB (CPB chunks per part) one k-step ahead
// `side(c, s)`: LDS stores of the NEXT phase's operand images ride behind MFMA s (0..4) of k-step c (0 .. 8 NBW - 1): the vector-memory /
// LDS ports idle beside the weight gradients' MFMAs, and a staging round of its own costs 1-2k cycles with the matrix pipe idle (round 6)
struct NoSide2 {
    __device__ __forceinline__ void operator()(int, int) const {}
};
template <int CPB, int NBW, int CS, int POLICY = 0, int POLICY_LAST = POLICY, typename Side = NoSide2>
__device__ __forceinline__ void grad_tiles(const Parts (&A)[8], const u8 *SB, int it, int jc, int jt_store0, float *__restrict__ dW, int ldw,
                                           int cols_real, int lane, const Side &side = Side())
{
    constexpr bool SIDE = !std::is_same<Side, NoSide2>::value;
    const TrOperand<CPB> tb(SB, lane);
    const int l31 = lane & 31, hi = lane >> 5;
    u32x2 rq[2][6];
    tb.issue(jc, 0, rq[0]);
    auto store = [&](const f32x16 &t, int k, auto pol) {
        const int i = 32 * (jt_store0 + jc + CS * k) + l31;
        if (i < cols_real && (decltype(pol)::value != 2 || reinterpret_cast<uintptr_t>(dW) == 1)) {      // (policy 2, diagnostics: computed, never stored)
            float *o = dW + (size_t)(32 * it + 4 * hi) * ldw + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_store_as<decltype(pol)::value>(t[r], o + (size_t)((r & 3) + 8 * (r >> 2)) * ldw);
        }
    };
    f32x16 done;
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        f32x16 acc = {0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = 8 * k + ks;
            if (c + 1 < 8 * NBW) tb.issue(jc + CS * ((c + 1) / 8), (c + 1) % 8, rq[(c + 1) & 1]);
            // the next k-step's reads go out BEFORE this k-step's MFMAs (192 cycles of cover): left in one scheduling region hipcc
            // lets the two operand buffers share registers and sinks the reads behind the fifth MFMA, one MFMA (32 cycles) before
            // their use -- every k-step then waited for the LDS (round 3: 50-52 cycles per MFMA in dW1 / dW2, floor 32)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SIDE) {
                const Parts &a = A[ks];
                const Parts b = parts_of(rq[c & 1]);
                acc = mfma_bf(a.m, b.m, acc);                 // (mma6's order: the same bits)
                __builtin_amdgcn_sched_barrier(0);
                side(c, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf(a.l, b.h, acc);
                __builtin_amdgcn_sched_barrier(0);
                side(c, 1);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf(a.h, b.l, acc);
                __builtin_amdgcn_sched_barrier(0);
                side(c, 2);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf(a.m, b.h, acc);
                __builtin_amdgcn_sched_barrier(0);
                side(c, 3);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf(a.h, b.m, acc);
                __builtin_amdgcn_sched_barrier(0);
                side(c, 4);
                __builtin_amdgcn_sched_barrier(0);
                acc = mfma_bf(a.h, b.h, acc);
            } else {
                mma6(A[ks], parts_of(rq[c & 1]), acc);
            }
            __builtin_amdgcn_sched_barrier(0);
            // a finished tile is stored behind the NEXT tile's first k-step: its last MFMA has long retired, no wait at the seam
            if (ks == 0 && k > 0) {
                store(done, k - 1, std::integral_constant<int, POLICY>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        done = acc;
    }
    store(done, NBW - 1, std::integral_constant<int, POLICY_LAST>{});
}

// LDS pool (bytes): [IMG2: W2 image, later SA][IMG1: W1 image, later SB][RW3: W3 copy fp32, later RC dY^T][biases][norm][s_part][s_red]
constexpr int kS3Img2 = 128 * 768;
constexpr int kS3Img1 = 128 * 384;
constexpr int kS3W3 = 16 * 132 * 4;
static_assert(kS3W3 >= 16 * PLD * 4, "dY^T overlays the W3 copy");
static_assert(kS3Img2 >= 128 * PLD * 4, "H2^T (fp32, feature-major) overlays the W2 image");
constexpr int kS3Small = (128 + 128 + 16 + 64 + 64 + QNW * 16 + 16 + 8) * 4;       // (the last 8 words: phase stamps of sampled launches)
constexpr size_t kS3LdsBytes = (size_t)kS3Img2 + kS3Img1 + kS3W3 + kS3Small;
static_assert(kS3LdsBytes <= 160 * 1024, "LDS budget");

// PRE: the W2 image comes ready from memory (g.w2img: built by the update loop, refreshed by clip + Adam) by LDS-DMA under the
// first layer; otherwise every workgroup splits W2 itself (stand-alone calls of erl_ppo_step_f32)
template <bool ACTOR, int KXP, int N1, int N2, bool VEC, bool PRE>     // KXP: input tiles of 32 (1: S <= 32, 2: S <= 64); 0: S <= 8
__device__ __forceinline__ void ppo_block_s3(const Ppo2Args &g, u8 *smem, SpanStamps &sps, const int slab_ix)
{
    constexpr bool TINY = KXP == 0;
    constexpr int KX = TINY ? 1 : KXP;
    constexpr int NK1 = TINY ? 1 : 2 * KX;                  // k-steps of 16 of the input actually multiplied
    constexpr int CP1 = 4 * KX, CP2 = 4 * N1, CPH2 = 4 * N2; // chunks per part: W1 / X images, W2 / H1 / dZ1 images, dZ2 image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int h1 = 32 * N1, h2 = 32 * N2;
    const int S = g.S, OUT = ACTOR ? g.A : 1;
    const Dims d{S, h1, h2, OUT};
    const float *P = g.P[net];
    const float *std_log = P + d.oStd();

    u8 *IMG2 = smem;                        // W2 image [h2][3][h1 bf16]; later SA (the dZ^T operand) / H2^T fp32
    u8 *IMG1 = IMG2 + kS3Img2;              // W1 image [h1][3][32 KX bf16]; later SB (the other operand)
    float *RW3 = reinterpret_cast<float *>(IMG1 + kS3Img1);     // W3 copy [16][ld3] fp32 (rows >= OUT zero); later RC = dY^T [16][PLD]
    float *s_b1 = RW3 + kS3W3 / 4, *s_b2 = s_b1 + 128, *s_b3 = s_b2 + 128;
    float *s_nr = s_b3 + 16, *s_nn = s_nr + 64;                 // input normalisation: x * nr + nn
    float *s_part = s_nn + 64;
    float *s_red = s_part + QNW * 16;
    constexpr int ld3 = lds_ld(128);

    PROF(0);
    // ---- prologue, trip 1: the sample id, the weights, the biases, the normalisation constants
    const int col = 32 * wave + m;                         // sample slot inside the workgroup
    const int64_t bidx = (int64_t)slab_ix * PB + col;
    const bool valid = bidx < g.B;
    const int64_t id = g.ids[valid ? bidx : 0];
    const AdvNorm advn = adv_norm_consts(ACTOR ? g.adv_stats : nullptr);   // (under the id's round trip; scalar registers)
    // FAST (round 4; images from the update loop, 16-byte-aligned rows, S > 8): W1 comes as a ready image by LDS-DMA, issued while
    // the sample ids are in flight (the old prologue spent 2.4k cycles splitting it in every workgroup), and the state rows are
    // gathered with whole-row coalesced loads (below)
    constexpr bool FAST = PRE && VEC && !TINY;
    float4 c1[FAST ? 1 : N1 * KX], c2[PRE ? 1 : N1 * N2], c3[2];
    if constexpr (FAST) {
        constexpr int KB1 = h1 * 48 * CP1 / 1024;           // the W1 image in 1 KB pieces
        static_assert(h1 * 48 * CP1 % 1024 == 0, "W1 image size");
        static_assert(KB1 % QNW == 0, "W1 image pieces per wave");
        const u8 *src1 = g.w1img[net] + 16 * lane;
#pragma unroll
        for (int i = 0; i < KB1 / QNW; ++i) {
            const int k = wave + QNW * i;                   // wave-uniform
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(src1 + 1024 * k), reinterpret_cast<float *>(IMG1 + 1024 * k), 16, 0, 0);
        }
    } else {
        copy_load<VEC, N1 * KX, QNT>(c1, P + d.oW1(), h1, S, h1, 32 * KX, tid);
    }
    // Every load of the prologue is UNCONDITIONAL (clamped addresses, values selected where they are stored): a load behind a
    // condition -- `tid < 16 ? P[..] : 0`, `valid && unmasks[row]`, a zero written over a load's own destination -- made hipcc drain
    // the whole vector-memory counter (s_waitcnt vmcnt(0)) inside the branch, five times between kernel entry and the first MFMA:
    // five exposed round trips with the weight images and the gathered rows behind them (round 4's prologue profile)
#pragma unroll
    for (int u = 0; u < 2; ++u) {                           // W3 rows [16][h2] (rows >= OUT zeroed when stored)
        const int e = tid + u * QNT, i = e / (h2 / 4), j4 = e - i * (h2 / 4);
        c3[u] = load4<VEC>(P + d.oW3() + (size_t)min(i, OUT - 1) * h2, 4 * j4, h2);
    }
    const float bias_raw = P[tid < 128 ? d.ob1() + min(tid, h1 - 1) : d.ob2() + min(tid - 128, h2 - 1)];
    const float b3_raw = P[d.ob3() + min(tid, OUT - 1)];
    const float sd_raw = g.sd[net][min(tid, S - 1)], avg_raw = g.avg[net][min(tid, S - 1)];

    // ---- trip 2: id -> (t = id % H, n = id // H) -> buffer row t*N + n  (AgentPPO.py:179-187) and its data
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
    PROF_X(16, row);                                        // (profile builds: the id has arrived)
    // this lane's own state row, in the operand order: features 16 ks + 8 hi + 0..7 of k-step ks
    float4 XR[NK1][2];
    // FAST: the rows are fetched COALESCED -- load i of a wave takes the whole rows of its samples 4 i .. 4 i + 3 (lane = (row in
    // the group, 16-byte chunk); the row numbers cross lanes by ds_bpermute) -- and pass through a wave-private LDS tile (chunks
    // XOR-swizzled by the sample) to the lanes that own them.  A lane fetching its own 32-byte pieces put 32 .. 64 different
    // lines into every load instruction: 3.3k cycles of address processing for 16 loads (round 4's prologue stamps).
    float4 GR[FAST ? 8 : 1];
    if constexpr (FAST) {
        const int r4 = lane >> 4, c16 = lane & 15;
        const uint32_t row_lo = (uint32_t)row, row_hi = (uint32_t)((uint64_t)row >> 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rw = (int64_t)(((uint64_t)(uint32_t)__shfl((int)row_hi, 4 * i + r4, 64) << 32) | (uint32_t)__shfl((int)row_lo, 4 * i + r4, 64));
            GR[i] = load4<true>(g.states + rw * S, 4 * c16, S);
        }
    } else {
        const float *xrow = g.states + row * S;
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) {
            XR[ks][0] = load4<VEC>(xrow, 16 * ks + 8 * hi, S);
            XR[ks][1] = load4<VEC>(xrow, 16 * ks + 8 * hi + 4, S);
        }
    }
    float um_k, xa_k, xb_k;
    float act_pre[4] = {0.f, 0.f, 0.f, 0.f}, sl_pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.aux) {
        // the sample's scalars from its 64-byte record (s3_image.h: [a_0 .. a_7 | logprob, raw advantage, reward_sum, unmask | pad], exact copies
        // made once per update loop): one line per sample and two loads per lane instead of four gathers from four arrays (uniform branch)
        const float4 *rec = reinterpret_cast<const float4 *>(g.aux + row * kS3AuxFloats);
        const float4 sc4 = rec[2];
        um_k = (valid && sc4.w != 0.f) ? 1.f : 0.f;
        xa_k = ACTOR ? sc4.x : sc4.z;
        xb_k = ACTOR ? sc4.y : 0.f;
        if (ACTOR) {
            const float4 v = rec[hi];
            act_pre[0] = v.x; act_pre[1] = v.y; act_pre[2] = v.z; act_pre[3] = v.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl_pre[j] = std_log[min(4 * hi + j, OUT - 1)];
        }
    } else {
        const uint8_t um_raw = g.unmasks[row];
        um_k = (valid && um_raw) ? 1.f : 0.f;
        xa_k = ACTOR ? g.logprobs[row] : g.reward_sums[row];
        xb_k = ACTOR ? g.advantages[row] : 0.f;
        if (ACTOR) {
            const bool act4 = (OUT & 3) == 0 && (reinterpret_cast<uintptr_t>(g.actions) & 15) == 0;      // uniform
            if (act4) {
                const float4 v = *reinterpret_cast<const float4 *>(g.actions + row * OUT + min(4 * hi, OUT - 4));
                act_pre[0] = v.x; act_pre[1] = v.y; act_pre[2] = v.z; act_pre[3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ac = min(4 * hi + j, OUT - 1);
                if (!act4) act_pre[j] = g.actions[row * OUT + ac];
                sl_pre[j] = std_log[ac];
            }
        }
    }
    const float &um = um_k, &xa = xa_k, &xb = xb_k;
    PROF_NV(17);                                            // (row loads issued)
    // ---- the W2 image.  Without FAST: requested here, last (the memory pipe returns W1 and the rows first), by the compiler's own
    // LDS-DMA intrinsic.  hipcc orders EVERY later LDS access behind an intrinsic LDS-DMA (it cannot tell the bytes apart): the
    // s_waitcnt vmcnt(0) it put in front of the W1 split below waited for the whole 96 KB image -- nothing "landed under the first
    // layer" (round 4's prologue stamps: 1.9k cycles, gone when the copy was taken out).  FAST requests it after barrier (0a) through
    // inline assembly the compiler does not see, and waits for it by hand before barrier (0b).
    constexpr int KB = h2 * 48 * CP2 / 1024;                // the image in 1 KB pieces: one wave instruction each, straight into LDS
    static_assert(h2 * 48 * CP2 % 1024 == 0 && KB % (2 * QNW) == 0, "image size");
    if constexpr (PRE && !FAST) {
        const u8 *src = g.w2img[net] + 16 * lane;
#pragma unroll
        for (int i = 0; i < KB / QNW; ++i) {
            const int k = wave + QNW * i;                   // wave-uniform
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(src + 1024 * k), reinterpret_cast<float *>(IMG2 + 1024 * k), 16, 0, 0);
        }
    }
    PROF_NV(18);                                            // (W2 image pieces issued / FAST: row loads issued)
    if constexpr (FAST) {
        // the rows through the wave-private tile (8 KB per wave at the start of the W2 image's region, which is still free):
        // [32 samples][16 chunks of 16 bytes], chunk c of sample s at c ^ (s & 15) -- conflict-free for the 16-byte stores by
        // (4 rows x 16 chunks) and for the reads by (sample, half)
        u8 *stg = IMG2 + 8192 * wave;
        const int r4 = lane >> 4, c16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sm = 4 * i + r4;
            *reinterpret_cast<float4 *>(stg + 256 * sm + 16 * (c16 ^ (sm & 15))) = GR[i];
        }
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) {
            XR[ks][0] = *reinterpret_cast<const float4 *>(stg + 256 * m + 16 * ((4 * ks + 2 * hi) ^ (m & 15)));
            XR[ks][1] = *reinterpret_cast<const float4 *>(stg + 256 * m + 16 * ((4 * ks + 2 * hi + 1) ^ (m & 15)));
        }
    } else {
        // ---- the weight images (split here; every workgroup converts the same 24k weights)
        img_store<N1 * KX, CP1>(c1, IMG1, h1, tid);
    }
    PROF_NV(19);                                            // (W1 has arrived and is split into its image / the rows are in their lanes)
#pragma unroll
    for (int e = tid; e < kS3W3 / 16; e += QNT) reinterpret_cast<float4 *>(RW3)[e] = zero4();
    s_b1[tid] = (tid < 128 ? tid < h1 : tid - 128 < h2) ? bias_raw : 0.f;          // s_b1 | s_b2 contiguous
    if (tid < 16) s_b3[tid] = tid < OUT ? b3_raw : 0.f;
    if (tid < 64) {
        const float nr = __builtin_amdgcn_rcpf(sd_raw + 1e-4f);                  // (x - avg) / (std + 1e-4)  (AgentPPO.py:360-361)
        s_nr[tid] = tid < S ? nr : 0.f;
        s_nn[tid] = tid < S ? -(avg_raw * nr) : 0.f;
    }
    PROF_NV(1);
    if constexpr (FAST) {
        // every value loaded so far is in its register (the tile's reads too) before the hidden copy below goes out: from here on
        // the compiler knows of no load in flight and inserts no vector-memory wait that the image pieces would stretch
        float act_l[4] = {act_pre[0], act_pre[1], act_pre[2], act_pre[3]}, sl_l[4] = {sl_pre[0], sl_pre[1], sl_pre[2], sl_pre[3]};
        float um_l = um, xa_l = xa, xb_l = xb;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(act_l[0]), "+v"(act_l[1]), "+v"(act_l[2]), "+v"(act_l[3]), "+v"(sl_l[0]), "+v"(sl_l[1]), "+v"(sl_l[2]), "+v"(sl_l[3]),
                       "+v"(um_l), "+v"(xa_l), "+v"(xb_l)::"memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) { act_pre[j] = act_l[j]; sl_pre[j] = sl_l[j]; }
        um_k = um_l; xa_k = xa_l; xb_k = xb_l;
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks)
            asm volatile("" : "+v"(XR[ks][0].x), "+v"(XR[ks][0].y), "+v"(XR[ks][0].z), "+v"(XR[ks][0].w), "+v"(XR[ks][1].x), "+v"(XR[ks][1].y),
                              "+v"(XR[ks][1].z), "+v"(XR[ks][1].w));
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(c3[u].x), "+v"(c3[u].y), "+v"(c3[u].z), "+v"(c3[u].w));
    }
    lds_barrier();                                                   // (0a) images, biases, constants visible; RW3 zeroed; row tiles read
    SPAN_STAMP(sps, 1);                                              // phase 0: prologue
    PROF_NV(2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                    // W3 copy [16][ld3], rows >= OUT zero; visible after (0b)
        const int e = tid + u * QNT, i = e / (h2 / 4), j4 = e - i * (h2 / 4);
        if (i < 16) *reinterpret_cast<float4 *>(RW3 + i * ld3 + 4 * j4) = i < OUT ? c3[u] : zero4();
    }
    // (without images) W2 is not needed before the second layer: requested only now, so that the prologue's burst is not stretched
    // by another 64 KB
    if constexpr (!PRE) copy_load<VEC, N1 * N2, QNT>(c2, P + d.oW2(), h2, h1, h2, h1, tid);
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
    PROF_X(20, XH[0][0]);                                   // (the own row has arrived and is normalised)
    f32x16 H1[N1], G1[N1], H2[N2], G2[N2];
    Parts H1p[2 * N1];
    if constexpr (FAST) {
        // the W2 image rides the first layer's k-steps: wave w copies the quarter [w KB / 4, (w + 1) KB / 4) KB of it, PPK pieces of
        // 1 KB behind the first MFMA of a k-step (one LDS base per k-step: the instruction offset advances the memory and the
        // LDS address alike); hidden from the compiler (see above), waited for by hand before barrier (0b)
        constexpr int KBW = KB / QNW, KS1 = N1 * NK1;                     // pieces per wave, k-steps of the first layer
        constexpr int PPK = (KBW + KS1 - 1) / KS1 <= 2 ? 2 : 4;           // pieces per k-step (2: 16 k-steps x 2 >= 24)
        static_assert(PPK * KS1 >= KBW && KBW % 2 == 0, "the first layer has too few k-steps for the W2 image");
        const u8 *src = g.w2img[net] + 16 * lane + KBW * 1024 * wave;
        const uint32_t l0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(IMG2 + KBW * 1024 * wave));
        auto w2_pieces = [&](int c) {
            const int q = PPK * c;
            if (q < KBW) {
                const u8 *pq = src + 1024 * q;
                if (PPK == 4 && q + 4 <= KBW)
                    asm volatile("s_mov_b32 m0, %1\n\t"
                                 "global_load_lds_dwordx4 %0, off\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:3072" ::"v"(pq), "s"(l0 + 1024u * q) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %1\n\t"
                                 "global_load_lds_dwordx4 %0, off\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:1024" ::"v"(pq), "s"(l0 + 1024u * q) : "memory");
            }
        };
        fwd_s3<NK1, N1, CP1>(IMG1, s_b1, Xp, XH, H1, G1, m, hi, w2_pieces);
    } else {
        fwd_s3<NK1, N1, CP1>(IMG1, s_b1, Xp, XH, H1, G1, m, hi);
    }
    PROF_NV(3);
    if constexpr (PRE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the W2 image have landed
    else img_store<N1 * N2, CP2>(c2, IMG2, h2, tid);
    lds_barrier();                                                   // (0b) W2 image, W3 copy visible; every wave is done with the W1 image
    SPAN_STAMP(sps, 2);                                              // phase 1: first layer forward (+ the W2 image's arrival)
    u8 *SA = IMG2, *SB = IMG1;
    {
        // the input goes to its place for dW1 right away (SB = the W1 image's bytes): 12 registers per k-step less from here on.
        // The X image always has CP1 chunks per part; k-steps the forward skipped (TINY) are zero
        Parts Xs[2 * KX];
#pragma unroll
        for (int ks = 0; ks < 2 * KX; ++ks) {
            if (ks < NK1) Xs[ks] = Xp[ks];
            else { Xs[ks].h = Xs[ks].m = Xs[ks].l = u32x4{0u, 0u, 0u, 0u}; }
        }
        stage_s3<2 * KX, CP1, 0>(SB, Xs, col, hi);
    }
#if defined(ERL_PROFILE) && defined(ERL_PROFILE_FINE)
    {
        // fine stamps: the second layer's tile boundaries (slots 21..23) and the end of its MFMA loop (24)
        auto fine = [&](int c) {
            if (c > 0 && c % (2 * N1) == 0) { if (c / (2 * N1) == 1) PROF_NV(21); else if (c / (2 * N1) == 2) PROF_NV(22); else if (c / (2 * N1) == 3) PROF_NV(23); else PROF_NV(24); }
        };
        fwd_s3<2 * N1, N2, CP2>(IMG2, s_b2, H1p, H1, H2, G2, m, hi, fine);
    }
#else
    fwd_s3<2 * N1, N2, CP2>(IMG2, s_b2, H1p, H1, H2, G2, m, hi);       // splits H1 into H1p on the way
#endif
    SPAN_STAMP(sps, 3);                                              // phase 2: second layer forward
    PROF(4);

    // ---- output layer (fp32, as in ppo_step_w4_impl.h; H2[T][4 gq + j] is feature 32 T + 16 (gq >> 1) + 8 hi + 4 (gq & 1) + j here)
    float Y[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACTOR) {
        f32x4 ya[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) ya[q >> 1][q & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *w3a = RW3 + (lane & 3) * ld3 + 8 * hi;
        constexpr int DEPTH = 2;
        float4 wq[DEPTH + 1][2];
        auto issue = [&](int c, float4(&dst)[2]) {
            const int T = c >> 2, gq = c & 3;
            dst[0] = *reinterpret_cast<const float4 *>(w3a + 32 * T + 16 * (gq >> 1) + 4 * (gq & 1));
            dst[1] = *reinterpret_cast<const float4 *>(w3a + 4 * ld3 + 32 * T + 16 * (gq >> 1) + 4 * (gq & 1));
        };
#pragma unroll
        for (int c = 0; c < DEPTH; ++c) issue(c, wq[c]);
#pragma unroll
        for (int c = 0; c < 4 * N2; ++c) {
            const int T = c >> 2, gq = c & 3;
            if (c + DEPTH < 4 * N2) issue(c + DEPTH, wq[(c + DEPTH) % (DEPTH + 1)]);
            const float4 w0 = wq[c % (DEPTH + 1)][0], w1 = wq[c % (DEPTH + 1)][1];
            const float a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ya[0][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[j], H2[T][4 * gq + j], ya[0][j & 1], 0, 0, 0);
                ya[1][j & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], H2[T][4 * gq + j], ya[1][j & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
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
        float4 wv[4 * N2];
#pragma unroll
        for (int c = 0; c < 4 * N2; ++c) wv[c] = *reinterpret_cast<const float4 *>(w3 + 32 * (c >> 2) + 16 * ((c & 3) >> 1) + 4 * (c & 1));
#pragma unroll
        for (int c = 0; c < 4 * N2; ++c) {
            const int T = c >> 2, gq = c & 3;
            yp = f32x2{wv[c].x, wv[c].y} * f32x2{H2[T][4 * gq + 0], H2[T][4 * gq + 1]} + yp;
            yq = f32x2{wv[c].z, wv[c].w} * f32x2{H2[T][4 * gq + 2], H2[T][4 * gq + 3]} + yq;
        }
        const float s = (yp.x + yp.y) + (yq.x + yq.y);
        Y[0] = s + __shfl_xor(s, 32, 64) + s_b3[0];
    }
    PROF(5);

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
        const PpoActorTerms o = ppo_actor_terms(g.objective, adv_normalized(xb, advn),   /* raw advantages are normalised here (AgentPPO.py:149) */
                                                    lp, xa, g.ratio_clip, g.lambda_entropy, um, OUT, true);
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

    // ---- dZ2 = (W3^T dY) * GELU'(z2)  (K = 8 outputs: four fp32 k-pairs; A-row m carries feature 32 To + phi(m));
    //      dZ1 = (W2^T dZ2) * GELU'(z1)
    PROF(6);
    Parts dZ2p[2 * N2];
    {
        const int pm = phi(m);
        float w3[N2][4];
#pragma unroll
        for (int To = 0; To < N2; ++To) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w3[To][j] = RW3[(4 * hi + j) * ld3 + 32 * To + pm];
        }
#pragma unroll
        for (int To = 0; To < N2; ++To) {
            f32x16 acc = {0};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma32(w3[To][j], dY[j], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) G2[To][r] *= acc[r];
        }
    }
    Parts dZ1p[2 * N1];
#if defined(ERL_PROFILE) && defined(ERL_PROFILE_FINE)
    {
        PROF(29);                                                    // (dZ2 is formed)
        auto fine = [&](int c) {
            if (c > 0 && c % (2 * N2) == 0) { if (c / (2 * N2) == 1) PROF_NV(25); else if (c / (2 * N2) == 2) PROF_NV(26); else if (c / (2 * N2) == 3) PROF_NV(27); else PROF_NV(28); }
        };
        bwd_s3<2 * N2, N1, CP2>(IMG2, dZ2p, G2, G1, dZ1p, lane, fine);
    }
#else
    bwd_s3<2 * N2, N1, CP2>(IMG2, dZ2p, G2, G1, dZ1p, lane);         // splits dZ2 into dZ2p on the way; dZ1 leaves split
#endif
    PROF(7);
    lds_barrier();                                                   // (1) every wave is done with the weight images and W3
    SPAN_STAMP(sps, 4);                                              // phase 3: output layer, objective, backward (dZ2, dZ1)
    PROF(8);

    float *slab = g.slabs + (size_t)slab_ix * g.stride + (ACTOR ? 0 : g.Pa);
    float *RC = RW3;
    // ---- layer 1: dW1 = dZ1^T . X, db1
    stage_s3<2 * N1, CP2, 0>(SA, dZ1p, col, hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        RC[(4 * hi + j) * PLD + col] = dY[j];
        RC[(8 + 4 * hi + j) * PLD + col] = dsl[j];       // rows 8..15: per-sample dL/dstd_log (zero for the critic)
    }
    lds_barrier();                                                   // (2)
    PROF(9);
#if ERL_K6_DW_ORDER != 2
    {
        // wave w owns row tile it = w % N1; the CS = 4 / N1 waves of a row tile split the KX column tiles
        constexpr int CS = 4 / N1, NBW = (KX + CS - 1) / CS;
        const int it = wave % N1, jc = wave / N1;
        if (jc < KX) {
            Parts A[8];
            grad_a_load<CP2>(SA, it, A, lane);
            grad_tiles<CP1, NBW, CS>(A, SB, it, jc, 0, slab + d.oW1(), S, S, lane);
            if (jc == 0) grad_bias(A, slab + d.ob1(), it, lane);
        }
    }
    PROF(10);
    lds_barrier();                                                   // (3) dZ1, X images consumed
    SPAN_STAMP(sps, 5);                                              // phase 4: staging + dW1, db1
#endif

    constexpr int NH1 = (N1 > 2) ? 2 : 1;                            // passes over H1's features: SB holds 64 of them
    constexpr int KSH = 2 * N1 / NH1;                                // k-steps (16 features) per pass
    constexpr int CPB2 = KSH / 2 * 4;                                // chunks per part of the SB image in dW2
    // H2^T (fp32, feature-major) over SA for dW3
    auto stage_h2t = [&]() {
        float *T2 = reinterpret_cast<float *>(SA);
#pragma unroll
        for (int t = 0; t < N2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T2[(32 * t + 16 * (r >> 3) + 8 * hi + (r & 7)) * PLD + col] = H2[t][r];
        }
    };
    // ---- output layer: dW3 (16 x h2) = dY^T . H2 on 16x16x4 fp32 MFMA (H2^T staged feature-major in fp32), db3, dstd_log
    // its operands -- the wave's rows of dY^T and of H2^T, 8 + 8 NR3 reads of 16 bytes -- are all requested up front by dw3_load(), so that the
    // phase at the kernel's end does not wait for the LDS every eight MFMAs (round 6: -0.2k cycles; requested one phase earlier, in front
    // of db2's MFMAs, they cost dW2's phase what they saved here: profiles/r06_k6_dw3_preload_ab.txt)
    constexpr int NR3 = (2 * N2 + QNW - 1) / QNW;
    float4 d3a[PB / 16], d3b[NR3][PB / 16];
    auto dw3_load = [&]() {
        const float *T2 = reinterpret_cast<const float *>(SA);
        const int l15 = lane & 15, q = lane >> 4;
        const float *a = RC + l15 * PLD + 4 * q;
#pragma unroll
        for (int j = 0; j < PB / 16; ++j) d3a[j] = *reinterpret_cast<const float4 *>(a + 16 * j);
#pragma unroll
        for (int rep = 0; rep < NR3; ++rep) {
            const int it = min(wave + QNW * rep, 2 * N2 - 1);
            const float *b = T2 + (16 * it + l15) * PLD + 4 * q;
#pragma unroll
            for (int j = 0; j < PB / 16; ++j) d3b[rep][j] = *reinterpret_cast<const float4 *>(b + 16 * j);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto dw3 = [&]() {
        const int l15 = lane & 15, q = lane >> 4;
        f32x2 hs = {0.f, 0.f};
#pragma unroll
        for (int rep = 0; rep < NR3; ++rep) {
            const int it = wave + QNW * rep;                            // 16-column tile of dW3 (wave-uniform)
            if (it >= 2 * N2) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < PB / 16; ++j) {
                const float4 av = d3a[j], bv = d3b[rep][j];
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
                if (a_ < OUT) slab_store(acc[r], slab + d.oW3() + (size_t)a_ * h2 + 16 * it + l15);
            }
        }
        float s = hs.x + hs.y;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (wave == 0 && q == 0) {
            if (l15 < OUT) slab[d.ob3() + l15] = s;
            else if (ACTOR && l15 >= 8 && l15 - 8 < OUT) slab[d.oStd() + l15 - 8] = s;
        }
    };
#if ERL_K6_DW_ORDER == 0
    // order dW1, dW3, dW2: the first half of H1 goes to SB while H2^T is staged
    stage_h2t();
    stage_s3<KSH, CPB2, 0>(SB, H1p, col, hi);
    lds_barrier();                                                   // (4)
    PROF(11);
    dw3_load();
    dw3();
    lds_barrier();                                                   // (5) H2^T consumed
    stage_s3<2 * N2, CPH2, 0>(SA, dZ2p, col, hi);
    lds_barrier();                                                   // (6)
    SPAN_STAMP(sps, 6);                                              // phase 5: staging + dW3, db3 + staging; phase 6 = dW2, db2, logs, store drain
    PROF(12);

    // ---- layer 2: dW2 = dZ2^T . H1, db2  (wave w: row tile w % N2; its column tiles pass by pass)
    {
        constexpr int CS = 4 / N2;
        constexpr int TPP = N1 / NH1;                                // column tiles per pass
        constexpr int NBW = (TPP + CS - 1) / CS;
        const int it = wave % N2, jc = wave / N2;
        Parts A[8];
        grad_a_load<CPH2>(SA, it, A, lane);
        if (jc < TPP) grad_tiles<CPB2, NBW, CS, (ERL_K6_EXP & 2) ? 1 : 0>(A, SB, it, jc, 0, slab + d.oW2(), h1, h1, lane);
        if (NH1 == 2) {
            lds_barrier();                                           // (7) first half of H1 consumed
            stage_s3<KSH, CPB2, (NH1 == 2 ? KSH : 0)>(SB, H1p, col, hi);
            lds_barrier();                                           // (8)
            if (jc < TPP) grad_tiles<CPB2, NBW, CS, (ERL_K6_EXP & 3) ? 1 : 0, (ERL_K6_EXP & 11) ? 1 : 0>(A, SB, it, jc, TPP, slab + d.oW2(), h1, h1, lane);
        }
        if (jc == 0) grad_bias(A, slab + d.ob2(), it, lane);
    }
    PROF(13);
#elif ERL_K6_DW_ORDER == 2
    // order dW1, dW2, dW3 with the operand images of the NEXT phase staged UNDER the current phase's MFMAs (round 6).  SA is read only by a
    // phase's first instructions (a wave's 96-register A operand) and by dW3: once every wave holds its rows of dZ1^T (barrier 2b) the dZ2
    // image is written over it behind dW1's MFMAs, and once every wave holds its rows of dZ2^T (barrier 4) H2^T behind the first pass of dW2.
    // Seven barriers instead of eight; the staging rounds (5)-(6) and half of (3)-(4) of the other orders are gone.
    {
        constexpr int CS1 = 4 / N1, NBW1 = (KX + CS1 - 1) / CS1;
        const int it = wave % N1, jc = wave / N1;
        Parts A[8];
        grad_a_load<CP2>(SA, it, A, lane);
        lds_barrier();                                               // (2b) every wave holds its rows of dZ1^T: SA is free
        PROF(10);
        constexpr int NP = 6 * N2;                                   // the dZ2 image of this lane's sample: 2 N2 k-steps x 3 parts of 16 bytes
        constexpr int PPK = (NP + 8 * NBW1 - 1) / (8 * NBW1);        // pieces per k-step
        static_assert(PPK <= 5, "dZ2 image pieces per k-step of dW1");
        u8 *zb = SA + col * (48 * CPH2);
        const int zsw = swz<CPH2>(col);
        auto piece = [&](int i) {
            const int ks = i / 3, pl = i % 3;
            u8 *dst = zb + 16 * ((2 * ks + hi) ^ zsw) + pl * (16 * CPH2);
            *reinterpret_cast<u32x4 *>(dst) = pl == 0 ? dZ2p[ks].h : pl == 1 ? dZ2p[ks].m : dZ2p[ks].l;
        };
        // the logged sums' per-wave parts (block_sum's first half: the wave butterfly, one hop per k-step, behind MFMA 3 or 4; its scratch is
        // free until the kernel's end): the four parts are added at the end in block_sum's order -- the same bits -- without its two butterflies
        // and four barriers there, where every cycle is on the critical path (round 6: -1.9k cycles at the end)
        float lw0 = loss0, lw1 = loss1;
        auto logs_hop = [&](int hop) {
            if (hop < 6) {
                lw0 += __shfl_xor(lw0, 32 >> hop, 64);
                lw1 += __shfl_xor(lw1, 32 >> hop, 64);
            } else if (hop == 6 && lane == 0) {
                s_red[wave] = lw0;
                s_red[QNW + wave] = lw1;
            }
        };
        if (jc < KX) {
            auto side = [&](int c, int sl) {
                if (sl < PPK && PPK * c + sl < NP) piece(PPK * c + sl);
                if constexpr (ERL_K6_EARLY_LOGS) {
                    if (sl == (PPK < 4 ? 3 : 4) && c < 7) logs_hop(c);
                }
            };
            grad_tiles<CP1, NBW1, CS1, 0, 0>(A, SB, it, jc, 0, slab + d.oW1(), S, S, lane, side);
            if (jc == 0) grad_bias(A, slab + d.ob1(), it, lane);
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) piece(i);
            if constexpr (ERL_K6_EARLY_LOGS) {
#pragma unroll
                for (int hop = 0; hop < 7; ++hop) logs_hop(hop);
            }
        }
    }
    PROF(11);
    lds_barrier();                                                   // (3) the X image is consumed, the dZ2 image complete
    SPAN_STAMP(sps, 5);                                              // phase 4: staging + dW1, db1 (+ the dZ2 image)
    {
        constexpr int CS = 4 / N2;
        constexpr int TPP = N1 / NH1;                                // column tiles per pass
        constexpr int NBW = (TPP + CS - 1) / CS;
        static_assert(TPP >= CS, "every wave owns a column tile of dW2");
        const int it = wave % N2, jc = wave / N2;
        Parts A[8];
        grad_a_load<CPH2>(SA, it, A, lane);
        stage_s3<KSH, CPB2, 0>(SB, H1p, col, hi);
        lds_barrier();                                               // (4) every wave holds its rows of dZ2^T: SA is free; first half of H1 in SB
        PROF(12);
        constexpr int NP = 16 * N2;                                  // H2^T (fp32, feature-major), one 4-byte store per element
        constexpr int PPK = (NP + 8 * NBW - 1) / (8 * NBW);
        static_assert(PPK <= 5, "H2^T elements per k-step of dW2");
        float *T2 = reinterpret_cast<float *>(SA);
        auto side = [&](int c, int sl) {
            const int i = PPK * c + sl;
            if (sl < PPK && i < NP) {
                const int t = i >> 4, r = i & 15;
                T2[(32 * t + 16 * (r >> 3) + 8 * hi + (r & 7)) * PLD + col] = H2[t][r];
            }
        };
        constexpr int PW2 = (ERL_K6_EXP & 32) ? 2 : 0;               // (diagnostics: dW2 not stored)
        grad_tiles<CPB2, NBW, CS, PW2, PW2>(A, SB, it, jc, 0, slab + d.oW2(), h1, h1, lane, side);
        if (NH1 == 2) {
            lds_barrier();                                           // (5) first half of H1 consumed; H2^T written
            stage_s3<KSH, CPB2, (NH1 == 2 ? KSH : 0)>(SB, H1p, col, hi);
            lds_barrier();                                           // (6)
            grad_tiles<CPB2, NBW, CS, PW2, PW2>(A, SB, it, jc, TPP, slab + d.oW2(), h1, h1, lane);
        }
        if (NH1 != 2) lds_barrier();                                 // (5) H2^T written
        if (jc == 0) grad_bias(A, slab + d.ob2(), it, lane);
    }
    SPAN_STAMP(sps, 6);                                              // phase 5: dW2, db2 (+ H2^T); phase 6 = dW3, db3, logs, store drain
    PROF(13);
    dw3_load();
    dw3();
    PROF_NV(14);
#else
    // order dW1, dW2, dW3 (round 6): the 64 KB of dW2 -- two thirds of a workgroup's slab -- leave by write-through stores while dW3 is
    // still computing, and the kernel ends behind the 4 KB of dW3 instead of behind the drain of 256 x 64 KB; H2 waits in registers
    stage_s3<2 * N2, CPH2, 0>(SA, dZ2p, col, hi);
    stage_s3<KSH, CPB2, 0>(SB, H1p, col, hi);
    lds_barrier();                                                   // (4)
    PROF(11);
    // ---- layer 2: dW2 = dZ2^T . H1, db2  (wave w: row tile w % N2; its column tiles pass by pass)
    {
        constexpr int CS = 4 / N2;
        constexpr int TPP = N1 / NH1;                                // column tiles per pass
        constexpr int NBW = (TPP + CS - 1) / CS;
        const int it = wave % N2, jc = wave / N2;
        Parts A[8];
        grad_a_load<CPH2>(SA, it, A, lane);
        if (jc < TPP) grad_tiles<CPB2, NBW, CS>(A, SB, it, jc, 0, slab + d.oW2(), h1, h1, lane);
        if (NH1 == 2) {
            lds_barrier();                                           // (5) first half of H1 consumed; every wave holds its rows of dZ2^T
            stage_s3<KSH, CPB2, (NH1 == 2 ? KSH : 0)>(SB, H1p, col, hi);
            stage_h2t();
            lds_barrier();                                           // (6)
            if (jc < TPP) grad_tiles<CPB2, NBW, CS>(A, SB, it, jc, TPP, slab + d.oW2(), h1, h1, lane);
        }
        if (jc == 0) grad_bias(A, slab + d.ob2(), it, lane);
        if (NH1 != 2) {
            lds_barrier();                                           // (5) dZ2^T consumed
            stage_h2t();
            lds_barrier();                                           // (6)
        }
    }
    SPAN_STAMP(sps, 6);                                              // phase 5: staging + dW2, db2; phase 6 = dW3, db3, logs, store drain
    PROF(12);
    dw3_load();
    dw3();
    PROF(13);
#endif

    // ---- objective partial sums (scaled by 1/B so that the slab reduction yields the means)
    float t0, t1;
    if constexpr (ERL_K6_EARLY_LOGS && ERL_K6_DW_ORDER == 2) {
        t0 = 0.f; t1 = 0.f;
        if (tid == 0) {                                   // (visible since barrier (1))
#pragma unroll
            for (int w = 0; w < QNW; ++w) { t0 += s_red[w]; t1 += s_red[QNW + w]; }
        }
    } else {
        t0 = block_sum(loss0, s_red);
        t1 = block_sum(loss1, s_red);
    }
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
#if ERL_K6_DW_ORDER == 2
    PROF_NV(15);
#endif
#if ERL_K6_EXP & 4
    // (experiment) the critic's workgroups finish ~5k cycles before the actor's: pull the NEXT minibatch's rows of this slab towards this
    // XCD's L2 (the actor's workgroup of the same slab index sits on the same XCD: 128 % 8 == 0), where the next launch's prologue finds them
    if (!ACTOR && g.next_ids && tid < PB) {
        const int64_t nb = (int64_t)slab_ix * PB + tid;
        const int64_t nid = g.next_ids[nb < g.B ? nb : 0];
        const int64_t nn = nid / g.H, nt = nid - nn * g.H, nrow = nt * g.N + nn;
        const float *xr = g.states + nrow * S;
        float sink = 0.f;
        for (int c = 0; c < S; c += 32) sink += xr[c] * 0.f;                                    // one load per 128-byte line of the row
        sink += g.actions[nrow * g.A] + g.logprobs[nrow] + g.advantages[nrow] + g.reward_sums[nrow] + (float)g.unmasks[nrow];
        asm volatile("" ::"v"(sink));
    }
#endif
}

template <int KX, int N1, int N2, bool VEC, bool PRE>
__global__ __launch_bounds__(QNT) void ppo_step_s3_kernel(Ppo2Args g)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_s3[];
    const SpanT t_span = span_enter(g);
    if (k6_code_touch(g)) return;
    SpanStamps sps{reinterpret_cast<uint32_t *>(smem_s3 + kS3LdsBytes - 32)};
    const K6Wg wg = k6_wg_map(g);
#if ERL_K6_EXP & 16
    // (diagnostics) every workgroup on ONE network's code path: do two code paths of 55 KB each on a 64 KB instruction cache matter?
    const bool as_actor = g.exp_net == 0 || (g.exp_net != 1 && wg.actor);
#else
    const bool as_actor = wg.actor;
#endif
    if (as_actor) ppo_block_s3<true, KX, N1, N2, VEC, PRE>(g, smem_s3, sps, wg.slab);
    else ppo_block_s3<false, KX, N1, N2, VEC, PRE>(g, smem_s3, sps, wg.slab);
    span_exit(g, t_span, as_actor ? &sps : nullptr);
}

template <int KX, int N1, int N2, bool VEC, bool PRE>
int launch_s3(const Ppo2Args &g, int n_slabs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)ppo_step_s3_kernel<KX, N1, N2, VEC, PRE>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS3LdsBytes),
                                "hipFuncSetAttribute(ppo_step_s3_kernel)");
        if (rc) return rc;
        attr_set = true;
    }
    if (g.code_touch_bytes) {
        // a code-touch request (ppo_step.h k6_code_touch): where THIS instantiation's code lives on this device -- asked of the kernel
        // itself once (a one-workgroup launch that reports its program counter; one stream sync per instantiation, device and process)
        struct Range { bool tried = false; const unsigned char *base = nullptr; unsigned bytes = 0; };
        static Range ranges[64];
        int dev = 0;
        Ppo2Args t = g;
        t.code_touch = nullptr; t.code_touch_bytes = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            Range &r = ranges[dev];
            if (!r.tried) {
                r.tried = true;
                unsigned long long *d_pc = nullptr, pc = 0;
                if (hipMalloc((void **)&d_pc, sizeof(pc)) == hipSuccess) {
                    Ppo2Args q = g;
                    q.code_touch_bytes = 0; q.span = nullptr; q.pc_out = d_pc;
                    hipLaunchKernelGGL((ppo_step_s3_kernel<KX, N1, N2, VEC, PRE>), dim3(1, 1), dim3(QNT), kS3LdsBytes, stream, q);
                    if (hipMemcpyAsync(&pc, d_pc, sizeof(pc), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess && pc)
                        (void)erl_k6_code_range(pc, (size_t)128 << 10, &r.base, &r.bytes);
                    (void)hipFree(d_pc);
                }
                (void)hipGetLastError();
            }
            t.code_touch = r.base; t.code_touch_bytes = r.base ? r.bytes : 0;
        }
        hipLaunchKernelGGL((ppo_step_s3_kernel<KX, N1, N2, VEC, PRE>), dim3(n_slabs, g.only_net >= 0 ? 1 : 2), dim3(QNT), kS3LdsBytes, stream, t);
        return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32");
    }
    hipLaunchKernelGGL((ppo_step_s3_kernel<KX, N1, N2, VEC, PRE>), dim3(n_slabs, g.only_net >= 0 ? 1 : 2), dim3(QNT), kS3LdsBytes, stream, g);
    return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32");
}

template <int N1, int N2, bool PRE>
int launch_s3_shape(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream)
{
    if (g.S <= 8) return launch_s3<0, N1, N2, false, PRE>(g, n_slabs, stream);
    if (vec) return g.S > 32 ? launch_s3<2, N1, N2, true, PRE>(g, n_slabs, stream) : launch_s3<1, N1, N2, true, PRE>(g, n_slabs, stream);
    return g.S > 32 ? launch_s3<2, N1, N2, false, PRE>(g, n_slabs, stream) : launch_s3<1, N1, N2, false, PRE>(g, n_slabs, stream);
}

}  // namespace
