// K6: one PPO minibatch (gather + actor & critic forward + objective + full backward), gfx950 fp32 MFMA.
//
// Replaces AgentPPO.update_objectives up to the optimizer steps (elegantrl/agents/AgentPPO.py:173-204) and
// ActorPPO.get_logprob_entropy (:378-386).  grid = (ceil(B / 128), 2): blockIdx.y = 0 actor, 1 critic; a workgroup
// of 8 waves owns 128 samples and writes ONE gradient slab (no accumulation across workgroups here; the slabs are
// summed in a fixed order by erl_grad_reduce_f32 -> deterministic).
//
// Register-chained transposed formulation.  Wave w owns 16 samples; every layer is computed transposed,
//     outT (feat x 16 samples) = W (feat x K) . inT (K x 16 samples),
// on v_mfma_f32_16x16x4_f32 (exact fp32): A = 16 weight rows x 4 k, B = 4 k x 16 samples, and the result tile
// leaves lane (m = lane & 15, q = lane >> 4) holding features 16 t + 4 q + r (r = 0..3) of sample m.  The K loop of
// the NEXT layer is ordered so that step (t, r) consumes k = 16 t + 4 q + r from lane group q: that is exactly
// the register the lane already holds, so activations never leave the register file between layers (no LDS
// round trip, no barrier), and the matching A operand is one 16-byte read W[row][16 t + 4 q .. + 3] per 4 MFMAs.
// The same holds for the backward-input chain (dZ2 -> dZ1) with W3^T / W2^T as A operands.  All three weight
// matrices and the biases are copied into LDS once per workgroup (zero padded to the tile grid; row stride
// 4 * odd floats: conflict-free ds_read_b128 along k for the forward pass, conflict-free ds_read_b32 along the
// transposed direction for the backward pass), so the compute phases never wait on L2 / HBM latency.
//
// Weight gradients need the sample dimension as the MFMA K dimension, i.e. the transpose of what the lanes hold:
// the 8 waves stage their tiles feature-major in LDS (T[feature][sample], row stride PLD = 132) and split the dW output
// tiles (v_mfma_f32_32x32x2_f32, K = 128 samples; the sum over samples is order-free, so each lane half reads four
// consecutive samples with one ds_read_b128 per operand and feeds four MFMAs).  GELU'(z1), needed again only by the last
// step of the chain, is parked in the workgroup's own still-dead gradient slab instead of occupying 32 VGPRs across the
// whole forward pass.  Seven workgroup barriers (LDS-only, no vector-memory drain) in total; all control flow in the hot
// instantiation is compile-time (tile counts are template parameters, out-of-range loads are clamped + selected instead
// of branched).
#include <stdlib.h>
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "ppo_step.h"
#include "ppo_step_wd.h"
#include "s3_image.h"
#include <cstring>
#include <mutex>

namespace {

constexpr int PNW = 8;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;

// ---------------------------------------------------------------------------------------------------------
// backward through a layer's input on registers:  g[jt] <- g[jt] * ( W^T . dz ),  W = zero-padded LDS copy
// [16 kt][ldw] (A operand = W^T: lane (j, q) supplies W[16 t + 4 q + r][16 jt + j], four ds_read_b32 per k-tile).
// On entry g holds the gate GELU'(z_in) (kept in registers from the forward pass, or read back from the slab).
// ---------------------------------------------------------------------------------------------------------
template <int KT>
__device__ __forceinline__ void backward_input(const float *W, int ldw, int kt_rt, int nin, const f32x4 (&dz)[8], f32x4 (&g)[8],
                                               int l15, int q)
{
    constexpr int NCH = KT ? (KT + PCH - 1) / PCH : 8 / PCH;
    constexpr int NC = 8 * NCH;
    const int kt = KT ? KT : kt_rt;
    float4 wq[2][PCH];
    auto issue = [&](int c, float4(&dst)[PCH]) {
        const int jt = c / NCH, th = c % NCH;
#pragma unroll
        for (int j = 0; j < PCH; ++j) {
            const int t = PCH * th + j;
            if (jt < nin && t < kt) {
                const float *p = W + (16 * t + 4 * q) * ldw + 16 * jt + l15;
                dst[j] = make_float4(p[0], p[ldw], p[2 * ldw], p[3 * ldw]);
            }
        }
    };
    issue(0, wq[0]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int jt = c / NCH, th = c % NCH;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (jt < nin) {
            if (th == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < PCH; ++j) {
                const int t = PCH * th + j;
                if (t < kt) {
                    const float4 wv = wq[c & 1][j];
                    acc = mfma16(wv.x, dz[t][0], acc);
                    acc = mfma16(wv.y, dz[t][1], acc);
                    acc = mfma16(wv.z, dz[t][2], acc);
                    acc = mfma16(wv.w, dz[t][3], acc);
                }
            }
            if (th == NCH - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g[jt][r] *= acc[r];
            }
        }
    }
}

// stage a register-resident activation (D layout) feature-major into LDS: T[feature][16 w + m]
__device__ __forceinline__ void stage(float *T, const f32x4 (&a)[8], int nt, int col, int q)
{
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 * t + 4 * q + r) * PLD + col] = a[t][r];
        }
}

// LDS pool (floats): [RA: W2 copy, later staged tiles][RB: W1 copy, later staged tiles][RC: dY^T][RW3: W3 copy]
//                    [s_bias: b1 | b2 | b3(16)][s_part: 8*16][s_red: 16]
constexpr int kRFloats = 128 * 68 + 64 * PLD > 128 * 132 ? 128 * 68 + 64 * PLD : 128 * 132;
static_assert(kRFloats >= 128 * PLD && kRFloats % 4 == 0, "staged tiles must fit the weight-copy regions");
// >= 128 * lds_ld(128) (W2 copy), >= 128 * PLD (staged tile), >= 128 * lds_ld(64) + 64 * PLD (W1 copy | X^T, tuned shape)
constexpr int kRCFloats = 16 * PLD;
constexpr int kRW3Floats = 16 * 132;
constexpr int kBiasFloats = 128 + 128 + 16;
constexpr size_t kPpoLdsBytes = (size_t)(2 * kRFloats + kRCFloats + kRW3Floats + kBiasFloats + PNW * 16 + 16) * sizeof(float);

template <bool ACTOR, int NS_, int N1_, int N2_, bool VEC>
__device__ __forceinline__ void ppo_block(const Ppo2Args &g, float *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int net = ACTOR ? 0 : 1;
    const Dims d{g.S, g.h1, g.h2, ACTOR ? g.A : 1};
    const int S = d.S, h1 = N1_ ? 16 * N1_ : d.h1, h2 = N2_ ? 16 * N2_ : d.h2, OUT = d.out;
    const int ns = NS_ ? NS_ : (S + 15) >> 4, n1 = N1_ ? N1_ : h1 >> 4, n2 = N2_ ? N2_ : h2 >> 4;
    const float *P = g.P[net];
    const float *std_log = P + d.oStd();

    float *RA = smem;                      // W2 copy [h2][ld2], later staged tiles [128][PLD]
    float *RB = RA + kRFloats;             // W1 copy [h1][ld1], later staged tiles [128][PLD]
    float *RC = RB + kRFloats;             // [16][PLD]   dY^T
    float *RW3 = RC + kRCFloats;           // W3 copy [16][ld3] (rows >= OUT are zero)
    float *s_b1 = RW3 + kRW3Floats, *s_b2 = s_b1 + 128, *s_b3 = s_b2 + 128;
    float *s_part = s_b3 + 16;             // [8 waves][16]  per-wave dstd_log partials
    float *s_red = s_part + PNW * 16;      // [16] block_sum scratch
    const int ld1 = lds_ld(16 * ns), ld2 = lds_ld(h1), ld3 = lds_ld(h2);

    PROF(0);
    // ---- prologue: two global round trips.  Trip 1: the sample id and the weight/bias copies.
    const int col = 16 * wave + l15;                       // sample slot inside the workgroup
    const int64_t bidx = (int64_t)blockIdx.x * PB + col;
    const bool valid = bidx < g.B;
    const int64_t id = g.ids[valid ? bidx : 0];
    const AdvNorm advn = adv_norm_consts(ACTOR ? g.adv_stats : nullptr);   // (under the id's round trip; scalar registers)
    float4 c2[8], c1[8], c3[1];
    copy_load<VEC, 8, PNW * 64>(c2, P + d.oW2(), h2, h1, h2, h1, tid);
    copy_load<VEC, 8, PNW * 64>(c1, P + d.oW1(), h1, S, h1, 16 * ns, tid);
    copy_load<VEC, 1, PNW * 64>(c3, P + d.oW3(), OUT, h2, 16, h2, tid);
    float bias_pre = 0.f;                                  // b1 | b2 | b3 (one element per thread 0..271)
    if (tid < 128) bias_pre = (tid < h1) ? P[d.ob1() + tid] : 0.f;
    else if (tid < 256) bias_pre = (tid - 128 < h2) ? P[d.ob2() + tid - 128] : 0.f;
    else if (tid < 272) bias_pre = (tid - 256 < OUT) ? P[d.ob3() + tid - 256] : 0.f;

    // ---- trip 2: id -> (t = id % H, n = id // H) -> buffer row t*N + n  (AgentPPO.py:179-187) and its data
    int64_t n_, t_;
    if (g.H * g.N <= 0x7fffffffLL) {       // uniform branch: ids < H N fit 32 bits, a 32-bit divide is ~4x shorter and it sits
        const uint32_t i32 = (uint32_t)id, h32 = (uint32_t)g.H, n32 = i32 / h32;   // between the two dependent round trips
        n_ = n32;
        t_ = i32 - n32 * h32;
    } else {
        n_ = id / g.H;
        t_ = id - n_ * g.H;
    }
    const int64_t row = valid ? t_ * g.N + n_ : 0;
    // this sample's raw state slice, features 16 t + 4 q + r; normalised by norm_x (AgentPPO.py:360-361)
    const float *srow = g.states + row * S;
    const float *avg = g.avg[net], *sdv = g.sd[net];
    auto load_x_raw = [&](float4(&R)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ns) R[t] = load4<VEC>(srow, 16 * t + 4 * q, S);
    };
    auto norm_x = [&](const float4(&R)[8], f32x4(&X)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < ns) {
                const int k0 = 16 * t + 4 * q;
                const float4 a4 = load4<VEC>(avg, k0, S), s4 = load4<VEC>(sdv, k0, S);
                const float rr[4] = {R[t].x, R[t].y, R[t].z, R[t].w}, aa[4] = {a4.x, a4.y, a4.z, a4.w},
                            ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xn = (rr[r] - aa[r]) / (ss[r] + 1e-4f);
                    X[t][r] = (valid && k0 + r < S) ? xn : 0.f;
                }
            }
        }
    };
    float4 XR[8];
    load_x_raw(XR);

    // ---- publish the LDS copies (zero padded to the tile grid), visible after barrier (0)
    copy_store<8, PNW * 64>(c2, RA, ld2, h2, h1, tid);
    copy_store<8, PNW * 64>(c1, RB, ld1, h1, 16 * ns, tid);
    copy_store<1, PNW * 64>(c3, RW3, ld3, 16, h2, tid);
    if (tid < 272) s_b1[tid] = bias_pre;                   // s_b1 | s_b2 | s_b3 are contiguous

    // X^T for dW1 lives next to the W1 copy when both fit (S <= 64): staged once, here, from the registers
    constexpr bool EARLY_X = NS_ != 0 && NS_ <= 4;
    float *RX = EARLY_X ? RB + 128 * lds_ld(64) : RB;
    // GELU'(z1) is needed again only at the very end of the chain (dZ1).  Holding it in registers across the whole
    // forward pass pushes the live set past 256 VGPRs (spill reloads cost 10-20k cycles each); recomputing it there costs
    // 128 MFMAs + ~650 VALU ops per wave.  The tuned shape parks it in memory instead: this workgroup's own gradient
    // slab is dead until the weight-gradient phases, so each lane writes its 8 float4 there right after L1 and reads
    // the same 8 float4 back before dZ1 (64 KB per workgroup, same lane, same address: no synchronisation; the slab
    // is at least 16384 + 3 floats long for this shape).
    constexpr bool GMEM = EARLY_X && N1_ == 8;
    static_assert(GMEM || !EARLY_X, "the tuned instantiation parks GELU'(z1) in the slab; generic shapes keep it in registers");
    float *slab = g.slabs + (size_t)blockIdx.x * g.stride + (ACTOR ? 0 : g.Pa);
    float *gscr = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(slab) + 15) & ~static_cast<uintptr_t>(15)) + 4 * tid;
    f32x4 H1[8], G1[8], H2[8], G2[8];
    f32x4 X[8];
    norm_x(XR, X);
    if (EARLY_X) stage(RX, X, ns, col, q);
    PROF(1);
    lds_barrier();                                                   // (0) weight copies visible
    PROF(2);
    forward_layer<true, NS_, (!EARLY_X || GMEM)>(RB, ld1, s_b1, ns, n1, X, H1, G1, l15, q);
    if (GMEM) {
#pragma unroll
        for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4 *>(gscr + t * (4 * PNW * 64)) = G1[t];
    }
    PROF(3);
    forward_layer<true, N1_>(RA, ld2, s_b2, n1, n2, H1, H2, G2, l15, q);
    PROF(4);
    // per-sample scalars: issued here (L2 / MALL hits by now), consumed after the output layer
    const float um = (valid && g.unmasks[row]) ? 1.f : 0.f;
    const float xa = ACTOR ? g.logprobs[row] : g.reward_sums[row];
    const float xb = ACTOR ? g.advantages[row] : 0.f;
    float act_pre[4] = {0.f, 0.f, 0.f, 0.f}, sl_pre[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's actions a = 4 q + r (actor)
    if (ACTOR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ac = min(4 * q + r, OUT - 1);
            act_pre[r] = g.actions[row * OUT + ac];
            sl_pre[r] = std_log[ac];
        }
    }
    f32x4 Y[8], dummy[8];
    forward_layer<false, N2_>(RW3, ld3, s_b3, n2, 1, H2, Y, dummy, l15, q);
    if (GMEM) {   // GELU'(z1) back from the slab; the laundered pointer keeps the compiler from forwarding the stores
        const float *gl = gscr;
        asm volatile("" : "+v"(gl));
#pragma unroll
        for (int t = 0; t < 8; ++t) G1[t] = *reinterpret_cast<const f32x4 *>(gl + t * (4 * PNW * 64));
    }
    PROF(5);

    // ---- objective and dL/dY for this lane's outputs a = 4 q + r   (AgentPPO.py:189-204)
    f32x4 dY[8];
    float loss0 = 0.f, loss1 = 0.f;
    float dsl[4] = {0.f, 0.f, 0.f, 0.f};
    if (!ACTOR) {
        const float diff = Y[0][0] - xa;                  // only (q = 0, r = 0) is the value head
        const bool head = q == 0;
        loss0 = head ? diff * diff * um : 0.f;
        dY[0] = f32x4{head ? 2.f * diff * um * g.inv_batch : 0.f, 0.f, 0.f, 0.f};
    } else {
        float diffv[4], varv[4];
        float lp = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * q + r;
            const float sl = sl_pre[r], sd_ = expf(sl);
            const float diff = act_pre[r] - Y[0][r];
            const bool on = a < OUT;
            varv[r] = sd_ * sd_;
            diffv[r] = on ? diff : 0.f;
            const float term = -(diff * diff) / (2.f * varv[r]) - logf(sd_) - kLogSqrt2Pi;
            lp += on ? term : 0.f;
        }
        lp += __shfl_xor(lp, 16, 64);
        lp += __shfl_xor(lp, 32, 64);
        const PpoActorTerms o = ppo_actor_terms(g.objective, adv_normalized(xb, advn),   /* raw advantages are normalised here (AgentPPO.py:149) */
                                                    lp, xa, g.ratio_clip, g.lambda_entropy, um, OUT, false);
        if (q == 0) {
            loss0 = valid ? o.logged : 0.f;                    // padding rows contribute 0
            loss1 = valid ? o.ent_mask : 0.f;
        }
        const float dlp = (valid ? o.dlp : 0.f) * g.inv_batch;  // d loss / dlogp_new
        const float ent_term = (valid ? o.ent_w : 0.f) * g.inv_batch;
        f32x4 dy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool on = 4 * q + r < OUT;
            dy[r] = on ? dlp * (diffv[r] / varv[r]) : 0.f;                                     // dL/dmean
            dsl[r] = on ? dlp * (diffv[r] * diffv[r] / varv[r] - 1.f) + ent_term : 0.f;        // dL/dstd_log, this sample
        }
        dY[0] = dy;
    }

    // ---- per-wave dstd_log partials
    if (ACTOR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = dsl[r];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            if (l15 == 0) s_part[wave * 16 + 4 * q + r] = s;
        }
    }
    // ---- dZ2 = (W3^T dY) * GELU'(z2)  (K = 16 outputs: one k-tile);  dZ1 = (W2^T dZ2) * GELU'(z1)
    PROF(6);
    backward_input<1>(RW3, ld3, 1, n2, dY, G2, l15, q);            // G2 (the gate) <- dZ2
    backward_input<N2_>(RA, ld2, n2, n1, G2, G1, l15, q);          // G1 (the gate) <- dZ1
    PROF(7);
    lds_barrier();                                                   // (1) every wave is done with the weight copies
    PROF(8);

    // ---- layer 1: dW1 = dZ1^T . X, db1;  (dY^T is staged alongside for the output layer)
    stage(RA, G1, n1, col, q);                                      // dZ1^T
#pragma unroll
    for (int r = 0; r < 4; ++r) RC[(4 * q + r) * PLD + col] = dY[0][r];
    if (!EARLY_X) {                                                 // generic shapes: re-gather X now (the W1 copy is dead)
        load_x_raw(XR);
        norm_x(XR, X);
        stage(RX, X, ns, col, q);
    }
    lds_barrier();                                                   // (2)
    PROF(9);
    weight_grad<PNW>(RA, h1 >> 5, RX, (S + 31) >> 5, slab + d.oW1(), S, S, wave, lane);
    bias_grad<PNW>(RA, h1, slab + d.ob1(), wave, lane);
    if (wave == 0) {
        bias_grad<PNW>(RC, OUT, slab + d.ob3(), 0, lane);
        if (ACTOR && lane < OUT) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < PNW; ++u) s += s_part[u * 16 + lane];
            slab[d.oStd() + lane] = s;
        }
    }
    PROF(10);
    lds_barrier();                                                   // (3) dZ1^T, X^T consumed

    // ---- output layer: dW3 (16 x h2) = dY^T . H2 on 16x16x4 MFMA, one 16-column tile per wave
    stage(RA, H2, n2, col, q);                                      // H2^T
    stage(RB, H1, n1, col, q);                                      // H1^T (for dW2)
    lds_barrier();                                                   // (4)
    PROF(11);
    for (int it = wave; it < n2; it += PNW) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float *a = RC + l15 * PLD + 4 * q;                    // lane group q: samples 16 j + 4 q + {0..3}
        const float *b = RA + (16 * it + l15) * PLD + 4 * q;
#pragma unroll
        for (int j = 0; j < PB / 16; ++j) {
            const float4 av = *reinterpret_cast<const float4 *>(a + 16 * j), bv = *reinterpret_cast<const float4 *>(b + 16 * j);
            acc = mfma16(av.x, bv.x, acc);
            acc = mfma16(av.y, bv.y, acc);
            acc = mfma16(av.z, bv.z, acc);
            acc = mfma16(av.w, bv.w, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a_ = 4 * q + r;
            if (a_ < OUT) __builtin_nontemporal_store(acc[r], slab + d.oW3() + (size_t)a_ * h2 + 16 * it + l15);
        }
    }
    lds_barrier();                                                   // (5) H2^T consumed
    stage(RA, G2, n2, col, q);                                      // dZ2^T
    lds_barrier();                                                   // (6)
    PROF(12);

    // ---- layer 2: dW2 = dZ2^T . H1, db2
    weight_grad<PNW>(RA, h2 >> 5, RB, h1 >> 5, slab + d.oW2(), h1, h1, wave, lane);
    bias_grad<PNW>(RA, h2, slab + d.ob2(), wave, lane);
    PROF(13);

    // ---- objective partial sums (scaled by 1/B so that the slab reduction yields the means)
    const float t0 = block_sum(loss0, s_red);
    const float t1 = block_sum(loss1, s_red);
    if (tid == 0) {
        float *logs = g.slabs + (size_t)blockIdx.x * g.stride + g.Pa + g.Pc;
        if (ACTOR) {
            float ent = 0.f;
            for (int a = 0; a < OUT; ++a) ent += 1.4189385332046727418f + logf(expf(std_log[a]));  // 0.5 + 0.5 log(2 pi) + log(std)
            logs[1] = t0 * g.inv_batch;
            logs[2] = ent * t1 * g.inv_batch;
        } else {
            logs[0] = t0 * g.inv_batch;
            logs[3] = 0.f;
            for (int64_t e = g.Pa + g.Pc + 4; e < g.stride; ++e) logs[e - (g.Pa + g.Pc)] = 0.f;   // the row's pad
        }
    }
}

template <int NS_, int N1_, int N2_, bool VEC>
__global__ __launch_bounds__(PNW * 64) void ppo_step2_kernel(Ppo2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SpanT t_span = span_enter(g);
    if (blockIdx.y == 0) ppo_block<true, NS_, N1_, N2_, VEC>(g, smem);
    else ppo_block<false, NS_, N1_, N2_, VEC>(g, smem);
    span_exit(g, t_span);
}

bool dims_ok2(int S, int h1, int h2, int out)
{
    if (erl_ppo_wd_supported(S, h1, h2, out)) return true;      // net_dims = (256, h2): ppo_step_wd.hip
    return S >= 1 && S <= ERL_MAX_STATE_DIM && h1 >= 32 && h1 <= ERL_MAX_HIDDEN && (h1 % 32) == 0 && h2 >= 32 &&
           h2 <= ERL_MAX_HIDDEN && (h2 % 32) == 0 && out >= 1 && out <= ERL_MAX_ACTION_DIM;
}

template <int NS_, int N1_, int N2_, bool VEC>
int launch(const Ppo2Args &g, int n_slabs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)ppo_step2_kernel<NS_, N1_, N2_, VEC>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPpoLdsBytes),
                                "hipFuncSetAttribute(ppo_step2_kernel)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((ppo_step2_kernel<NS_, N1_, N2_, VEC>), dim3(n_slabs, 2), dim3(PNW * 64), kPpoLdsBytes, stream, g);
    return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32");
}

long long *g_ppo_prof = nullptr;
int g_ppo_prof_block = 0;

}  // namespace

unsigned long long *erl_k6_timing_begin(hipStream_t stream, int n_slabs);   // api.cpp (measurement hook, no-op unless enabled; grid = (n_slabs, 2))
void erl_k6_timing_end(hipStream_t stream);

namespace {
__global__ void null_kernel() {}
}  // namespace
void erl_launch_null_kernel(hipStream_t stream) { hipLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, stream); }

#ifdef ERL_PROFILE
// profiling builds only (make EXTRA=-DERL_PROFILE): device buffer of 2 * 8 * 32 int64 cycle stamps
extern "C" __attribute__((visibility("default"))) void erl_debug_set_ppo_profile(long long *dev_buf) { g_ppo_prof = dev_buf; erl_ppo_wd_set_prof(dev_buf, g_ppo_prof_block); }
extern "C" __attribute__((visibility("default"))) void erl_debug_set_ppo_profile_block(int b) { g_ppo_prof_block = b; erl_ppo_wd_set_prof(g_ppo_prof, b); }
#endif

namespace {
constexpr int kDefaultArith = ERL_PPO_ARITH_SPLIT;     // fp32-equivalent by test (tests/test_kernels_gpu.py::test_ppo_step_split_arith) and ~1.2x faster
int g_k6_arith = ERL_PPO_ARITH_AUTO;
int k6_arith_resolved()
{
    if (g_k6_arith != ERL_PPO_ARITH_AUTO) return g_k6_arith;
    static const int env = [] {
        const char *e = getenv("ERL_K6_ARITH");
        if (e && !strcmp(e, "f32")) return (int)ERL_PPO_ARITH_F32;
        if (e && !strcmp(e, "split")) return (int)ERL_PPO_ARITH_SPLIT;
        return kDefaultArith;
    }();
    return env;
}
int k6_form()     // 0 = automatic, 8 = always the 8-wave 16x16x4 kernel (A/B measurements: ERL_K6_FORM=8)
{
    static const int form = [] { const char *e = getenv("ERL_K6_FORM"); return e ? atoi(e) : 0; }();
    return form;
}
// the minibatch kernels' workgroup map (ppo_step.h, k6_wg_map).  ERL_K6_WG_MAP=0 / 1 / 2 (read per launch: A/B runs flip it inside one
// process) forces a map; otherwise the FIRST full-chip launch of a kernel family on a device measures map 0 against map 2
// (erl_k6_wg_map_for_launch below) and the device keeps the faster one: map 2 saves 9-12 us of 47-55 on about one box in four of the pool
// and is even with map 0 on the rest (DESIGN.md "K6 in round 5").
constexpr int kWgMapDevices = 64;
constexpr int kWgMapAlt = 2;          // the map measured against map 0 (k6_wg_map: one network per pair of shader engines)
struct WgMapChoice {
    int map = -1;                 // -1: not measured yet
    double us[2] = {0.0, 0.0};    // per launch, back to back: [0] map 0, [1] map kWgMapAlt (0 when never measured)
};
constexpr int kWgMapFamilies = 2;
WgMapChoice g_wg_map[kWgMapFamilies][kWgMapDevices];
int k6_wg_map_env()
{
    const char *e = getenv("ERL_K6_WG_MAP");
    return e && (*e == '0' || *e == '1' || *e == '2') ? *e - '0' : -1;
}
}  // namespace

extern "C" int erl_ppo_set_arith(int arith)
{
    const int prev = g_k6_arith;
    if (arith >= ERL_PPO_ARITH_AUTO && arith <= ERL_PPO_ARITH_SPLIT) g_k6_arith = arith;
    return prev;
}

// the arithmetic a call gets: its own request (the mode word's arith bits), else the process-wide default
int erl_ppo_arith_for_call(int S, int h1, int h2, int A, int arith_call)
{
    if (erl_ppo_wd_supported(S, h1, h2, A)) return ERL_PPO_ARITH_SPLIT;      // the (256, h2) kernel exists in this arithmetic only
    const int want = (arith_call == ERL_PPO_ARITH_F32 || arith_call == ERL_PPO_ARITH_SPLIT) ? arith_call : k6_arith_resolved();
    return (k6_form() != 8 && want == ERL_PPO_ARITH_SPLIT && erl_ppo_s3_supported(S, h1, h2, A)) ? ERL_PPO_ARITH_SPLIT : ERL_PPO_ARITH_F32;
}

extern "C" int erl_ppo_arith_in_use(int S, int h1, int h2, int A) { return erl_ppo_arith_for_call(S, h1, h2, A, ERL_PPO_ARITH_AUTO); }

extern "C" int64_t erl_ppo_slab_stride(int S, int h1, int h2, int A)
{
    if (!dims_ok2(S, h1, h2, A)) return -1;
    // Pa + Pc + 4 (actor gradient | critic gradient | 4 logged values), rounded up to 32 floats: every slab then starts on a
    // 128-byte line and the non-temporal gradient stores write whole lines (an unaligned pitch split each one in two partial
    // writes: WRITE_SIZE 37.6 MB per launch for 26.6 MB of slabs).  The pad [Pa + Pc + 4, stride) is written as zeros.
    const int64_t n = Dims{S, h1, h2, A}.count(true) + Dims{S, h1, h2, 1}.count(false) + 4;
    return (n + 31) / 32 * 32;
}

extern "C" int erl_ppo_num_slabs(int64_t B) { return B >= 1 && B < (1LL << 37) ? (int)erl_cdiv(B, PB) : -1; }

// Which workgroup map this launch runs under.  Not forced and not measured yet on this device for this kernel family: launch the kernel
// with the CALL'S OWN arguments under both maps, alternating (1 + 4 launches per leg, 3 legs per map, HIP events on the call's stream; the
// kernel writes nothing but the gradient slabs and its scratch, which the real launch that follows rewrites) and keep map 2 when it is
// at least 3 % faster (map 1 and map 2 recover the same boxes; map 2 costs the others nothing, map 1 ~1.2 us: map 2 is the candidate).  Only a launch that fills the chip (>= 256 workgroups) decides; a capturing stream or a failed event leaves the
// decision to a later call.
int erl_k6_wg_map_for_launch(int family, int n_slabs, hipStream_t st, const std::function<int(int)> &launch)
{
    const int forced = k6_wg_map_env();
    if (forced >= 0) return forced;
    int dev = 0;
    if (family < 0 || family >= kWgMapFamilies || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kWgMapDevices) return 0;
    static std::mutex mu;                              // agents of several devices may step from their own threads
    std::lock_guard<std::mutex> lock(mu);
    WgMapChoice &c = g_wg_map[family][dev];
    if (c.map >= 0) return c.map;
    if (2 * n_slabs < 256) return 0;
    if (getenv("ERL_K6_NO_TUNE")) return 0;            // (diagnostics: no measurement, map 0)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    constexpr int kLegs = 3, kReps = 4;                 // three alternations; map 2 must win EVERY one of them by 3 % (round 6: was one pooled comparison of two)
    hipEvent_t ev[2 * kLegs][2] = {};
    bool ok = true;
    for (auto &e : ev) ok = ok && hipEventCreate(&e[0]) == hipSuccess && hipEventCreate(&e[1]) == hipSuccess;
    for (int leg = 0; ok && leg < 2 * kLegs; ++leg) {
        for (int k = 0; ok && k <= kReps; ++k) {
            if (k == 1) ok = hipEventRecord(ev[leg][0], st) == hipSuccess;
            ok = ok && launch((leg & 1) ? kWgMapAlt : 0) == ERL_OK;
        }
        ok = ok && hipEventRecord(ev[leg][1], st) == hipSuccess;
    }
    double us[2] = {0.0, 0.0}, leg_us[2 * kLegs] = {};
    for (int leg = 0; ok && leg < 2 * kLegs; ++leg) {
        float ms = 0.f;
        ok = hipEventSynchronize(ev[leg][1]) == hipSuccess && hipEventElapsedTime(&ms, ev[leg][0], ev[leg][1]) == hipSuccess;
        leg_us[leg] = (double)ms * 1e3 / kReps;
        us[leg & 1] += leg_us[leg] / kLegs;
    }
    for (auto &e : ev) { if (e[0]) (void)hipEventDestroy(e[0]); if (e[1]) (void)hipEventDestroy(e[1]); }
    if (!ok) { (void)hipGetLastError(); return 0; }
    bool alt_wins = true;
    for (int i = 0; i < kLegs; ++i) alt_wins = alt_wins && leg_us[2 * i + 1] < 0.97 * leg_us[2 * i];
    c.us[0] = us[0]; c.us[1] = us[1];
    c.map = alt_wins ? kWgMapAlt : 0;
    return c.map;
}

// ---- code touch (ppo_step.h, k6_code_touch) ----------------------------------------------------------------------------------------
// comm.cpp announces the first launch of an update loop; the launch that follows consumes the request if the device is one whose
// instruction fetch is slow: the workgroup-map measurement chose map 2 there (ERL_K6_CODE_TOUCH=1 / 0 forces it on / off).
static bool g_touch_request[kWgMapDevices] = {};
void erl_k6_touch_next_launch()
{
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kWgMapDevices) g_touch_request[dev] = true;
}
bool erl_k6_code_touch_wanted(int family)
{
    int dev = 0;
    if (family < 0 || family >= kWgMapFamilies || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kWgMapDevices || !g_touch_request[dev]) return false;
    g_touch_request[dev] = false;
    const char *e = getenv("ERL_K6_CODE_TOUCH");
    if (e && (*e == '0' || *e == '1')) return *e == '1';
    return k6_wg_map_env() < 0 && g_wg_map[family][dev].map == kWgMapAlt;
}
// the loader's allocation around a kernel's program counter (ROCr, resolved at run time: libhsa-runtime64 is in the process under HIP):
// [max(allocation base, pc - 512 rounded down to 256), min(allocation end, + want_bytes)), a multiple of 16 bytes
bool erl_k6_code_range(unsigned long long pc, size_t want_bytes, const unsigned char **base, unsigned *bytes)
{
    using Fn = hsa_status_t (*)(const void *, hsa_amd_pointer_info_t *, void *(*)(size_t), uint32_t *, hsa_agent_t **);
    static const Fn fn = (Fn)dlsym(RTLD_DEFAULT, "hsa_amd_pointer_info");
    *base = nullptr;
    *bytes = 0;
    if (!fn || !pc) return false;
    hsa_amd_pointer_info_t info;
    memset(&info, 0, sizeof(info));
    info.size = sizeof(info);
    if (fn((const void *)pc, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || info.type == HSA_EXT_POINTER_TYPE_UNKNOWN || !info.agentBaseAddress) return false;
    const uintptr_t lo = (uintptr_t)info.agentBaseAddress, hi = lo + info.sizeInBytes;
    uintptr_t b = ((uintptr_t)pc - 512) & ~(uintptr_t)255;
    if (b < lo) b = lo;
    uintptr_t e = b + want_bytes;
    if (e > hi) e = hi;
    if ((uintptr_t)pc < lo || (uintptr_t)pc >= hi || e <= b + 16) return false;
    *base = (const unsigned char *)b;
    *bytes = (unsigned)((e - b) & ~(uintptr_t)15);
    return true;
}

// the map the current device keeps for a kernel family: -1 not measured yet, 0 / 2 (a forced map counts as measured)
int erl_k6_wg_map_choice(int family)
{
    const int forced = k6_wg_map_env();
    if (forced >= 0) return forced;
    int dev = 0;
    if (family < 0 || family >= kWgMapFamilies || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kWgMapDevices) return -1;
    if (getenv("ERL_K6_NO_TUNE")) return 0;
    return g_wg_map[family][dev].map;
}

extern "C" int erl_ppo_wg_map_info(int device, int *map, double *us_map0, double *us_map2)
{
    const int family = (device >> 8) & 0xff;              // ERL_PPO_WG_FAMILY_WIDE
    device &= 0xff;
    ERL_REQUIRE(device >= 0 && device < kWgMapDevices && family < kWgMapFamilies, "erl_ppo_wg_map_info: device %d family %d", device, family);
    const int forced = k6_wg_map_env();
    if (map) *map = forced >= 0 ? forced : g_wg_map[family][device].map;
    if (us_map0) *us_map0 = g_wg_map[family][device].us[0];
    if (us_map2) *us_map2 = g_wg_map[family][device].us[1];
    return ERL_OK;
}

// erl_ppo_step_f32 with the pre-split W2 images of the split-arithmetic kernel (s3_image.h; nullptr: none)
int erl_ppo_step_images_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                            const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, const float *states,
                            const float *actions, const uint8_t *unmasks, const float *logprobs, const float *advantages,
                            const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B, float ratio_clip,
                            float lambda_entropy, float inv_batch, int objective, float *slabs, int n_slabs, const S3Images *images,
                            const double *adv_stats, const int64_t *next_ids, void *stream, int only_net)
{
    const int arith_call = (objective >> 8) & 3;       // ERL_PPO_MODE(objective, arith): the call's own arithmetic (0: process default)
    objective &= 0xff;
    ERL_REQUIRE(actor_params && critic_params && act_avg && act_std && cri_avg && cri_std && states && actions && unmasks &&
                    logprobs && advantages && reward_sums && ids && slabs,
                "erl_ppo_step_f32: NULL tensor");
    ERL_REQUIRE(dims_ok2(S, h1, h2, A), "erl_ppo_step_f32: unsupported dims S=%d net=[%d,%d] A=%d", S, h1, h2, A);
    ERL_REQUIRE(H >= 1 && N >= 1 && B >= 1, "erl_ppo_step_f32: bad shape");
    ERL_REQUIRE(objective >= ERL_PPO_OBJ_REFERENCE && objective <= ERL_PPO_OBJ_A2C, "erl_ppo_step_f32: unknown objective %d", objective);
    ERL_REQUIRE(n_slabs == erl_ppo_num_slabs(B), "erl_ppo_step_f32: n_slabs=%d, expected erl_ppo_num_slabs(B=%lld)=%d", n_slabs,
                (long long)B, erl_ppo_num_slabs(B));
    ERL_REQUIRE(only_net < 0 || (only_net <= 1 && !erl_ppo_wd_supported(S, h1, h2, A) &&
                                 erl_ppo_arith_for_call(S, h1, h2, A, arith_call) == ERL_PPO_ARITH_SPLIT),
                "erl_ppo_step_f32: a one-network launch exists for the split-arithmetic (128 | 64, h2) kernels only");
    if (erl_ppo_wd_supported(S, h1, h2, A))
        return erl_ppo_wd_step(actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, states, actions, unmasks, logprobs,
                               advantages, reward_sums, H, N, ids, B, ratio_clip, lambda_entropy, inv_batch, objective, slabs, n_slabs,
                               erl_ppo_slab_stride(S, h1, h2, A), images, adv_stats, stream);
    Ppo2Args g;
    g.P[0] = actor_params; g.P[1] = critic_params;
    g.avg[0] = act_avg; g.avg[1] = cri_avg;
    g.sd[0] = act_std; g.sd[1] = cri_std;
    g.states = states; g.actions = actions; g.logprobs = logprobs; g.advantages = advantages; g.reward_sums = reward_sums;
    g.unmasks = unmasks; g.ids = ids;
    g.H = H; g.N = N; g.B = B;
    g.S = S; g.h1 = h1; g.h2 = h2; g.A = A;
    g.ratio_clip = ratio_clip; g.lambda_entropy = lambda_entropy; g.inv_batch = inv_batch;
    g.objective = objective;
    g.slabs = slabs;
    g.Pa = Dims{S, h1, h2, A}.count(true);
    g.Pc = Dims{S, h1, h2, 1}.count(false);
    g.stride = erl_ppo_slab_stride(S, h1, h2, A);
    g.adv_stats = adv_stats;
    g.next_ids = next_ids;
    g.wg_map = 0;
    g.exp_net = [] { const char *e = getenv("ERL_K6_ONLY_NET"); return e ? atoi(e) : -1; }();      // (diagnostic builds read it)
    g.w2img[0] = images ? images->net[0].img : nullptr;
    g.w2img[1] = images ? images->net[1].img : nullptr;
    g.w1img[0] = images ? images->net[0].img1 : nullptr;
    g.w1img[1] = images ? images->net[1].img1 : nullptr;
    g.aux = images ? images->aux : nullptr;
    g.prof = g_ppo_prof;
    g.prof_block = g_ppo_prof_block;
    // 16-byte vector path: every row / parameter block / normalisation vector must be 16-byte aligned
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (S % 4 == 0) && al(actor_params) && al(critic_params) && al(states) && al(act_avg) && al(act_std) &&
                     al(cri_avg) && al(cri_std);
    hipStream_t st = (hipStream_t)stream;
    const int ns = (S + 15) / 16;
    const bool split = erl_ppo_arith_for_call(S, h1, h2, A, arith_call) == ERL_PPO_ARITH_SPLIT;
    const bool pre = g.w2img[0] && g.w2img[1] && g.w1img[0] && g.w1img[1];
    g.span = nullptr;
    g.only_net = only_net;
    if (split && only_net < 0)                                                   // (the first full-chip launch on a device measures both maps)
        g.wg_map = erl_k6_wg_map_for_launch(0, n_slabs, st, [&](int m) {
            Ppo2Args t = g;
            t.wg_map = m;
            return pre ? erl_ppo_s3_launch_pre(t, n_slabs, vec, st) : erl_ppo_s3_launch(t, n_slabs, vec, st);
        });
    if (split && only_net < 0 && erl_k6_code_touch_wanted(0)) g.code_touch_bytes = 1;          // (a request: launch_s3 resolves the range of its instantiation)
    g.span = erl_k6_timing_begin(st, n_slabs);
    int rc;
    // K6 form: 0 = automatic (one-wave-per-SIMD kernels where their shape classes apply: the split-bf16 one if selected, else
    // the fp32 32x32x2 one), 8 = always the 8-wave 16x16x4 kernel
    const int form = k6_form();
    if (split) rc = pre ? erl_ppo_s3_launch_pre(g, n_slabs, vec, st) : erl_ppo_s3_launch(g, n_slabs, vec, st);
    else if (form != 8 && erl_ppo_w4_supported(S, h1, h2, A)) rc = erl_ppo_w4_launch(g, n_slabs, vec, st);   // configs 2 / 4 / 5
    else if (vec && ns == 4 && h1 == 128 && h2 == 128) rc = launch<4, 8, 8, true>(g, n_slabs, st);
    else if (vec) rc = launch<0, 0, 0, true>(g, n_slabs, st);
    else rc = launch<0, 0, 0, false>(g, n_slabs, st);
    erl_k6_timing_end(st);
    return rc;
}

extern "C" int erl_ppo_step_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                                const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, const float *states,
                                const float *actions, const uint8_t *unmasks, const float *logprobs, const float *advantages,
                                const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B, float ratio_clip,
                                float lambda_entropy, float inv_batch, int objective, float *slabs, int n_slabs, void *stream)
{
    return erl_ppo_step_images_f32(actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, states, actions, unmasks,
                                   logprobs, advantages, reward_sums, H, N, ids, B, ratio_clip, lambda_entropy, inv_batch, objective, slabs,
                                   n_slabs, nullptr, nullptr, nullptr, stream);
}
