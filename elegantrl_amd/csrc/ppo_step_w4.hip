// K6, one-wave-per-SIMD form: one PPO minibatch (gather + actor & critic forward + objective + full backward) for the
// BASELINE shape class  S <= 64, net [128, 128], A <= 8  (configs 4 / 5).  gfx950 fp32 MFMA.
//
// Replaces AgentPPO.update_objectives up to the optimizer steps (elegantrl/agents/AgentPPO.py:173-204) and
// ActorPPO.get_logprob_entropy (:378-386), like ppo_step.hip, and writes the same slabs.  What differs is the mapping:
//
//   * grid = (ceil(B / 128), 2 nets), 256 threads: FOUR waves, one per SIMD, each owning 32 samples.  A wave alone on
//     its SIMD has the whole 512-entry register file (256 arch + 256 acc VGPRs): X, H1, GELU'(z1), H2, GELU'(z2) of its
//     32 samples (288 registers per lane) stay in registers from the gather to the last weight gradient -- nothing is
//     parked in memory (the 8-wave kernel round-trips GELU'(z1) through its slab: 33.6 MB per launch at B = 16384).
//   * every layer is computed transposed on v_mfma_f32_32x32x2_f32 (64 cycles per SIMD, dependent-accumulate latency
//     64: a single accumulator chain keeps the pipe full, which the 16x16x4 shape -- 32-cycle issue, 40-cycle dependent
//     latency -- cannot do without a second wave):
//         outT (32 features x 32 samples) += W (32 rows x 2 k) . inT (2 k x 32 samples).
//     The result tile leaves lane (m = lane & 31, hi = lane >> 5) holding features 8 g + 4 hi + j (acc[4 g + j]) of
//     sample m.  The next layer walks its reduction index in the order (tile, g, j) and pairs k = 8 g + j (lane half 0)
//     with k = 8 g + 4 + j (lane half 1): the B operand of step (g, j) is then exactly acc[4 g + j] of the previous
//     layer -- the register chain of ppo_step.hip in the 32x32 layout -- and the A operand of four consecutive steps is
//     one 16-byte LDS read W[row][32 T + 8 g + 4 hi .. + 3].  Half the LDS operand traffic of the 16x16x4 form.
//   * the two instruction streams a SIMD used to interleave (two waves) are one stream here: the GELU epilogue of an
//     output tile is issued between the MFMAs of the next tile by the compiler's scheduler (every layer is one fully
//     unrolled basic block).
//
// Weight gradients are the same staged scheme as ppo_step.hip (T[feature][sample] tiles in LDS, 32x32x2 tiles, K = 128
// samples, output tiles split over the waves).
#include "ppo_step.h"

namespace {

constexpr int QNW = 4;           // waves per workgroup
constexpr int QNT = QNW * 64;

// LDS pool (floats): [RA: W2 copy, later staged tiles][RB: W1 copy | X^T, later staged tiles][RC: dY^T][RW3: W3 copy]
//                    [s_bias: b1 | b2 | b3(16)][s_part: 4*16][s_red: 16]
constexpr int kQR = 128 * 68 + 64 * PLD;                  // >= 128 * PLD
static_assert(kQR >= 128 * PLD && kQR % 4 == 0, "staged tiles must fit the weight-copy regions");
constexpr int kQRC = 16 * PLD;
constexpr int kQRW3 = 16 * 132;
constexpr int kQBias = 128 + 128 + 16;
constexpr size_t kW4LdsBytes = (size_t)(2 * kQR + kQRC + kQRW3 + kQBias + QNW * 16 + 16) * sizeof(float);
static_assert(kW4LdsBytes <= 160 * 1024, "LDS budget");

// ---------------------------------------------------------------------------------------------------------
// Instruction-stream control.  One wave per SIMD: whatever has to hide under the MFMAs must sit BETWEEN them in program
// order (an in-order wave stalls at the next dependent MFMA).  Two rules shape every MFMA phase of this kernel:
//   1. An MFMA that accumulates into the register its predecessor wrote only issues back-to-back when NOTHING sits
//      between the two (accumulator forwarding); one VALU op, s_waitcnt or s_nop in between costs ~40 cycles (measured:
//      the first version of this kernel ran its layers at 60-65 % of the MFMA rate with the epilogue ops between
//      dependent MFMAs).  So every reduction is split over TWO accumulator chains that alternate instruction by
//      instruction (chain 0: even reduction groups, chain 1: odd groups, summed at the end): consecutive MFMAs are
//      independent, and each 64-cycle gap can hide ~12 other instructions.
//   2. The compiler must neither hoist all the operand reads of an unrolled layer (that alone overflowed the 512-entry
//      register file) nor sink the epilogues behind the MFMA chain: every "super-group" (two 16-byte A-operand reads
//      issued one super-group ahead, eight alternating MFMAs, a share of the previous output tile's epilogue) is fenced
//      by sched_barrier, ordered inside by sched_group_barrier, and epilogue results are pinned (ERL_PIN) because LLVM's
//      IR-level code sinking moves pure arithmetic across sched_barrier towards its first use.
// ---------------------------------------------------------------------------------------------------------
#define ERL_PIN1(a) asm volatile("" : "+v"(a))
#define ERL_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define ERL_SGB_DSREAD(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define ERL_SGB_MFMA_VALU(nv)                                \
    do {                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x002, (nv), 0); \
    } while (0)
#define ERL_SGB_8X(nv)                                                                                     \
    do {                                                                                                   \
        ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv);        \
        ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv); ERL_SGB_MFMA_VALU(nv);        \
    } while (0)

// the two partial accumulators of an output tile whose epilogue is still to be done (z = a0 + a1, bias inside a0)
struct Pend {
    f32x16 a0, a1;
};

// bias of output tile To in the D layout: element e <-> feature 32 To + 8 (e >> 2) + 4 hi + (e & 3).
// HEAD16: the bias vector has 16 entries only (output layer): elements 4.. are padding rows and start from 0.
template <bool HEAD16>
__device__ __forceinline__ void load_bias16(const float *bias, int To, int hi, f32x16 &pb)
{
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        if (HEAD16 && gq > 0) {
            pb[4 * gq + 0] = 0.f; pb[4 * gq + 1] = 0.f; pb[4 * gq + 2] = 0.f; pb[4 * gq + 3] = 0.f;
        } else {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * To + 8 * gq + 4 * hi);
            pb[4 * gq + 0] = b4.x; pb[4 * gq + 1] = b4.y; pb[4 * gq + 2] = b4.z; pb[4 * gq + 3] = b4.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward layer on registers: out[To] (32 features x 32 samples) = GELU( bias + W[32 To .. + 31][:] . in ), To < NOUT.
// W, bias: zero-padded LDS copies (row stride ldw = 4 * odd floats: the 16-byte reads of 16 consecutive rows hit 16
// distinct 16-byte bank groups); `arow` = this lane's row inside a 32-row tile.  KT = input tiles of 32.
// Software pipeline across tiles AND layers: the GELU epilogue of tile To rides in the super-groups of tile To + 1; the
// last tile's accumulators are handed to the caller (`pout`), whose next layer finishes them (`pin`, PEND_IN) during
// the first super-groups of its own first output tile -- that tile consumes the pending input tile KT - 1 last.
// ---------------------------------------------------------------------------------------------------------
template <int KT, bool PEND_IN, int NOUT, bool HEAD16>
__device__ __forceinline__ void fwd32(const float *W, int ldw, const float *bias, int arow, f32x16 (&in)[KT], f32x16 (&inG)[KT],
                                      const Pend &pin, f32x16 (&outH)[4], f32x16 (&outG)[4], Pend &pout, int hi)
{
    constexpr int NS = 2 * KT, NC = NOUT * NS;                 // super-groups per output tile / in total
    constexpr int EPS = (16 + NS - 1) / NS;                    // epilogue elements of the previous tile per super-group
    constexpr int NPS = NS > 2 ? NS - 2 : 1;                   // the pending tile is consumed by the last two super-groups
    constexpr int PPS = (16 + NPS - 1) / NPS;                  // pending-input elements per super-group
    constexpr int NV = (24 * ((PEND_IN && PPS > EPS) || NOUT == 1 ? PPS : EPS) + 7) / 8;   // VALU slots per MFMA gap
    const float *wbase = W + arow * ldw + 4 * hi;
    float4 wq[2][2];
    auto issue = [&](int c, float4(&dst)[2]) {
        const int To = c / NS, s = c % NS;
        const float *p = wbase + 32 * To * ldw + 16 * s;
        dst[0] = *reinterpret_cast<const float4 *>(p);
        dst[1] = *reinterpret_cast<const float4 *>(p + 8);
    };
    issue(0, wq[0]);
    f32x16 nb;
    load_bias16<HEAD16>(bias, 0, hi, nb);
    f32x16 acc0 = {0}, acc1 = {0}, p0 = {0}, p1 = {0};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int To = c / NS, s = c % NS;
        const int Ti0 = (2 * s) >> 2, g0 = (2 * s) & 3, Ti1 = (2 * s + 1) >> 2, g1 = (2 * s + 1) & 3;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        if (s == 0) {
            acc0 = nb;
            acc1 = f32x16{0};
        }
        const float4 a0 = wq[c & 1][0], a1 = wq[c & 1][1];
        acc0 = mfma32(a0.x, in[Ti0][4 * g0 + 0], acc0);
        acc1 = mfma32(a1.x, in[Ti1][4 * g1 + 0], acc1);
        acc0 = mfma32(a0.y, in[Ti0][4 * g0 + 1], acc0);
        acc1 = mfma32(a1.y, in[Ti1][4 * g1 + 1], acc1);
        acc0 = mfma32(a0.z, in[Ti0][4 * g0 + 2], acc0);
        acc1 = mfma32(a1.z, in[Ti1][4 * g1 + 2], acc1);
        acc0 = mfma32(a0.w, in[Ti0][4 * g0 + 3], acc0);
        acc1 = mfma32(a1.w, in[Ti1][4 * g1 + 3], acc1);
        if (To > 0) {                                          // epilogue share of tile To - 1
#pragma unroll
            for (int u = 0; u < EPS; ++u) {
                const int e = s * EPS + u;
                if (e < 16) {
                    float y, gd;
                    gelu_and_grad_fast(p0[e] + p1[e], y, gd);
                    ERL_PIN2(y, gd);
                    outH[To - 1][e] = y;
                    outG[To - 1][e] = gd;
                }
            }
        } else if (PEND_IN && s < NPS) {                       // the producer layer's last tile
#pragma unroll
            for (int u = 0; u < PPS; ++u) {
                const int e = s * PPS + u;
                if (e < 16) {
                    float y, gd;
                    gelu_and_grad_fast(pin.a0[e] + pin.a1[e], y, gd);
                    ERL_PIN2(y, gd);
                    in[KT - 1][e] = y;
                    inG[KT - 1][e] = gd;
                }
            }
        }
        if (s == NS - 1 && To + 1 < NOUT) load_bias16<HEAD16>(bias, To + 1, hi, nb);
        ERL_SGB_DSREAD(2);
        ERL_SGB_8X(NV);
        __builtin_amdgcn_sched_barrier(0);
        if (s == NS - 1) {
            p0 = acc0;
            p1 = acc1;
        }
    }
    pout.a0 = p0;
    pout.a1 = p1;
}

// finish a pending tile outside any MFMA phase (S <= 32: the next layer is too short to hide it)
__device__ __forceinline__ void finish_pend(const Pend &p, f32x16 &H, f32x16 &G)
{
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float y, gd;
        gelu_and_grad_fast(p.a0[e] + p.a1[e], y, gd);
        H[e] = y;
        G[e] = gd;
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward through a layer's input on registers:  gate[To] <- gate[To] * ( W^T . dz ),  W = LDS copy [32 KT rows][ldw]
// (A operand = W^T: lane (i, hi) supplies W[8 gi + 4 hi + j][32 To + i] for reduction group gi, four ds_read_b32 per
// group, issued one super-group ahead; two alternating accumulator chains; the gate multiplies of tile To ride in the
// first super-group of tile To + 1).
// ---------------------------------------------------------------------------------------------------------
template <int KT>
__device__ __forceinline__ void bwd32(const float *W, int ldw, const f32x16 (&dz)[KT], f32x16 (&gate)[4], int m, int hi)
{
    constexpr int NS = 2 * KT, NC = 4 * NS;
    const float *wbase = W + (4 * hi) * ldw + m;
    float wq[2][8];
    auto issue = [&](int c, float(&dst)[8]) {
        const int To = c / NS, s = c % NS;
        const float *p = wbase + (16 * s) * ldw + 32 * To;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dst[u] = p[u * ldw];
            dst[4 + u] = p[(8 + u) * ldw];
        }
    };
    issue(0, wq[0]);
    f32x16 acc0 = {0}, acc1 = {0}, p0 = {0}, p1 = {0};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int To = c / NS, s = c % NS;
        const int Tk0 = (2 * s) >> 2, g0 = (2 * s) & 3, Tk1 = (2 * s + 1) >> 2, g1 = (2 * s + 1) & 3;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        if (s == 0) {
            acc0 = f32x16{0};
            acc1 = f32x16{0};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = mfma32(wq[c & 1][j], dz[Tk0][4 * g0 + j], acc0);
            acc1 = mfma32(wq[c & 1][4 + j], dz[Tk1][4 * g1 + j], acc1);
        }
        if (To > 0 && s == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = gate[To - 1][e] * (p0[e] + p1[e]);
                ERL_PIN1(v);
                gate[To - 1][e] = v;
            }
        }
        ERL_SGB_DSREAD(8);
        ERL_SGB_8X(4);
        __builtin_amdgcn_sched_barrier(0);
        if (s == NS - 1) {
            p0 = acc0;
            p1 = acc1;
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) gate[3][e] *= p0[e] + p1[e];
}

// dW (nA32*32 x nB32*32) = TA . TB^T over the 128 staged samples, output tiles split over the waves.  The sum over
// samples is order-free: lane half `hi` takes samples 8 j + 4 hi + {0..3} of every group of 8 (one 16-byte read per
// operand feeds four MFMAs), even groups feed accumulator chain 0 and odd groups chain 1 (alternating, see rule 1).
__device__ __forceinline__ void weight_grad_w4(const float *TA, int nA32, const float *TB, int nB32, float *__restrict__ dW,
                                               int ldw, int cols_real, int wave, int lane)
{
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntiles = nA32 * nB32;
    for (int tile = wave; tile < ntiles; tile += QNW) {
        const int it = tile / nB32, jt = tile - it * nB32;
        const float *a4 = TA + (32 * it + l31) * PLD + 4 * hi;
        const float *b4 = TB + (32 * jt + l31) * PLD + 4 * hi;
        f32x16 acc0 = {0}, acc1 = {0};
        float4 av[2][2], bv[2][2];
        auto issue = [&](int pr, float4(&a)[2], float4(&b)[2]) {
            a[0] = *reinterpret_cast<const float4 *>(a4 + 16 * pr);
            b[0] = *reinterpret_cast<const float4 *>(b4 + 16 * pr);
            a[1] = *reinterpret_cast<const float4 *>(a4 + 16 * pr + 8);
            b[1] = *reinterpret_cast<const float4 *>(b4 + 16 * pr + 8);
        };
        issue(0, av[0], bv[0]);
#pragma unroll
        for (int pr = 0; pr < PB / 16; ++pr) {
            if (pr + 1 < PB / 16) issue(pr + 1, av[(pr + 1) & 1], bv[(pr + 1) & 1]);
            const float4 x0 = av[pr & 1][0], y0 = bv[pr & 1][0], x1 = av[pr & 1][1], y1 = bv[pr & 1][1];
            acc0 = mfma32(x0.x, y0.x, acc0);
            acc1 = mfma32(x1.x, y1.x, acc1);
            acc0 = mfma32(x0.y, y0.y, acc0);
            acc1 = mfma32(x1.y, y1.y, acc1);
            acc0 = mfma32(x0.z, y0.z, acc0);
            acc1 = mfma32(x1.z, y1.z, acc1);
            acc0 = mfma32(x0.w, y0.w, acc0);
            acc1 = mfma32(x1.w, y1.w, acc1);
            ERL_SGB_DSREAD(4);
            ERL_SGB_8X(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int i = 32 * jt + l31;
        if (i < cols_real) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dW[(size_t)(32 * it + crow(r, hi)) * ldw + i] = acc0[r] + acc1[r];
        }
    }
}

// stage a register-resident activation (32x32 D layout) feature-major into LDS: T[feature][sample col]
template <int NT_>
__device__ __forceinline__ void stage32(float *T, const f32x16 (&a)[NT_], int col, int hi)
{
#pragma unroll
    for (int t = 0; t < NT_; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[(32 * t + crow(r, hi)) * PLD + col] = a[t][r];
    }
}

template <bool ACTOR, int KX, bool VEC>
__device__ __forceinline__ void ppo_block_w4(const Ppo2Args &g, float *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int h1 = 128, h2 = 128;
    const int S = g.S, OUT = ACTOR ? g.A : 1;
    const Dims d{S, h1, h2, OUT};
    const float *P = g.P[net];
    const float *std_log = P + d.oStd();

    float *RA = smem;                      // W2 copy [128][ld2], later staged tiles [128][PLD]
    float *RB = RA + kQR;                  // W1 copy [128][ld1] | X^T [32 KX][PLD], later staged tiles
    float *RC = RB + kQR;                  // [16][PLD]   dY^T
    float *RW3 = RC + kQRC;                // W3 copy [16][ld3] (rows >= OUT are zero)
    float *s_b1 = RW3 + kQRW3, *s_b2 = s_b1 + 128, *s_b3 = s_b2 + 128;
    float *s_part = s_b3 + 16;             // [4 waves][16]  per-wave dstd_log partials
    float *s_red = s_part + QNW * 16;      // [16] block_sum scratch
    constexpr int ld1 = lds_ld(32 * KX), ld2 = lds_ld(128), ld3 = lds_ld(128);
    float *RX = RB + 128 * lds_ld(64);

    PROF(0);
    // ---- prologue.  Trip 1: the sample id, W1 and the biases.
    const int col = 32 * wave + m;                         // sample slot inside the workgroup
    const int64_t bidx = (int64_t)blockIdx.x * PB + col;
    const bool valid = bidx < g.B;
    const int64_t id = g.ids[valid ? bidx : 0];
    float4 c1[4 * KX];
    copy_load<VEC, 4 * KX, QNT>(c1, P + d.oW1(), h1, S, h1, 32 * KX, tid);
    float bias_pre = (tid < 128) ? P[d.ob1() + tid] : P[d.ob2() + tid - 128];
    float b3_pre = 0.f;
    if (tid < 16) b3_pre = (tid < OUT) ? P[d.ob3() + tid] : 0.f;

    // ---- trip 2: id -> (t = id % H, n = id // H) -> buffer row t*N + n  (AgentPPO.py:179-187) and its data
    int64_t n_, t_;
    if (g.H * g.N <= 0x7fffffffLL) {       // uniform branch: ids < H N fit 32 bits (a 32-bit divide is ~4x shorter)
        const uint32_t i32 = (uint32_t)id, h32 = (uint32_t)g.H, n32 = i32 / h32;
        n_ = n32;
        t_ = i32 - n32 * h32;
    } else {
        n_ = id / g.H;
        t_ = id - n_ * g.H;
    }
    const int64_t row = valid ? t_ * g.N + n_ : 0;
    const float *srow = g.states + row * S;
    const float *avg = g.avg[net], *sdv = g.sd[net];
    // this sample's raw state slice, features 32 T + 8 g + 4 hi + j
    float4 XR[4 * KX];
#pragma unroll
    for (int t = 0; t < 4 * KX; ++t) XR[t] = load4<VEC>(srow, 8 * t + 4 * hi, S);
    // per-sample scalars (consumed after the output layer)
    const float um = (valid && g.unmasks[row]) ? 1.f : 0.f;
    const float xa = ACTOR ? g.logprobs[row] : g.reward_sums[row];
    const float xb = ACTOR ? g.advantages[row] : 0.f;
    float act_pre[4] = {0.f, 0.f, 0.f, 0.f}, sl_pre[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's actions a = 4 hi + j (actor)
    if (ACTOR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ac = min(4 * hi + j, OUT - 1);
            act_pre[j] = g.actions[row * OUT + ac];
            sl_pre[j] = std_log[ac];
        }
    }
    // ---- W2, W3 ride behind: they are not needed before the second layer
    float4 c2[16], c3[2];
    copy_load<VEC, 16, QNT>(c2, P + d.oW2(), h2, h1, h2, h1, tid);
    copy_load<VEC, 2, QNT>(c3, P + d.oW3(), OUT, h2, 16, h2, tid);

    // ---- publish the W1 copy, the biases and X^T (zero padded to the tile grid), visible after barrier (0a)
    copy_store<4 * KX, QNT>(c1, RB, ld1, h1, 32 * KX, tid);
    s_b1[tid] = bias_pre;                                   // s_b1 | s_b2 contiguous
    if (tid < 16) s_b3[tid] = b3_pre;
    f32x16 X[KX];
#pragma unroll
    for (int t = 0; t < 4 * KX; ++t) {                      // (x - avg) / (std + 1e-4)   (AgentPPO.py:360-361)
        const int k0 = 8 * t + 4 * hi;
        const float4 a4 = load4<VEC>(avg, k0, S), s4 = load4<VEC>(sdv, k0, S);
        const float rr[4] = {XR[t].x, XR[t].y, XR[t].z, XR[t].w}, aa[4] = {a4.x, a4.y, a4.z, a4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xn = (rr[j] - aa[j]) / (ss[j] + 1e-4f);
            X[t >> 2][4 * (t & 3) + j] = (valid && k0 + j < S) ? xn : 0.f;
        }
    }
    stage32<KX>(RX, X, col, hi);
    PROF_NV(1);
    lds_barrier();                                                   // (0a) W1 copy, biases visible
    PROF_NV(2);
    f32x16 H1[4], G1[4], H2[4], G2[4];
    Pend pend1, pend2, pendY;
    fwd32<KX, false, 4, false>(RB, ld1, s_b1, m, X, X, pend1, H1, G1, pend1, hi);          // H1[3] left pending
    PROF_NV(3);
    copy_store<16, QNT>(c2, RA, ld2, h2, h1, tid);
    copy_store<2, QNT>(c3, RW3, ld3, 16, h2, tid);
    lds_barrier();                                                   // (0b) W2, W3 copies visible
    fwd32<4, true, 4, false>(RA, ld2, s_b2, m, H1, G1, pend1, H2, G2, pend2, hi);           // finishes H1[3]; H2[3] pending
    PROF(4);
    // ---- output layer: one 32-row tile whose rows >= A are zero (lanes m >= 16 read row 15 of the 16-row copy, which is
    // zero because A <= 8); finishes H2[3] on the way.  No activation: Y = a0 + a1, outputs a = 4 hi + j in elements 0..3.
    fwd32<4, true, 1, true>(RW3, ld3, s_b3, m < 16 ? m : 15, H2, G2, pend2, H2, G2, pendY, hi);
    float Y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Y[j] = pendY.a0[j] + pendY.a1[j];
    PROF(5);

    // ---- objective and dL/dY for this lane's outputs a = 4 hi + j   (AgentPPO.py:189-204)
    float dY[4] = {0.f, 0.f, 0.f, 0.f};
    float loss0 = 0.f, loss1 = 0.f;
    float dsl[4] = {0.f, 0.f, 0.f, 0.f};
    if (!ACTOR) {
        const float diff = Y[0] - xa;                  // only (hi = 0, j = 0) is the value head
        const bool head = hi == 0;
        loss0 = head ? diff * diff * um : 0.f;
        dY[0] = head ? 2.f * diff * um * g.inv_batch : 0.f;
    } else {
        // Normal(mean, exp(std_log)).log_prob(a) = -(a - mean)^2 / (2 var) - log(std) - log(sqrt(2 pi))  with log(std) = std_log
        // and 1 / var = exp(-2 std_log): hardware exp2 / no division (the library expf / logf / IEEE divisions of the
        // 8-wave kernel are ~400 instructions per lane, which no second wave hides here).
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
        const float ratio = __expf(lp - xa);
        float surr, dsurr;
        ppo_surrogate(xb, ratio, g.ratio_clip, g.canonical, surr, dsurr);
        surr = valid ? surr : 0.f;                          // padding rows contribute 0
        dsurr = valid ? dsurr : 0.f;
        if (hi == 0) {
            loss0 = surr * um;
            loss1 = um;
        }
        const float dlp = -(dsurr * um) * g.inv_batch;      // d(-mean(surr um)) / dlogp_new
        const float ent_term = g.lambda_entropy * um * g.inv_batch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = 4 * hi + j < OUT;
            dY[j] = on ? dlp * (diffv[j] * ivar[j]) : 0.f;                                     // dL/dmean
            dsl[j] = on ? dlp * (diffv[j] * diffv[j] * ivar[j] - 1.f) + ent_term : 0.f;        // dL/dstd_log, this sample
        }
    }

    // ---- per-wave dstd_log partials (sum over the wave's 32 samples)
    if (ACTOR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = dsl[j];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 16, 64);
            if (m == 0) s_part[wave * 16 + 4 * hi + j] = s;
        }
    }
    // ---- dZ2 = (W3^T dY) * GELU'(z2)  (K = 8 outputs: four k-pairs);  dZ1 = (W2^T dZ2) * GELU'(z1)
    PROF(6);
    {
        float w3[4][4];
#pragma unroll
        for (int To = 0; To < 4; ++To) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w3[To][j] = RW3[(4 * hi + j) * ld3 + 32 * To + m];
        }
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
        for (int j = 0; j < 4; ++j) {                               // four independent chains, interleaved
#pragma unroll
            for (int To = 0; To < 4; ++To) acc[To] = mfma32(w3[To][j], dY[j], acc[To]);
        }
#pragma unroll
        for (int To = 0; To < 4; ++To) {
#pragma unroll
            for (int r = 0; r < 16; ++r) G2[To][r] *= acc[To][r];
        }
    }
    bwd32<4>(RA, ld2, G2, G1, m, hi);                               // G1 (the gate) <- dZ1
    PROF(7);
    lds_barrier();                                                   // (1) every wave is done with the weight copies
    PROF(8);

    float *slab = g.slabs + (size_t)blockIdx.x * g.stride + (ACTOR ? 0 : g.Pa);
    // ---- layer 1: dW1 = dZ1^T . X, db1;  (dY^T is staged alongside for the output layer)
    stage32<4>(RA, G1, col, hi);                                    // dZ1^T
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        RC[(4 * hi + j) * PLD + col] = dY[j];
        RC[(8 + 4 * hi + j) * PLD + col] = 0.f;
    }
    lds_barrier();                                                   // (2)
    PROF(9);
    weight_grad_w4(RA, 4, RX, (S + 31) >> 5, slab + d.oW1(), S, S, wave, lane);
    bias_grad<QNW>(RA, h1, slab + d.ob1(), wave, lane);
    if (wave == 0) {
        bias_grad<QNW>(RC, OUT, slab + d.ob3(), 0, lane);
        if (ACTOR && lane < OUT) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < QNW; ++u) s += s_part[u * 16 + lane];
            slab[d.oStd() + lane] = s;
        }
    }
    PROF(10);
    lds_barrier();                                                   // (3) dZ1^T, X^T consumed

    // ---- output layer: dW3 (16 x h2) = dY^T . H2 on 16x16x4 MFMA, 16-column tiles split over the waves
    stage32<4>(RA, H2, col, hi);                                    // H2^T
    stage32<4>(RB, H1, col, hi);                                    // H1^T (for dW2)
    lds_barrier();                                                   // (4)
    PROF(11);
    {
        const int l15 = lane & 15, q = lane >> 4;
        for (int it = wave; it < 8; it += QNW) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float *a = RC + l15 * PLD + 4 * q;                    // lane group q: samples 16 j + 4 q + {0..3}
            const float *b = RA + (16 * it + l15) * PLD + 4 * q;
#pragma unroll
            for (int j = 0; j < PB / 16; j += 2) {                      // two alternating chains (16x16x4: 40-cycle dependent latency)
                const float4 av = *reinterpret_cast<const float4 *>(a + 16 * j), bv = *reinterpret_cast<const float4 *>(b + 16 * j);
                const float4 aw = *reinterpret_cast<const float4 *>(a + 16 * j + 16), bw = *reinterpret_cast<const float4 *>(b + 16 * j + 16);
                acc = mfma16(av.x, bv.x, acc);
                acc1 = mfma16(aw.x, bw.x, acc1);
                acc = mfma16(av.y, bv.y, acc);
                acc1 = mfma16(aw.y, bw.y, acc1);
                acc = mfma16(av.z, bv.z, acc);
                acc1 = mfma16(aw.z, bw.z, acc1);
                acc = mfma16(av.w, bv.w, acc);
                acc1 = mfma16(aw.w, bw.w, acc1);
            }
            acc += acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a_ = 4 * q + r;
                if (a_ < OUT) slab[d.oW3() + (size_t)a_ * h2 + 16 * it + l15] = acc[r];
            }
        }
    }
    lds_barrier();                                                   // (5) H2^T consumed
    stage32<4>(RA, G2, col, hi);                                    // dZ2^T
    lds_barrier();                                                   // (6)
    PROF(12);

    // ---- layer 2: dW2 = dZ2^T . H1, db2
    weight_grad_w4(RA, 4, RB, 4, slab + d.oW2(), h1, h1, wave, lane);
    bias_grad<QNW>(RA, h2, slab + d.ob2(), wave, lane);
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
        }
    }
}

template <int KX, bool VEC>
__global__ __launch_bounds__(QNT) void ppo_step_w4_kernel(Ppo2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.y == 0) ppo_block_w4<true, KX, VEC>(g, smem);
    else ppo_block_w4<false, KX, VEC>(g, smem);
}

template <int KX, bool VEC>
int launch_w4(const Ppo2Args &g, int n_slabs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)ppo_step_w4_kernel<KX, VEC>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kW4LdsBytes),
                                "hipFuncSetAttribute(ppo_step_w4_kernel)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((ppo_step_w4_kernel<KX, VEC>), dim3(n_slabs, 2), dim3(QNT), kW4LdsBytes, stream, g);
    return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32");
}

}  // namespace

bool erl_ppo_w4_supported(int S, int h1, int h2, int A) { return S >= 1 && S <= 64 && h1 == 128 && h2 == 128 && A >= 1 && A <= 8; }

int erl_ppo_w4_launch(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream)
{
    if (g.S > 32) return vec ? launch_w4<2, true>(g, n_slabs, stream) : launch_w4<2, false>(g, n_slabs, stream);
    return vec ? launch_w4<1, true>(g, n_slabs, stream) : launch_w4<1, false>(g, n_slabs, stream);
}
