// K6, one-wave-per-SIMD form: one PPO minibatch (gather + actor & critic forward + objective + full backward) for the
// shape class  S <= 64, net [h1, h2] with h1, h2 in {64, 128}, A <= 8  (BASELINE configs 2 / 4 / 5: [128,128] and the
// Pendulum demo's [128,64]).  gfx950 fp32 MFMA.  This header holds the templates; ppo_step_w4*.hip instantiate one
// (h1, h2) pair each (N1 = h1 / 32, N2 = h2 / 32 tile counts) so that the shapes compile in parallel.
//
// Replaces AgentPPO.update_objectives up to the optimizer steps (elegantrl/agents/AgentPPO.py:173-204) and
// ActorPPO.get_logprob_entropy (:378-386), like ppo_step.hip, and writes the same slabs.  What differs is the mapping:
//
//   * grid = (ceil(B / 128), 2 nets), 256 threads: FOUR waves, one per SIMD, each owning 32 samples.  A wave alone on
//     its SIMD has the whole 512-entry register file (256 arch + 256 acc VGPRs): H1, GELU'(z1), H2, GELU'(z2) of its 32
//     samples (256 registers per lane) stay in registers from the first layer to the last weight gradient -- nothing is
//     parked in memory (the 8-wave kernel round-trips GELU'(z1) through its slab: 33.6 MB per launch at B = 16384).
//   * every layer is computed transposed on v_mfma_f32_32x32x2_f32:
//         outT (32 features x 32 samples) += W (32 rows x 2 k) . inT (2 k x 32 samples).
//     The result tile leaves lane (m = lane & 31, hi = lane >> 5) holding features 8 g + 4 hi + j (acc[4 g + j]) of
//     sample m.  The next layer walks its reduction index in the order (tile, g, j) and pairs k = 8 g + j (lane half 0)
//     with k = 8 g + 4 + j (lane half 1): the B operand of step (g, j) is then exactly acc[4 g + j] of the previous
//     layer -- the register chain of ppo_step.hip in the 32x32 layout -- and the A operand of four consecutive steps is
//     one 16-byte LDS read W[row][32 T + 8 g + 4 hi .. + 3].  Half the LDS operand traffic of the 16x16x4 form.
//
// What bounds it (tools/mfma_issue_bench.hip, profiles/r02_mfma_issue_bench.txt): the fp32 MFMA runs on the vector
// ALUs -- a wave's VALU instructions do NOT overlap its fp32 MFMAs, each one adds its 4 cycles (transcendentals 8) to the
// 64 of an MFMA, while LDS reads, waits and s_nops between MFMAs are free and dependent MFMAs issue back to back.  So the
// kernel's time is (MFMA count x 64 + VALU count x 4) cycles plus whatever latency is exposed -- a burst of n VALU
// instructions between two MFMAs costs ~10 + 4 n cycles, a coalesced global store ~10, a global load ~17 -- and the design
// rules are: no padded MFMA work (the actor's 8-row output layer is 128 v_mfma_f32_4x4x1 on 4-sample x 4-action blocks,
// not a 32-row tile), as few and as few-but-long VALU bursts as possible (packed-fp32 GELU as one block per tile,
// normalisation folded into one packed FMA per two elements, biases loaded straight into the accumulators by ds_read,
// LDS reads off one base register with immediate offsets), operands prefetched one group ahead, and fences
// (sched_barrier) that keep the compiler from hoisting a whole unrolled layer's operand reads into registers.
//
// Weight gradients are the staged scheme of ppo_step.hip (T[feature][sample] tiles in LDS, 32x32x2 tiles, K = 128 samples;
// wave = row tile, so that the bias gradient falls out of the operand reads) and leave the CU by NON-TEMPORAL stores: the
// slab is written once and read once by the reduction, and keeping its 26 MB per launch out of L2 is worth 3-4 us.
#pragma once
#include "ppo_step.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int QNW = 4;           // waves per workgroup
constexpr int QNT = QNW * 64;

// LDS pool (floats): [RA: X sample-major, then W2 copy, later staged tiles][RB: W1 copy | X^T, later staged tiles]
//                    [RC: dY^T][RW3: W3 copy][s_bias: b1 | b2 | b3(16)][s_part: 4*16][s_red: 16]
constexpr int kQR = 128 * 68 + 64 * PLD;                  // >= 128 * PLD
static_assert(kQR >= 128 * PLD && kQR % 4 == 0, "staged tiles must fit the weight-copy regions");
constexpr int kQRC = 16 * PLD;
constexpr int kQRW3 = 16 * 132;
constexpr int kQBias = 128 + 128 + 16;
constexpr size_t kW4LdsBytes = (size_t)(2 * kQR + kQRC + kQRW3 + kQBias + QNW * 16 + 16) * sizeof(float);
static_assert(kW4LdsBytes <= 160 * 1024, "LDS budget");

// LLVM's IR-level code sinking moves pure arithmetic (an epilogue whose result is first used a few basic blocks later)
// across sched_barrier towards its first use; an empty volatile asm that "modifies" the value keeps it where it is written.
#define ERL_PIN4(a, b) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y))

// exact-erf GELU and its derivative for two elements at once on packed fp32 (v_pk_fma_f32: two FMAs per lane and issue
// slot).  erf through Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7) like gelu_and_grad_fast; the scalings are folded into
// the constants so that exp(-z^2 / 2) is ONE v_exp_f32 (2^x) of -(c |z|)^2, c = sqrt(log2(e) / 2), and the cdf comes out
// of half-scaled coefficients as 0.5 + copysign(0.5 erf(|z| / sqrt 2), z).  16 plain + 4 transcendental ops per pair.
__device__ __forceinline__ void gelu2(f32x2 z, f32x2 &y, f32x2 &gd)
{
    constexpr float kC = 0.84932180028801904272f;              // sqrt(log2(e) / 2)
    constexpr float kP = 0.3275911f * 0.70710678118654752440f / kC;   // A&S p, for the rescaled argument
    const f32x2 xa = {fabsf(z.x) * kC, fabsf(z.y) * kC};
    const f32x2 den = xa * kP + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f32x2 w = -(xa * xa);
    const f32x2 u = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};   // exp(-z^2 / 2)
    f32x2 p = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    p = t * p + (0.5f * 1.421413741f);
    p = t * p + (0.5f * -0.284496736f);
    p = t * p + (0.5f * 0.254829592f);
    p = p * t;
    const f32x2 h = 0.5f - p * u;                              // 0.5 erf(|z| / sqrt 2)
    const f32x2 hs = {copysignf(h.x, z.x), copysignf(h.y, z.y)};
    const f32x2 cdf = hs + 0.5f;
    y = z * cdf;
    gd = (z * u) * 0.39894228040143267794f + cdf;
}

// bias of output tile To in the D layout, loaded straight into an accumulator: element e <-> feature
// 32 To + 8 (e >> 2) + 4 hi + (e & 3)
__device__ __forceinline__ void load_bias16(const float *bias, int To, int hi, f32x16 &pb)
{
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * To + 8 * gq + 4 * hi);
        pb[4 * gq + 0] = b4.x; pb[4 * gq + 1] = b4.y; pb[4 * gq + 2] = b4.z; pb[4 * gq + 3] = b4.w;
    }
}

// The same for two pairs, written step by step for both: the dependent packed-FMA chains of the two pairs alternate in
// program order, which fills the wait state gfx950 needs between a packed op and its consumer (hipcc otherwise pads
// every dependent pair with an s_nop: 4 cycles each, ~10 per pair; it does not interleave two gelu2 calls by itself).
__device__ __forceinline__ void gelu4(f32x2 za, f32x2 zb, f32x2 &ya, f32x2 &ga, f32x2 &yb, f32x2 &gb)
{
    constexpr float kC = 0.84932180028801904272f;
    constexpr float kP = 0.3275911f * 0.70710678118654752440f / kC;
    const f32x2 xa = {fabsf(za.x) * kC, fabsf(za.y) * kC};
    const f32x2 xb = {fabsf(zb.x) * kC, fabsf(zb.y) * kC};
    const f32x2 da = xa * kP + 1.0f;
    const f32x2 db = xb * kP + 1.0f;
    const f32x2 wa = -(xa * xa);
    const f32x2 wb = -(xb * xb);
    const f32x2 ta = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
    const f32x2 tb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
    const f32x2 ua = {__builtin_amdgcn_exp2f(wa.x), __builtin_amdgcn_exp2f(wa.y)};
    const f32x2 ub = {__builtin_amdgcn_exp2f(wb.x), __builtin_amdgcn_exp2f(wb.y)};
    f32x2 pa = ta * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    f32x2 pb = tb * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    pa = ta * pa + (0.5f * 1.421413741f);
    pb = tb * pb + (0.5f * 1.421413741f);
    pa = ta * pa + (0.5f * -0.284496736f);
    pb = tb * pb + (0.5f * -0.284496736f);
    pa = ta * pa + (0.5f * 0.254829592f);
    pb = tb * pb + (0.5f * 0.254829592f);
    pa = pa * ta;
    pb = pb * tb;
    const f32x2 zua = za * ua;
    const f32x2 zub = zb * ub;
    const f32x2 ha = 0.5f - pa * ua;
    const f32x2 hb = 0.5f - pb * ub;
    const f32x2 sa = {copysignf(ha.x, za.x), copysignf(ha.y, za.y)};
    const f32x2 sb = {copysignf(hb.x, zb.x), copysignf(hb.y, zb.y)};
    const f32x2 ca = sa + 0.5f;
    const f32x2 cb = sb + 0.5f;
    ya = za * ca;
    yb = zb * cb;
    ga = zua * 0.39894228040143267794f + ca;
    gb = zub * 0.39894228040143267794f + cb;
}

__device__ __forceinline__ void gelu_tile(const f32x16 &acc, f32x16 &H, f32x16 &G)
{
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        f32x2 y0, g0, y1, g1;
        gelu4(f32x2{acc[e], acc[e + 1]}, f32x2{acc[e + 2], acc[e + 3]}, y0, g0, y1, g1);
        ERL_PIN4(y0, g0);
        ERL_PIN4(y1, g1);
        H[e] = y0.x; H[e + 1] = y0.y; H[e + 2] = y1.x; H[e + 3] = y1.y;
        G[e] = g0.x; G[e + 1] = g0.y; G[e + 2] = g1.x; G[e + 3] = g1.y;
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward layer on registers: out[To] (32 features x 32 samples) = GELU( bias + W[32 To .. + 31][:] . in ), To = 0..NO-1.
// W, bias: zero-padded LDS copies (row stride ldw = 4 * odd floats: the 16-byte reads of 16 consecutive rows hit 16
// distinct 16-byte bank groups).  KT = input tiles of 32.  One group = one 16-byte A read (issued a group ahead) + four
// MFMAs; the accumulator starts from the bias (read a tile ahead); the GELU epilogue of tile To sits after the first
// group of tile To + 1, when its accumulator has long been written back.
// ---------------------------------------------------------------------------------------------------------
template <int KT, int NO, int NG = 4 * KT>      // NG: reduction groups of 8 (default: the whole KT tiles; 1 for S <= 8)
__device__ __forceinline__ void fwd32(const float *W, int ldw, const float *bias, const f32x16 (&in)[KT], f32x16 (&outH)[NO],
                                      f32x16 (&outG)[NO], int m, int hi)
{
    constexpr int NC = NO * NG;
    const float *wbase = W + m * ldw + 4 * hi;
    float4 wq[2];
    auto issue = [&](int c, float4 &dst) {
        const int To = c / NG, gi = c % NG;
        dst = *reinterpret_cast<const float4 *>(wbase + 32 * To * ldw + 8 * gi);
    };
    issue(0, wq[0]);
    f32x16 acc, nb, prev;
    load_bias16(bias, 0, hi, nb);
    auto group = [&](int To, int gi) {
        const int c = To * NG + gi, Ti = gi >> 2, gq = gi & 3;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        const float4 a = wq[c & 1];
        acc = mfma32(a.x, in[Ti][4 * gq + 0], acc);
        acc = mfma32(a.y, in[Ti][4 * gq + 1], acc);
        acc = mfma32(a.z, in[Ti][4 * gq + 2], acc);
        acc = mfma32(a.w, in[Ti][4 * gq + 3], acc);
        __builtin_amdgcn_sched_barrier(0);
    };
    // (nested loops: one flat loop with the epilogue inside exceeds LLVM's pragma-unroll size limit and is left rolled,
    // which demotes every register array to scratch)
#pragma unroll
    for (int To = 0; To < NO; ++To) {
        acc = nb;
        group(To, 0);
        if (To + 1 < NO) load_bias16(bias, To + 1, hi, nb);
        if (To > 0) {
            gelu_tile(prev, outH[To - 1], outG[To - 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int gi = 1; gi < NG; ++gi) group(To, gi);
        prev = acc;
    }
    gelu_tile(prev, outH[NO - 1], outG[NO - 1]);
}

// ---------------------------------------------------------------------------------------------------------
// backward through a layer's input on registers:  gate[To] <- gate[To] * ( W^T . dz ),  W = LDS copy [32 KT rows][ldw]
// (A operand = W^T: lane (i, hi) supplies W[8 gi + 4 hi + j][32 To + i] for reduction group gi, four ds_read_b32 per
// group, issued one group ahead; the gate multiplies of tile To sit after the first group of tile To + 1).
// ---------------------------------------------------------------------------------------------------------
template <int KT, int NO>
__device__ __forceinline__ void bwd32(const float *W, int ldw, const f32x16 (&dz)[KT], f32x16 (&gate)[NO], int m, int hi)
{
    constexpr int NG = 4 * KT, NC = NO * NG;
    const float *wbase = W + (4 * hi) * ldw + m;
    float wq[2][4];
    auto issue = [&](int c, float(&dst)[4]) {
        const int To = c / NG, gi = c % NG;
        const float *p = wbase + (8 * gi) * ldw + 32 * To;
        // volatile: keeps them four ds_read_b32 with 16-bit immediate offsets off ONE base register (every offset of the
        // layer, <= 15 * 8 * 528 + 3 * 528 + 384 bytes, fits); merged into ds_read2_b32 (8-bit offsets) each pair needs its
        // own v_add, and a lone VALU instruction between two MFMAs costs ~14 cycles (tools/mfma_issue_bench.hip, mode 51)
        typedef const volatile __attribute__((address_space(3))) float *lds_vptr;
        lds_vptr q = (lds_vptr)(uint32_t)(uintptr_t)p;           // low half of a generic LDS pointer = the LDS byte address
        dst[0] = q[0]; dst[1] = q[ldw]; dst[2] = q[2 * ldw]; dst[3] = q[3 * ldw];
    };
    issue(0, wq[0]);
    f32x16 acc = {0}, prev = {0};
    auto group = [&](int To, int gi) {
        const int c = To * NG + gi, Tk = gi >> 2, gq = gi & 3;
        if (c + 1 < NC) issue(c + 1, wq[(c + 1) & 1]);
        acc = mfma32(wq[c & 1][0], dz[Tk][4 * gq + 0], acc);
        acc = mfma32(wq[c & 1][1], dz[Tk][4 * gq + 1], acc);
        acc = mfma32(wq[c & 1][2], dz[Tk][4 * gq + 2], acc);
        acc = mfma32(wq[c & 1][3], dz[Tk][4 * gq + 3], acc);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int To = 0; To < NO; ++To) {
        acc = f32x16{0};
        group(To, 0);
        if (To > 0) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                f32x2 v = f32x2{gate[To - 1][e], gate[To - 1][e + 1]} * f32x2{prev[e], prev[e + 1]};
                asm volatile("" : "+v"(v.x), "+v"(v.y));
                gate[To - 1][e] = v.x;
                gate[To - 1][e + 1] = v.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int gi = 1; gi < NG; ++gi) group(To, gi);
        prev = acc;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) gate[NO - 1][e] *= prev[e];
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

// dW (NR*32 x NB*32) = TA . TB^T over the 128 staged samples, output tiles split over the four waves.  The sum over
// samples is order-free: lane half `hi` takes samples 8 j + 4 hi + {0..3} of every group of 8 (one 16-byte read per
// operand feeds four MFMAs); operands are read one group ahead.
template <int NR, int NB>
__device__ __forceinline__ void weight_grad_w4(const float *TA, const float *TB, float *__restrict__ dW, int ldw, int cols_real,
                                               float *__restrict__ db, int wave, int lane)
{
    // wave w owns row tile it = w % NR (features 32 it .. 32 it + 31 of dZ^T); the CS = 4 / NR waves of a row tile split its
    // NB column tiles (jt = jc, jc + CS, ...; [128,128]: wave = row tile, every column tile).  A wave's A operand is the same
    // for every tile, so the bias gradient -- the row sums of dZ^T -- falls out of the operand reads of the first tile (two
    // packed adds per four MFMAs) instead of a second pass over the staged tile.  The (tile, group) loop is flat: operands are
    // read one group ahead across tile seams, and a finished tile is stored after the first group of the next one (its last
    // MFMA has drained by then), into the other accumulator.
    static_assert(NR == 1 || NR == 2 || NR == 4, "row tiles per layer: 1, 2 or 4");
    constexpr int CS = 4 / NR, NBW = (NB + CS - 1) / CS;      // waves per row tile, column tiles per wave
    static_assert(NB % CS == 0 || NBW == 1, "a wave has all of its column tiles or none");
    const int l31 = lane & 31, hi = lane >> 5;
    const int it = wave % NR, jc = wave / NR;
    if (NB % CS != 0 && jc >= NB) return;                     // wave-uniform: more waves than tiles
    const float *a4 = TA + (32 * it + l31) * PLD + 4 * hi;
    const float *b4 = TB + (32 * jc + l31) * PLD + 4 * hi;
    constexpr int NG = PB / 8;
    f32x16 acc[2];
    f32x2 bs = {0.f, 0.f};
    float4 av[2], bv[2];
    av[0] = *reinterpret_cast<const float4 *>(a4);
    bv[0] = *reinterpret_cast<const float4 *>(b4);
    // The slab is written once and read once, by another kernel: NON-TEMPORAL stores keep its 104 KB per workgroup (26 MB per
    // launch) from being allocated in L2 -- with plain stores the kernel is 3-4 us slower inside the PPO loop (after the slab
    // reduction has left its lines spread over the XCDs' L2s: 54.8 -> 51.0 us) and 2 us slower back to back (49.6 -> 47.7 us);
    // on the pool's slow boxes the difference is 83 -> 57 us.
    auto store = [&](int k, const f32x16 &c) {
        const int i = 32 * (jc + CS * k) + l31;
        if (i < cols_real) {
            float *o = dW + (size_t)(32 * it + 4 * hi) * ldw + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_store(c[r], o + (size_t)((r & 3) + 8 * (r >> 2)) * ldw);
        }
    };
#pragma unroll
    for (int jt = 0; jt < NBW; ++jt) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int gi = jt * NG + j;
            if (gi + 1 < NBW * NG) {
                const int jn = (gi + 1) % NG, tn = (gi + 1) / NG;
                av[(gi + 1) & 1] = *reinterpret_cast<const float4 *>(a4 + 8 * jn);
                bv[(gi + 1) & 1] = *reinterpret_cast<const float4 *>(b4 + 32 * CS * tn * PLD + 8 * jn);
            }
            const float4 x = av[gi & 1], y = bv[gi & 1];
            f32x16 &c = acc[jt & 1];
            if (j == 0) {
                const f32x16 zero = {0};
                c = mfma32(x.x, y.x, zero);
            } else {
                c = mfma32(x.x, y.x, c);
            }
            c = mfma32(x.y, y.y, c);
            c = mfma32(x.z, y.z, c);
            c = mfma32(x.w, y.w, c);
            if (jt == 0) {
                bs += f32x2{x.x, x.y};
                bs += f32x2{x.z, x.w};
            }
            if (j == 0 && jt > 0) store(jt - 1, acc[(jt - 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    store(NBW - 1, acc[(NBW - 1) & 1]);
    float s = bs.x + bs.y;
    s += __shfl_xor(s, 32, 64);
    if (hi == 0 && jc == 0) db[32 * it + l31] = s;
}

// dW1 and db1 for S <= 8 inputs (the Pendulum demo's S = 3) on the vector ALUs: a 32-column MFMA tile would be at least 3/4
// padding (64 MFMAs = 4.1k cycles per wave for 24 real columns' worth at S = 3).  Lane (l31 = feature of the wave's row tile, hi =
// sample half) reduces its 64 samples of dZ1^T[f][:] against X^T[k][:] (broadcast 16-byte reads), the halves meet by one shuffle.
template <int NR, int KM>      // KM = 4 (S <= 4) or 8: columns reduced unconditionally -- rows k >= S of X^T are zero; no branch in the loop
__device__ __forceinline__ void weight_grad_tiny_k(const float *TA, const float *TX, float *__restrict__ dW, int S, float *__restrict__ db,
                                                   int wave, int lane)
{
    const int l31 = lane & 31, hi = lane >> 5;
    const int it = wave % NR;
    const float *a = TA + (32 * it + l31) * PLD + 64 * hi;
    const float *x = TX + 64 * hi;
    f32x2 acc[KM][2];
#pragma unroll
    for (int k = 0; k < KM; ++k) acc[k][0] = acc[k][1] = f32x2{0.f, 0.f};
    f32x2 bs = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 dz = *reinterpret_cast<const float4 *>(a + 4 * j);
        bs += f32x2{dz.x, dz.y};
        bs += f32x2{dz.z, dz.w};
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + k * PLD + 4 * j);      // same address in every lane of a half: broadcast
            acc[k][0] += f32x2{dz.x, dz.y} * f32x2{xv.x, xv.y};
            acc[k][1] += f32x2{dz.z, dz.w} * f32x2{xv.z, xv.w};
        }
    }
    float s = bs.x + bs.y;
    s += __shfl_xor(s, 32, 64);
    float r[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        r[k] = (acc[k][0].x + acc[k][0].y) + (acc[k][1].x + acc[k][1].y);
        r[k] += __shfl_xor(r[k], 32, 64);
    }
    if (hi == 0) {
        const int f = 32 * it + l31;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < S) slab_store(r[k], dW + (size_t)f * S + k);
        db[f] = s;
    }
}

template <int NR>
__device__ __forceinline__ void weight_grad_tiny(const float *TA, const float *TX, float *__restrict__ dW, int S, float *__restrict__ db,
                                                 int wave, int lane)
{
    if (wave / NR > 0) return;                                // (wave-uniform) NR < 4: one wave per row tile is enough here
    if (S <= 4) weight_grad_tiny_k<NR, 4>(TA, TX, dW, S, db, wave, lane);
    else weight_grad_tiny_k<NR, 8>(TA, TX, dW, S, db, wave, lane);
}

// copy_load for a matrix that fills its padded tile exactly (rows x COLS, COLS % 4 == 0, 16-byte aligned): no clamps, no
// selects -- one address and immediate offsets (the generic copy_load spends ~14 VALU instructions per 16-byte load)
template <int MAXV, int COLS>
__device__ __forceinline__ void copy_load_full(float4 (&v)[MAXV], const float *__restrict__ src, int tid)
{
    const float4 *p = reinterpret_cast<const float4 *>(src) + tid;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) v[u] = p[u * QNT];
}

// LDS-DMA copy (global_load_lds_dwordx4: memory -> LDS without passing through registers) of a row-major [rows][128] fp32
// matrix into its padded LDS image [NROWS_PAD][132].  The image is a sequence of 16-byte units, 33 per row (32 data + 1
// pad); one wave instruction fills 64 consecutive units (LDS address = uniform base + 16 lane), every lane fetching the
// unit's own source address -- pad units fetch a neighbour, rows >= `rows` are left alone (the caller has zeroed them).
// Completion is tracked by the issuing wave's vmcnt.
template <int NROWS_PAD>
__device__ __forceinline__ void dma_copy128(const float *__restrict__ src, int rows, float *dst, int wave, int lane)
{
    constexpr int UNITS = NROWS_PAD * 33, NK = (UNITS + 63) / 64;
#pragma unroll
    for (int i = 0; i < (NK + QNW - 1) / QNW; ++i) {
        const int k = wave + QNW * i;                       // wave-uniform
        if (k < NK) {
            const int u = 64 * k + lane;
            const int row = (u * 1986) >> 16, cu = u - 33 * row;     // u / 33 for u < 4224
            if (u < UNITS && row < rows) __builtin_amdgcn_global_load_lds(src + row * 128 + 4 * min(cu, 31), dst + 256 * k, 16, 0, 0);
        }
    }
}

template <bool ACTOR, int KXP, int N1, int N2, bool VEC>     // KXP: input tiles of 32 (1: S <= 32, 2: S <= 64); 0: S <= 8
__device__ __forceinline__ void ppo_block_w4(const Ppo2Args &g, float *smem)
{
    constexpr bool TINY = KXP == 0;
    constexpr int KX = TINY ? 1 : KXP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    constexpr int net = ACTOR ? 0 : 1;
    constexpr int h1 = 32 * N1, h2 = 32 * N2;
    // W2 / W3 come in by LDS-DMA when their rows are 128 floats long and 16-byte aligned, through registers otherwise
    constexpr bool DMA2 = VEC && N1 == 4, DMA3 = VEC && N2 == 4;
    const int S = g.S, OUT = ACTOR ? g.A : 1;
    const Dims d{S, h1, h2, OUT};
    const float *P = g.P[net];
    const float *std_log = P + d.oStd();

    float *RA = smem;                      // X sample-major [128][XLD], then W2 copy [h2][ld2], later staged tiles [<= 128][PLD]
    float *RB = RA + kQR;                  // W1 copy [h1][ld1] | X^T [32 KX][PLD], later staged tiles
    float *RC = RB + kQR;                  // [16][PLD]   dY^T
    float *RW3 = RC + kQRC;                // W3 copy [16][ld3] (rows >= OUT are zero)
    float *s_b1 = RW3 + kQRW3, *s_b2 = s_b1 + 128, *s_b3 = s_b2 + 128;
    float *s_part = s_b3 + 16;             // [4 waves][16]  per-wave dstd_log partials
    float *s_red = s_part + QNW * 16;      // [16] block_sum scratch
    constexpr int ld1 = lds_ld(32 * KX), ld2 = lds_ld(h1), ld3 = lds_ld(128);
    constexpr int XLD = 32 * KX + 4;       // sample-major X rows: 16-byte aligned, 16 consecutive rows on 16 distinct bank groups
    float *RX = RB + 128 * lds_ld(64);

    PROF(0);
    // ---- prologue.  Trip 1: the sample id, W1 and the biases.
    const int col = 32 * wave + m;                         // sample slot inside the workgroup
    const int64_t bidx = (int64_t)blockIdx.x * PB + col;
    const bool valid = bidx < g.B;
    const int64_t id = g.ids[valid ? bidx : 0];
    const AdvNorm advn = adv_norm_consts(ACTOR ? g.adv_stats : nullptr);   // (under the id's round trip; scalar registers)
    float4 c1[N1 * KX];
    if (VEC && S == 32 * KX) copy_load_full<N1 * KX, 32 * KX>(c1, P + d.oW1(), tid);     // uniform branch
    else copy_load<VEC, N1 * KX, QNT>(c1, P + d.oW1(), h1, S, h1, 32 * KX, tid);
    const float bias_pre = (tid < 128) ? (tid < h1 ? P[d.ob1() + tid] : 0.f) : (tid - 128 < h2 ? P[d.ob2() + tid - 128] : 0.f);
    float b3_pre = 0.f;
    if (tid < 16) b3_pre = (tid < OUT) ? P[d.ob3() + tid] : 0.f;
    // this lane's normalisation constants: it gathers the 16-byte chunk xc of EVERY row it loads (see below)
    const int xr = lane >> 4, xc = lane & 15;
    const float *avg = g.avg[net], *sdv = g.sd[net];
    const float4 a4 = load4<VEC>(avg, 4 * xc, S), s4 = load4<VEC>(sdv, 4 * xc, S);

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
    const int64_t row = valid ? t_ * g.N + n_ : 0;          // padding slots read row 0 (finite data) and carry zero weight
    // The state rows are gathered COALESCED: lane (xr = lane >> 4, xc = lane & 15) loads chunk xc (16 bytes) of the rows of
    // samples 4 i + xr, i = 0..7 -- 16 lanes cover one 256-byte row -- whose row numbers sit in lanes 4 i + xr (ds_bpermute).
    float4 XR[8];
    {
        const int rlo = (int)(uint32_t)row, rhi = (int)(row >> 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int src = 4 * (4 * i + xr);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, rlo), hi32 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, rhi);
            const int64_t r_i = (int64_t)(((uint64_t)hi32 << 32) | lo);
            XR[i] = load4<VEC>(g.states + r_i * S, 4 * xc, S);
        }
    }
    // per-sample scalars (consumed after the output layer)
    const float um = (valid && g.unmasks[row]) ? 1.f : 0.f;
    const float xa = ACTOR ? g.logprobs[row] : g.reward_sums[row];
    const float xb = ACTOR ? g.advantages[row] : 0.f;
    float act_pre[4] = {0.f, 0.f, 0.f, 0.f}, sl_pre[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's actions a = 4 hi + j (actor)
    if (ACTOR) {
        // the lane's four actions: ONE 16-byte load when the rows allow it (a scattered dword load touches as many lines)
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
    // ---- publish the W1 copy and the biases (zero padded to the tile grid), visible after barrier (0a)
    copy_store<N1 * KX, QNT>(c1, RB, ld1, h1, 32 * KX, tid);
    s_b1[tid] = bias_pre;                                   // s_b1 | s_b2 contiguous
    if (tid < 16) s_b3[tid] = b3_pre;
#pragma unroll
    for (int e = tid; e < kQRW3 / 4; e += QNT) reinterpret_cast<float4 *>(RW3)[e] = zero4();   // rows >= OUT of the W3 copy stay zero
    // ---- normalise, (x - avg) / (std + 1e-4) (AgentPPO.py:360-361) as x * r + (-avg r), r = 1 / (std + 1e-4): one packed
    // FMA per two elements; write the rows sample-major (for this wave's own B operands) and feature-major (X^T for dW1)
    {
        const f32x2 r01 = {__builtin_amdgcn_rcpf(s4.x + 1e-4f), __builtin_amdgcn_rcpf(s4.y + 1e-4f)};
        const f32x2 r23 = {__builtin_amdgcn_rcpf(s4.z + 1e-4f), __builtin_amdgcn_rcpf(s4.w + 1e-4f)};
        const f32x2 n01 = -(f32x2{a4.x, a4.y} * r01), n23 = -(f32x2{a4.z, a4.w} * r23);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 x01 = f32x2{XR[i].x, XR[i].y} * r01 + n01, x23 = f32x2{XR[i].z, XR[i].w} * r23 + n23;
            const int sc = 32 * wave + 4 * i + xr;
            if (4 * xc < 32 * KX) {
                *reinterpret_cast<float4 *>(RA + sc * XLD + 4 * xc) = make_float4(x01.x, x01.y, x23.x, x23.y);
                float *t = RX + (4 * xc) * PLD + sc;
                t[0] = x01.x; t[PLD] = x01.y; t[2 * PLD] = x23.x; t[3 * PLD] = x23.y;
            }
        }
    }
    // this wave's own rows back as B operands (same wave: LDS executes a wave's accesses in order, no barrier needed)
    f32x16 X[KX];
    {
        const float *xs = RA + col * XLD + 4 * hi;
#pragma unroll
        for (int t = 0; t < 4 * KX; ++t) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + 8 * t);
            X[t >> 2][4 * (t & 3) + 0] = v.x; X[t >> 2][4 * (t & 3) + 1] = v.y;
            X[t >> 2][4 * (t & 3) + 2] = v.z; X[t >> 2][4 * (t & 3) + 3] = v.w;
        }
    }
    PROF_NV(1);
    lds_barrier();                                                   // (0a) W1 copy, biases visible; X rows consumed
    PROF_NV(2);
    // ---- W2, W3 are not needed before the second layer: requested only now, so that the prologue's burst (every CU pulls
    // its 32 KB of W1 and 32 KB of gathered rows at once, ~11 B/clk per CU) is not stretched by another 68 KB; they
    // stream in under the first layer's MFMAs, by LDS-DMA: held in registers they would need 72 VGPRs across the first layer
    // (hipcc spilled them to scratch, waiting for every load first)
    float4 c2[DMA2 ? 1 : N1 * N2], c3[DMA3 ? 1 : 2];
    if constexpr (DMA2) dma_copy128<h2>(P + d.oW2(), h2, RA, wave, lane);
    else copy_load<VEC, N1 * N2, QNT>(c2, P + d.oW2(), h2, h1, h2, h1, tid);
    if constexpr (DMA3) dma_copy128<16>(P + d.oW3(), OUT, RW3, wave, lane);
    else copy_load<VEC, 2, QNT>(c3, P + d.oW3(), OUT, h2, 16, h2, tid);
    f32x16 H1[N1], G1[N1], H2[N2], G2[N2];
    fwd32<KX, N1, TINY ? 1 : 4 * KX>(RB, ld1, s_b1, X, H1, G1, m, hi);       // S <= 8: one reduction group (k = 0..7)
    PROF_NV(3);
    if constexpr (!DMA2) copy_store<N1 * N2, QNT>(c2, RA, ld2, h2, h1, tid);
    if constexpr (!DMA3) copy_store<2, QNT>(c3, RW3, ld3, 16, h2, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's share of the W2 / W3 copies has landed
    lds_barrier();                                                   // (0b) W2, W3 copies visible
    fwd32<N1, N2>(RA, ld2, s_b2, H1, H2, G2, m, hi);
    PROF(4);
    // ---- output layer on the vector ALUs: A <= 8 rows would fill a quarter of a 32-row MFMA tile, and the fp32 MFMA runs
    // at the packed-FMA rate anyway.  Each lane reduces its own 64 features of H2 against the rows of W3 (broadcast
    // 16-byte LDS reads), the two lane halves meet through v_permlane32_swap, which also leaves outputs a = 4 hi + j of
    // sample m in lane (m, hi) -- the layout the objective and the dZ2 MFMAs want.
    float Y[4] = {0.f, 0.f, 0.f, 0.f};
    if (ACTOR) {
        // The 8 action rows on v_mfma_f32_4x4x1 (16 independent 4 x 4 blocks, K = 1, 8 cycles): block = 4 neighbouring lanes
        // = 4 samples (B operand: the lane's own H2 value of one feature), rows = 4 actions (A operand: lane (block, i) supplies
        // W3[i (+4)][feature]; the lane halves carry different features, which per-block operands allow).  Two instructions
        // per feature cover the 8 actions: 128 MFMAs = ~1k cycles per wave and 32 16-byte LDS reads per lane.  The packed-FMA
        // form this replaces issued the same ~1k cycles of arithmetic but read every W3 row as a broadcast operand -- 128
        // 16-byte reads per lane, 512 KB of LDS return traffic per workgroup = 4k cycles at 128 B/clk: the layer was
        // LDS-bandwidth bound (3.3k cycles measured), and the actor workgroups are the kernel's critical path.
        f32x4 ya[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) ya[q >> 1][q & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *w3a = RW3 + (lane & 3) * ld3 + 4 * hi;
        constexpr int DEPTH = 2;
        float4 wq[DEPTH + 1][2];
        auto issue = [&](int c, float4(&dst)[2]) {
            const int T = c >> 2, gq = c & 3;
            dst[0] = *reinterpret_cast<const float4 *>(w3a + 32 * T + 8 * gq);
            dst[1] = *reinterpret_cast<const float4 *>(w3a + 4 * ld3 + 32 * T + 8 * gq);
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
            const float lo = ya[0][0][j] + ya[0][1][j], hi_ = ya[1][0][j] + ya[1][1][j];     // this half's features: actions j and 4 + j
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi_), false, false);
            Y[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]) + bb[j];   // lanes < 32: output j; lanes >= 32: output 4 + j
        }
    } else {
        // value head: one row on the vector ALUs -- each lane reduces its own 64 features of H2 against W3's row (broadcast
        // 16-byte LDS reads, three batches ahead), the lane halves meet through a cross-half shuffle
        f32x2 yp = {0.f, 0.f}, yq = {0.f, 0.f};
        const float *w3 = RW3 + 4 * hi;
        float4 wv[4 * N2];
#pragma unroll
        for (int c = 0; c < 4 * N2; ++c) wv[c] = *reinterpret_cast<const float4 *>(w3 + 32 * (c >> 2) + 8 * (c & 3));
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
        const float diff = Y[0] - xa;                     // only (hi = 0, j = 0) is the value head
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
        const PpoActorTerms o = ppo_actor_terms(g.objective, adv_normalized(xb, advn),   /* raw advantages are normalised here (AgentPPO.py:149) */
                                                    lp, xa, g.ratio_clip, g.lambda_entropy, um, OUT, true);
        if (hi == 0) {
            loss0 = valid ? o.logged : 0.f;                     // padding rows contribute 0
            loss1 = valid ? o.ent_mask : 0.f;
        }
        const float dlp = (valid ? o.dlp : 0.f) * g.inv_batch;  // d loss / dlogp_new
        const float ent_term = (valid ? o.ent_w : 0.f) * g.inv_batch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = 4 * hi + j < OUT;
            dY[j] = on ? dlp * (diffv[j] * ivar[j]) : 0.f;                                     // dL/dmean
            dsl[j] = on ? dlp * (diffv[j] * diffv[j] * ivar[j] - 1.f) + ent_term : 0.f;        // dL/dstd_log, this sample
        }
    }

    // ---- dZ2 = (W3^T dY) * GELU'(z2)  (K = 8 outputs: four k-pairs);  dZ1 = (W2^T dZ2) * GELU'(z1)
    PROF(6);
    {
        float w3[N2][4];
#pragma unroll
        for (int To = 0; To < N2; ++To) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w3[To][j] = RW3[(4 * hi + j) * ld3 + 32 * To + m];
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
    bwd32<N2, N1>(RA, ld2, G2, G1, m, hi);                          // G1 (the gate) <- dZ1
    PROF(7);
    lds_barrier();                                                   // (1) every wave is done with the weight copies
    PROF(8);

    float *slab = g.slabs + (size_t)blockIdx.x * g.stride + (ACTOR ? 0 : g.Pa);
    // ---- layer 1: dW1 = dZ1^T . X, db1;  (dY^T is staged alongside for the output layer)
    stage32<N1>(RA, G1, col, hi);                                   // dZ1^T
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        RC[(4 * hi + j) * PLD + col] = dY[j];
        RC[(8 + 4 * hi + j) * PLD + col] = dsl[j];       // rows 8..15: per-sample dL/dstd_log (zero for the critic); their row sums
                                                         // are the std_log gradient, and dW3 never stores product rows >= A
    }
    lds_barrier();                                                   // (2)
    PROF(9);
    if constexpr (TINY) weight_grad_tiny<N1>(RA, RX, slab + d.oW1(), S, slab + d.ob1(), wave, lane);
    else weight_grad_w4<N1, KX>(RA, RX, slab + d.oW1(), S, S, slab + d.ob1(), wave, lane);   // dW1 and db1
    PROF(10);
    lds_barrier();                                                   // (3) dZ1^T, X^T consumed

    // ---- output layer: dW3 (16 x h2) = dY^T . H2 on 16x16x4 MFMA, 16-column tiles split over the waves
    stage32<N2>(RA, H2, col, hi);                                   // H2^T
    stage32<N1>(RB, H1, col, hi);                                   // H1^T (for dW2)
    lds_barrier();                                                   // (4)
    PROF(11);
    {
        // row sums of the A operand (rows a < OUT: db3; rows 8 + a: the std_log gradient) ride on wave 0's first tile
        const int l15 = lane & 15, q = lane >> 4;
        f32x2 hs = {0.f, 0.f};
#pragma unroll
        for (int rep = 0; rep < (2 * N2 + QNW - 1) / QNW; ++rep) {
            const int it = wave + QNW * rep;                            // 16-column tile of dW3 (wave-uniform)
            if (it >= 2 * N2) break;
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
    }
    lds_barrier();                                                   // (5) H2^T consumed
    stage32<N2>(RA, G2, col, hi);                                   // dZ2^T
    lds_barrier();                                                   // (6)
    PROF(12);

    // ---- layer 2: dW2 = dZ2^T . H1, db2
    weight_grad_w4<N2, N1>(RA, RB, slab + d.oW2(), h1, h1, slab + d.ob2(), wave, lane); // dW2 and db2
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

template <int KX, int N1, int N2, bool VEC>
__global__ __launch_bounds__(QNT) void ppo_step_w4_kernel(Ppo2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SpanT t_span = span_enter(g);
    if (blockIdx.y == 0) ppo_block_w4<true, KX, N1, N2, VEC>(g, smem);
    else ppo_block_w4<false, KX, N1, N2, VEC>(g, smem);
    span_exit(g, t_span);
}

template <int KX, int N1, int N2, bool VEC>
int launch_w4(const Ppo2Args &g, int n_slabs, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)ppo_step_w4_kernel<KX, N1, N2, VEC>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kW4LdsBytes),
                                "hipFuncSetAttribute(ppo_step_w4_kernel)");
        if (rc) return rc;
        attr_set = true;
    }
    hipLaunchKernelGGL((ppo_step_w4_kernel<KX, N1, N2, VEC>), dim3(n_slabs, 2), dim3(QNT), kW4LdsBytes, stream, g);
    return erl_hip_status(hipGetLastError(), "erl_ppo_step_f32");
}

// the five instantiations of one (h1, h2) pair: S <= 8 | S <= 32 / S <= 64, 16-byte-aligned inputs with S % 4 == 0 (vec) or not
template <int N1, int N2>
int launch_w4_shape(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream)
{
    if (g.S <= 8) return launch_w4<0, N1, N2, false>(g, n_slabs, stream);        // (element-wise loaders: at most 8 floats per row)
    if (vec) return g.S > 32 ? launch_w4<2, N1, N2, true>(g, n_slabs, stream) : launch_w4<1, N1, N2, true>(g, n_slabs, stream);
    return g.S > 32 ? launch_w4<2, N1, N2, false>(g, n_slabs, stream) : launch_w4<1, N1, N2, false>(g, n_slabs, stream);
}

}  // namespace
