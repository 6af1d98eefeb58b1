// K1 rollout step and K2 value pre-pass on the register-chained MLP (mlp_chain.h), slab reduction of K6, MFMA self-test.
// gfx950 / fp32 MFMA; the latency form of K1 runs its hidden layers on the bf16 matrix pipe (rollout_bf16.h).
//
// K1  one vectorised rollout step = ActorPPO.get_action + the three buffer stores + convert_action_for_env
//     (elegantrl/agents/AgentPPO.py:113-119, :368-376, :388-390).  A wave owns 16 envs; 4 waves per workgroup share
//     the LDS copies of W1 / W2 / W3.  Lane (m, q) ends up with the policy means of actions a = 4 q + r, samples
//     them (injected eps or Philox4x32-10 + Box-Muller), stores the pre-tanh action, tanh(action) for the env and
//     the log-prob (cross-lane sum over q).
// K2  value pre-pass: persistent workgroups of 8 waves; a wave walks 16-row tiles with the next tile's state rows
//     prefetched under the current tile's MFMAs  (AgentPPO.py:141-143, :219-220, :435-441).
#include "mlp_chain.h"
#include "rollout_bf16.h"
#include "ppo_step_wd.h"

namespace {

constexpr int64_t kSplitMaxEnvs = 16384;

// LDS pool of the forward kernels (floats): [W2 copy 128 x 132][W1 copy 128 x 132][W3 copy 16 x 132][b1 | b2 | b3]
constexpr int kFwdW = 128 * 132;
constexpr int kFwdW3 = 16 * 132;
constexpr size_t kFwdLdsBytes = (size_t)(2 * kFwdW + kFwdW3 + 128 + 128 + 16) * sizeof(float);

struct FwdArgs {
    const float *P, *avg, *sd;
    int S, h1, h2, out;
    const float *states;      // (rows, S)
    int64_t rows;
    // K2
    float *values;
    // K1
    const float *noise;
    uint64_t seed, counter;
    float *o_state, *o_action, *o_logprob, *o_env;
    long long *prof;          // ERL_PROFILE builds only: [wave][16] s_memtime stamps of workgroup 0 (rollout_split_kernel)
};

#ifdef ERL_PROFILE
long long *g_fwd_prof = nullptr;
#define RPROF(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (g.prof && blockIdx.x == 0 && lane == 0) g.prof[wave * 16 + (i)] = (long long)t_;      \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define RPROF(i) do { } while (0)
#endif

// copies W1 / W2 / W3 / biases of one network into LDS (zero padded to the tile grid) and barriers
template <bool VEC, int NW>
__device__ __forceinline__ void load_network(const FwdArgs &g, const Dims &d, int ns, float *W2c, float *W1c, float *W3c, float *sb,
                                             int tid)
{
    constexpr int NT = NW * 64;
    constexpr int MV = 8 * 512 / NT;     // float4 per thread for a 128 x 128 matrix
    const int ld1 = lds_ld(16 * ns), ld2 = lds_ld(d.h1), ld3 = lds_ld(d.h2);
    float4 c2[MV], c1[MV], c3[(512 + NT - 1) / NT];
    copy_load<VEC, MV, NT>(c2, g.P + d.oW2(), d.h2, d.h1, d.h2, d.h1, tid);
    copy_load<VEC, MV, NT>(c1, g.P + d.oW1(), d.h1, d.S, d.h1, 16 * ns, tid);
    copy_load<VEC, (512 + NT - 1) / NT, NT>(c3, g.P + d.oW3(), d.out, d.h2, 16, d.h2, tid);
    for (int i = tid; i < 272; i += NT) {
        float b = 0.f;
        if (i < 128) b = (i < d.h1) ? g.P[d.ob1() + i] : 0.f;
        else if (i < 256) b = (i - 128 < d.h2) ? g.P[d.ob2() + i - 128] : 0.f;
        else b = (i - 256 < d.out) ? g.P[d.ob3() + i - 256] : 0.f;
        sb[i] = b;
    }
    copy_store<MV, NT>(c2, W2c, ld2, d.h2, d.h1, tid);
    copy_store<MV, NT>(c1, W1c, ld1, d.h1, 16 * ns, tid);
    copy_store<(512 + NT - 1) / NT, NT>(c3, W3c, ld3, 16, d.h2, tid);
}

template <bool VEC>
__device__ __forceinline__ void load_rows_raw(float4 (&R)[8], const float *__restrict__ srow, int ns, int S, int q)
{
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < ns) R[t] = load4<VEC>(srow, 16 * t + 4 * q, S);
}

template <bool VEC>
__device__ __forceinline__ void normalise_rows(const float4 (&R)[8], f32x4 (&X)[8], const float *__restrict__ avg,
                                               const float *__restrict__ sd, int ns, int S, int q, bool valid)
{
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < ns) {
            const int k0 = 16 * t + 4 * q;
            const float4 a4 = load4<VEC>(avg, k0, S), s4 = load4<VEC>(sd, k0, S);
            const float rr[4] = {R[t].x, R[t].y, R[t].z, R[t].w}, aa[4] = {a4.x, a4.y, a4.z, a4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xn = (rr[r] - aa[r]) / (ss[r] + 1e-4f);   // (s - avg) / (std + 1e-4), AgentPPO.py:360-361
                X[t][r] = (valid && k0 + r < S) ? xn : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2
// ---------------------------------------------------------------------------------------------------------
template <int NS_, int N1_, int N2_, bool VEC>
__global__ __launch_bounds__(512) void value_forward2_kernel(FwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const Dims d{g.S, N1_ ? 16 * N1_ : g.h1, N2_ ? 16 * N2_ : g.h2, 1};
    const int S = d.S, ns = NS_ ? NS_ : (S + 15) >> 4, n1 = d.h1 >> 4, n2 = d.h2 >> 4;
    float *W2c = smem, *W1c = W2c + kFwdW, *W3c = W1c + kFwdW, *sb = W3c + kFwdW3;
    const int ld1 = lds_ld(16 * ns), ld2 = lds_ld(d.h1), ld3 = lds_ld(d.h2);

    const int64_t ntiles = (g.rows + 15) / 16;
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    const int64_t tstep = (int64_t)gridDim.x * NW;
    float4 XR[8];
    {
        const int64_t row = min(tile * 16 + l15, g.rows - 1);
        load_rows_raw<VEC>(XR, g.states + row * S, ns, S, q);
    }
    load_network<VEC, NW>(g, d, ns, W2c, W1c, W3c, sb, tid);
    lds_barrier();
    for (; tile < ntiles; tile += tstep) {
        const int64_t row = tile * 16 + l15;
        const bool valid = row < g.rows;
        f32x4 X[8], H1[8], H2[8], Y[8], dummy[8];
        normalise_rows<VEC>(XR, X, g.avg, g.sd, ns, S, q, valid);
        if (tile + tstep < ntiles) {   // prefetch the next tile's rows under this tile's MFMAs
            const int64_t nrow = min((tile + tstep) * 16 + l15, g.rows - 1);
            load_rows_raw<VEC>(XR, g.states + nrow * S, ns, S, q);
        }
        forward_layer<true, NS_, false>(W1c, ld1, sb, ns, n1, X, H1, dummy, l15, q);
        forward_layer<true, N1_, false>(W2c, ld2, sb + 128, n1, n2, H1, H2, dummy, l15, q);
        forward_layer<false, N2_>(W3c, ld3, sb + 256, n2, 1, H2, Y, dummy, l15, q);
        if (valid && q == 0) g.values[row] = Y[0][0];
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------------------
template <int NS_, int N1_, int N2_, bool VEC>
__global__ __launch_bounds__(256) void rollout_step2_kernel(FwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const Dims d{g.S, N1_ ? 16 * N1_ : g.h1, N2_ ? 16 * N2_ : g.h2, g.out};
    const int S = d.S, A = d.out, ns = NS_ ? NS_ : (S + 15) >> 4, n1 = d.h1 >> 4, n2 = d.h2 >> 4;
    float *W2c = smem, *W1c = W2c + kFwdW, *W3c = W1c + kFwdW, *sb = W3c + kFwdW3;
    const int ld1 = lds_ld(16 * ns), ld2 = lds_ld(d.h1), ld3 = lds_ld(d.h2);
    const float *std_log = g.P + d.oStd();

    const int64_t env = ((int64_t)blockIdx.x * NW + wave) * 16 + l15;
    const bool valid = env < g.rows;
    const int64_t row = valid ? env : g.rows - 1;
    float4 XR[8];
    load_rows_raw<VEC>(XR, g.states + row * S, ns, S, q);
    // this lane's action slots a = 4 q + r: noise and std
    float eps[4], sl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * q + r, ac = min(a, A - 1);
        sl[r] = std_log[ac];
        eps[r] = g.noise ? g.noise[row * A + ac] : philox_normal(g.seed, g.counter, (uint32_t)row, (uint32_t)ac);
    }
    load_network<VEC, NW>(g, d, ns, W2c, W1c, W3c, sb, tid);
    if (g.o_state && valid) {   // states[t] = state: raw rows, the lane's 4-float groups
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t < ns) {
                const int k0 = 16 * t + 4 * q;
                float *dst = g.o_state + row * S + k0;
                if (VEC) { if (k0 < S) *reinterpret_cast<float4 *>(dst) = XR[t]; }
                else {
                    const float xr[4] = {XR[t].x, XR[t].y, XR[t].z, XR[t].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (k0 + c < S) dst[c] = xr[c];
                }
            }
        }
    }
    f32x4 X[8], H1[8], H2[8], Y[8], dummy[8];
    normalise_rows<VEC>(XR, X, g.avg, g.sd, ns, S, q, valid);
    lds_barrier();
    forward_layer<true, NS_, false>(W1c, ld1, sb, ns, n1, X, H1, dummy, l15, q);
    forward_layer<true, N1_, false>(W2c, ld2, sb + 128, n1, n2, H1, H2, dummy, l15, q);
    forward_layer<false, N2_>(W3c, ld3, sb + 256, n2, 1, H2, Y, dummy, l15, q);

    // sample: a = mean + std * eps (torch.normal(mean, std)); Normal.log_prob summed over the action dims
    float lp = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * q + r;
        const bool on = a < A;
        const float sdv = expf(sl[r]), var = sdv * sdv;
        const float act = Y[0][r] + sdv * eps[r];
        const float diff = act - Y[0][r];
        const float term = -(diff * diff) / (2.f * var) - logf(sdv) - kLogSqrt2PiF;
        lp += on ? term : 0.f;
        if (on && valid) {
            if (g.o_action) g.o_action[row * A + a] = act;
            if (g.o_env) g.o_env[row * A + a] = tanhf(act);   // convert_action_for_env
        }
    }
    lp += __shfl_xor(lp, 16, 64);
    lp += __shfl_xor(lp, 32, 64);
    if (valid && q == 0 && g.o_logprob) g.o_logprob[row] = lp;
}

// ---------------------------------------------------------------------------------------------------------
// K1, latency form (N of a few thousand envs: one 16-env tile per workgroup fills the chip, and the step is a
// dependent chain, so what counts is the length of that chain, not throughput).  The 8 waves of a workgroup split
// the OUTPUT features of each layer: wave w owns feature tile w (rows 16 w .. 16 w + 15 of W1 and of W2) and reads
// exactly those weight rows straight from L2 into registers as MFMA A operands -- all of them, together with the
// state tile, in ONE round trip issued before anything else; no LDS weight image, no staging barrier.  Both hidden layers
// run on v_mfma_f32_16x16x32_bf16 from three-way bf16 splits of both operands (rollout_bf16.h: six partial products, fp32
// accumulation -- as close to fp64 as the fp32 instruction, 2.7x less matrix-pipe time), instruction for instruction what
// the persistent rollout (rollout_fused.hip) does per step.
//   L1: 16 x 16 tile of H1^T per wave (6 MFMAs per 32 state columns) -> split -> LDS tile T1[part][sample][feature] -> barrier
//   L2: every wave reads all of H1 back as B operands (3 ds_read_b128 per 32 features), 6 MFMAs each -> its H2^T tile in registers
//   out: the wave's H2 tile IS the k-slice 16 w .. 16 w + 15 of the output layer: 4 MFMAs give a partial Y^T;
//        the 8 partials meet in LDS, wave 0 adds them in a fixed order and does the sampling / log-prob / stores.
// ---------------------------------------------------------------------------------------------------------
template <int NS_, int N1_, int N2_, bool VEC>
__global__ __launch_bounds__(512) void rollout_split_kernel(FwdArgs g)
{
    __shared__ __attribute__((aligned(16))) u8 T1[RB_TBYTES];
    __shared__ __attribute__((aligned(16))) float PS[8 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const Dims d{g.S, N1_ ? 16 * N1_ : g.h1, N2_ ? 16 * N2_ : g.h2, g.out};
    const int S = d.S, A = d.out, n1 = d.h1 >> 4, n2 = d.h2 >> 4;
    const int ks_s = NS_ ? (NS_ + 1) >> 1 : (S + 31) >> 5, ks_1 = d.h1 >> 5;      // k-steps of 32 (rollout_bf16.h)
    const bool on1 = wave < n1, on2 = wave < n2;
    const float *std_log = g.P + d.oStd();

    const int64_t env = (int64_t)blockIdx.x * 16 + l15;
    const bool valid = env < g.rows;
    const int64_t row = valid ? env : g.rows - 1;

    RPROF(0);
    // ---- every global load of the step, issued back to back: the lane's 8-column groups k = 32 ks + 8 q .. + 7 ----
    float4 XR[8], w1r[8], w2r[8];
    {
        const float *srow = g.states + row * S;
        const float *r1 = g.P + d.oW1() + (size_t)min(16 * wave + l15, d.h1 - 1) * S;
        const float *r2 = g.P + d.oW2() + (size_t)min(16 * wave + l15, d.h2 - 1) * d.h1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k0 = 32 * ks + 8 * q;
            if (ks < ks_s) {
                XR[2 * ks] = load4<VEC>(srow, k0, S);
                XR[2 * ks + 1] = load4<VEC>(srow, k0 + 4, S);
                w1r[2 * ks] = load4<VEC>(r1, k0, S);
                w1r[2 * ks + 1] = load4<VEC>(r1, k0 + 4, S);
            }
            if (ks < ks_1) {
                w2r[2 * ks] = load4<VEC>(r2, k0, d.h1);
                w2r[2 * ks + 1] = load4<VEC>(r2, k0 + 4, d.h1);
            }
        }
    }
    const int kt = min(wave, n2 - 1);                      // this wave's k-tile of the output layer
    float4 w3 = load4<VEC>(g.P + d.oW3() + (size_t)min(l15, A - 1) * d.h2, 16 * kt + 4 * q, d.h2);
    if (l15 >= A || !on2) w3 = zero4();
    const float4 b1 = load4<VEC>(g.P + d.ob1(), 16 * min(wave, n1 - 1) + 4 * q, d.h1);
    const float4 b2 = load4<VEC>(g.P + d.ob2(), 16 * kt + 4 * q, d.h2);
    float eps[4], sl[4], b3[4];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * q + r, ac = min(a, A - 1);
            sl[r] = std_log[ac];
            b3[r] = g.P[d.ob3() + ac];
            eps[r] = g.noise ? g.noise[row * A + ac] : philox_normal(g.seed, g.counter, (uint32_t)row, (uint32_t)ac);
        }
    }
    if (wave == 7 && g.o_state && valid) {   // states[t] = state: raw rows, the lane's 4-float groups
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if ((t >> 1) < ks_s) {
                const int k0 = 32 * (t >> 1) + 8 * q + 4 * (t & 1);
                float *dst = g.o_state + row * S + k0;
                if (VEC) { if (k0 < S) *reinterpret_cast<float4 *>(dst) = XR[t]; }
                else {
                    const float xr[4] = {XR[t].x, XR[t].y, XR[t].z, XR[t].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (k0 + c < S) dst[c] = xr[c];
                }
            }
        }
    }
    RPROF(1);
    // (s - avg) / (std + 1e-4), AgentPPO.py:360-361; columns >= S and rows past the end are zero
    Parts X[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks < ks_s) {
            float4 xn[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int k0 = 32 * ks + 8 * q + 4 * hh;
                const float4 a4 = load4<VEC>(g.avg, k0, S), s4 = load4<VEC>(g.sd, k0, S), x4 = XR[2 * ks + hh];
                xn[hh].x = (valid && k0 + 0 < S) ? (x4.x - a4.x) / (s4.x + 1e-4f) : 0.f;
                xn[hh].y = (valid && k0 + 1 < S) ? (x4.y - a4.y) / (s4.y + 1e-4f) : 0.f;
                xn[hh].z = (valid && k0 + 2 < S) ? (x4.z - a4.z) / (s4.z + 1e-4f) : 0.f;
                xn[hh].w = (valid && k0 + 3 < S) ? (x4.w - a4.w) / (s4.w + 1e-4f) : 0.f;
            }
            X[ks] = rb_split8(xn[0], xn[1]);
        }
    }
    RPROF(2);

    // ---- L1: this wave's feature tile of H1^T ----
    {
        RbAcc acc;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < ks_s) rb_mma6(rb_split8(w1r[2 * ks], w1r[2 * ks + 1]), X[ks], acc);
        if (on1) {
            const float bb[4] = {b1.x, b1.y, b1.z, b1.w};
            float h[4], gd;
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
            rb_tile_put(T1, RB_TLD, l15, 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
        }
    }
    RPROF(3);
    lds_barrier();
    RPROF(4);

    // ---- L2: all of H1 back as B operands, this wave's feature tile of H2^T stays in registers ----
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    {
        RbAcc acc;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            if (ks < ks_1) rb_mma6(rb_split8(w2r[2 * ks], w2r[2 * ks + 1]), rb_tile_get(T1, RB_TLD, l15, ks, q), acc);
        const float bb[4] = {b2.x, b2.y, b2.z, b2.w};
        float h[4], gd;
#pragma unroll
        for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
        // ---- output layer, k-slice 16 w + 4 q + r: the B operand is the tile just computed ----
        part = mfma16(w3.x, h[0], part);
        part = mfma16(w3.y, h[1], part);
        part = mfma16(w3.z, h[2], part);
        part = mfma16(w3.w, h[3], part);
    }
    *reinterpret_cast<float4 *>(PS + (wave * 64 + lane) * 4) =
        on2 ? make_float4(part[0], part[1], part[2], part[3]) : zero4();
    RPROF(5);
    lds_barrier();
    RPROF(6);
    if (wave != 0) return;

    float Y[4];
    {
        float4 p[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) p[w] = *reinterpret_cast<const float4 *>(PS + (w * 64 + lane) * 4);
        Y[0] = ((p[0].x + p[1].x) + (p[2].x + p[3].x)) + ((p[4].x + p[5].x) + (p[6].x + p[7].x)) + b3[0];
        Y[1] = ((p[0].y + p[1].y) + (p[2].y + p[3].y)) + ((p[4].y + p[5].y) + (p[6].y + p[7].y)) + b3[1];
        Y[2] = ((p[0].z + p[1].z) + (p[2].z + p[3].z)) + ((p[4].z + p[5].z) + (p[6].z + p[7].z)) + b3[2];
        Y[3] = ((p[0].w + p[1].w) + (p[2].w + p[3].w)) + ((p[4].w + p[5].w) + (p[6].w + p[7].w)) + b3[3];
    }
    // sample: a = mean + std * eps (torch.normal(mean, std)); Normal.log_prob summed over the action dims
    float lp = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * q + r;
        const bool on = a < A;
        const float sdv = expf(sl[r]), var = sdv * sdv;
        const float act = Y[r] + sdv * eps[r];
        const float diff = act - Y[r];
        const float term = -(diff * diff) / (2.f * var) - sl[r] - kLogSqrt2PiF;      // log(exp(std_log)) = std_log
        lp += on ? term : 0.f;
        if (on && valid) {
            if (g.o_action) g.o_action[row * A + a] = act;
            if (g.o_env) g.o_env[row * A + a] = fast_tanh(act);   // convert_action_for_env
        }
    }
    lp += __shfl_xor(lp, 16, 64);
    lp += __shfl_xor(lp, 32, 64);
    if (valid && q == 0 && g.o_logprob) g.o_logprob[row] = lp;
    RPROF(7);
}

// sum the per-workgroup slabs into the flat gradient.  Deterministic: element e is summed by 4 threads (slab
// quarter p = threadIdx.x / 64 takes slabs k = p (mod 4)... in ascending order, 8 loads in flight), combined in a
// fixed order through LDS.  Latency bound (26 MB, 128 strided rows): the split buys 4x the loads in flight.
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float *__restrict__ slabs, int n_slabs, int64_t stride,
                                                          float *__restrict__ flat)
{
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < stride) {
        const float *src = slabs + i;
        int k = p;
        for (; k + 124 < n_slabs; k += 128) {      // 32 loads in flight: one round trip per 128 slabs (config 4: all of them)
            float x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = __builtin_nontemporal_load(src + (size_t)(k + 4 * u) * stride);
#pragma unroll
            for (int v = 0; v < 4; ++v)            // same association as the 8-wide loop below: s[u] += slab k + 32 v + 4 u
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += x[8 * v + u];
        }
        for (; k + 28 < n_slabs; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += src[(size_t)(k + 4 * u) * stride];
        }
        for (; k < n_slabs; k += 4) s[0] += src[(size_t)k * stride];
    }
    part[p][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (p == 0 && i < stride) flat[i] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

// ---------------------------------------------------------------------------------------------
// MFMA helper self-test: C (32x32) = A (32xK) . B (Kx32) through the same lane mapping as layer_forward
// ---------------------------------------------------------------------------------------------
__global__ void selftest_kernel(const float *A, const float *B, int K, float *C)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 8) {
        const int kb = k0 + 4 * hi;
        for (int j = 0; j < 4; ++j) acc = mfma32(A[l31 * K + kb + j], B[(kb + j) * 32 + l31], acc);
    }
    for (int r = 0; r < 16; ++r) C[crow(r, hi) * 32 + l31] = acc[r];
}


int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <typename Kern>
int prep_lds(Kern kern, bool *done)
{
    if (*done) return ERL_OK;
    int rc = erl_hip_status(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLdsBytes),
                            "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    if (!rc) *done = true;
    return rc;
}

bool vec_ok(const FwdArgs &g)
{
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (g.S % 4 == 0) && al(g.P) && al(g.states) && al(g.avg) && al(g.sd) && (!g.o_state || al(g.o_state));
}

}  // namespace

extern "C" int64_t erl_mlp_param_count(int S, int h1, int h2, int out, int with_std_log)
{
    // (net_dims = (256, h2): the minibatch kernel of ppo_step_wd.hip only -- rollouts of that shape take the layered erl_mlpn_* path)
    if (!mlp_dims_ok(S, h1, h2, out) && !erl_ppo_wd_supported(S, h1, h2, out)) return -1;
    return Dims{S, h1, h2, out}.count(with_std_log != 0);
}

extern "C" int erl_value_forward_f32(const float *critic_params, const float *state_avg, const float *state_std, int S, int h1,
                                     int h2, const float *states, int64_t rows, float *values, void *stream)
{
    ERL_REQUIRE(critic_params && state_avg && state_std && states && values, "erl_value_forward_f32: NULL tensor");
    ERL_REQUIRE(mlp_dims_ok(S, h1, h2, 1), "erl_value_forward_f32: unsupported dims S=%d net=[%d,%d]", S, h1, h2);
    ERL_REQUIRE(rows >= 0, "erl_value_forward_f32: rows < 0");
    if (rows == 0) return ERL_OK;
    FwdArgs g{};
    g.P = critic_params; g.avg = state_avg; g.sd = state_std;
    g.S = S; g.h1 = h1; g.h2 = h2; g.out = 1;
    g.states = states; g.rows = rows; g.values = values;
    const int64_t tiles = erl_cdiv(rows, 16);
    const int64_t want = erl_cdiv(tiles, 8);
    const int grid = (int)(want < num_cus() ? want : num_cus());
    const bool vec = vec_ok(g);
    const int ns = (S + 15) / 16;
    static bool d0 = false, d1 = false, d2 = false;
    int rc;
#define VF_LAUNCH(K, FLAG)                                                                        \
    do {                                                                                          \
        if ((rc = prep_lds(K, &FLAG))) return rc;                                                 \
        hipLaunchKernelGGL(K, dim3(grid), dim3(512), kFwdLdsBytes, (hipStream_t)stream, g);       \
    } while (0)
    if (vec && ns == 4 && h1 == 128 && h2 == 128) VF_LAUNCH((value_forward2_kernel<4, 8, 8, true>), d0);
    else if (vec) VF_LAUNCH((value_forward2_kernel<0, 0, 0, true>), d1);
    else VF_LAUNCH((value_forward2_kernel<0, 0, 0, false>), d2);
#undef VF_LAUNCH
    ERL_LAUNCH_CHECK("erl_value_forward_f32");
}

extern "C" int erl_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std, int S, int h1,
                                    int h2, int A, const float *state, int64_t N, const float *noise, uint64_t seed,
                                    uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                                    float *out_action_env, void *stream)
{
    ERL_REQUIRE(actor_params && state_avg && state_std && state, "erl_rollout_step_f32: NULL tensor");
    ERL_REQUIRE(mlp_dims_ok(S, h1, h2, A), "erl_rollout_step_f32: unsupported dims S=%d net=[%d,%d] A=%d", S, h1, h2, A);
    ERL_REQUIRE(N >= 1, "erl_rollout_step_f32: N < 1");
    FwdArgs g{};
    g.P = actor_params; g.avg = state_avg; g.sd = state_std;
    g.S = S; g.h1 = h1; g.h2 = h2; g.out = A;
    g.states = state; g.rows = N;
    g.noise = noise; g.seed = seed; g.counter = counter;
    g.o_state = out_state_row; g.o_action = out_action_row; g.o_logprob = out_logprob_row; g.o_env = out_action_env;
#ifdef ERL_PROFILE
    g.prof = g_fwd_prof;
#endif
    const bool vec = vec_ok(g);
    const int ns = (S + 15) / 16;
    // up to kSplitMaxEnvs envs the step is latency bound: one 16-env tile per workgroup, output features split over
    // the waves (rollout_split_kernel).  Beyond that the weight re-reads (one network per tile) outweigh the shorter
    // chain and the throughput form (4 tiles per workgroup behind one LDS weight image) takes over.
    static const int split_mode = [] { const char *e = getenv("ERL_ROLLOUT_SPLIT"); return e ? atoi(e) : -1; }();
    if (split_mode == 1 || (split_mode != 0 && N <= kSplitMaxEnvs)) {
        const unsigned sgrid = (unsigned)erl_cdiv(N, 16);
        if (vec && ns == 4 && h1 == 128 && h2 == 128)
            hipLaunchKernelGGL((rollout_split_kernel<4, 8, 8, true>), dim3(sgrid), dim3(512), 0, (hipStream_t)stream, g);
        else if (vec)
            hipLaunchKernelGGL((rollout_split_kernel<0, 0, 0, true>), dim3(sgrid), dim3(512), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL((rollout_split_kernel<0, 0, 0, false>), dim3(sgrid), dim3(512), 0, (hipStream_t)stream, g);
        ERL_LAUNCH_CHECK("erl_rollout_step_f32");
    }
    const int grid = (int)erl_cdiv(N, 64);
    static bool d0 = false, d1 = false, d2 = false;
    int rc;
#define RS_LAUNCH(K, FLAG)                                                                        \
    do {                                                                                          \
        if ((rc = prep_lds(K, &FLAG))) return rc;                                                 \
        hipLaunchKernelGGL(K, dim3(grid), dim3(256), kFwdLdsBytes, (hipStream_t)stream, g);       \
    } while (0)
    if (vec && ns == 4 && h1 == 128 && h2 == 128) RS_LAUNCH((rollout_step2_kernel<4, 8, 8, true>), d0);
    else if (vec) RS_LAUNCH((rollout_step2_kernel<0, 0, 0, true>), d1);
    else RS_LAUNCH((rollout_step2_kernel<0, 0, 0, false>), d2);
#undef RS_LAUNCH
    ERL_LAUNCH_CHECK("erl_rollout_step_f32");
}



#ifdef ERL_PROFILE
// profiling builds only (make EXTRA=-DERL_PROFILE): device buffer of 8 * 16 int64 cycle stamps
extern "C" __attribute__((visibility("default"))) void erl_debug_set_rollout_profile(long long *dev_buf) { g_fwd_prof = dev_buf; }
#endif

extern "C" int erl_grad_reduce_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, void *stream)
{
    ERL_REQUIRE(slabs && flat_grad && n_slabs >= 1 && stride >= 1, "erl_grad_reduce_f32: bad argument");
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)erl_cdiv(stride, 64)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                       stride, flat_grad);
    ERL_LAUNCH_CHECK("erl_grad_reduce_f32");
}

extern "C" int erl_selftest_mfma(float *max_err)
{
    ERL_REQUIRE(max_err, "erl_selftest_mfma: NULL");
    const int K = 24;
    float hA[32 * K], hB[K * 32], hC[32 * 32];
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (float &x : hA) x = rnd();
    for (float &x : hB) x = rnd();
    float *dA, *dB, *dC;
    int rc;
    if ((rc = erl_hip_status(hipMalloc(&dA, sizeof(hA)), "hipMalloc"))) return rc;
    if ((rc = erl_hip_status(hipMalloc(&dB, sizeof(hB)), "hipMalloc"))) return rc;
    if ((rc = erl_hip_status(hipMalloc(&dC, sizeof(hC)), "hipMalloc"))) return rc;
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    rc = erl_hip_status(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost), "selftest memcpy");
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    if (rc) return rc;
    float worst = 0.f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[i * K + k] * hB[k * 32 + j];
            const float e = fabsf((float)ref - hC[i * 32 + j]);
            worst = e > worst ? e : worst;
        }
    *max_err = worst;
    return ERL_OK;
}
