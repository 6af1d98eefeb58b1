// One vectorised rollout step for the reference's wider demo networks, net_dims = (256, 32..128) and (256, 32..128, 32..128): ActorPPO.get_action + the three buffer
// stores + convert_action_for_env (elegantrl/agents/AgentPPO.py:113-119, :368-376, :388-390; examples/demo_A2C_PPO.py:117, :171, :224 train
// net_dims = (256, 128), (256, 128, 64), (256, 128, 128)) as ONE launch instead of the layered path's five (normalise, three GEMMs, sample: ~45 us per 4096-env step).
//
// The latency form of K1 (mlp.hip rollout_split_kernel) with a first layer twice as wide: one 16-env tile per workgroup, the 8 waves split
// the OUTPUT features of each layer -- wave w owns first-layer feature tiles w and w + 8 (rows 16 w .. and 128 + 16 w .. of W1) and
// second-layer tile w -- and read exactly those weight rows from L2 into registers as MFMA A operands, all of them with the state tile in
// one round trip.  Both hidden layers run on v_mfma_f32_16x16x32_bf16 from three-way bf16 splits of both operands (rollout_bf16.h: six
// partial products, fp32 accumulation); H1 crosses the waves through a split [part][sample][256 features] LDS tile (one barrier); the
// wave's H2 tile is its k-slice of the output layer (fp32 MFMA), the 8 partial outputs meet in LDS, wave 0 samples and stores.
// Reached through erl_mlpn_rollout_step_f32 (the layered path's entry point, which AgentPPO already calls for this shape); ERL_WIDE_FUSED=0
// keeps the layered launches.  Same Philox keys as every other rollout kernel (seed, step, env, action-dim).
#include <cstdlib>

#include "mlp_chain.h"
#include "rollout_bf16.h"

namespace {

constexpr int RW_H1 = 256;
constexpr int RW_TLD = 2 * RW_H1 + 16;          // bytes per sample row of one part plane of the H1 tile: rows 132 dwords apart
constexpr int RW_TBYTES = 3 * 16 * RW_TLD;      // 25344

struct RwArgs {
    const float *P, *avg, *sd;
    int S, h2, h3, A;              // h3 = 0: two hidden layers
    const float *states;
    int64_t rows;
    const float *noise;
    uint64_t seed, counter;
    float *o_state, *o_action, *o_logprob, *o_env;
};

// parameter block of build_mlp([S, 256, h2, (h3,) A]) + action_std_log (include/erl_hip.h): W1 b1 W2 b2 (W3 b3) Wout bout std_log
struct RwOff {
    int S, h2, h3, A, hl;          // hl = width of the last hidden layer
    __host__ __device__ RwOff(int S_, int h2_, int h3_, int A_) : S(S_), h2(h2_), h3(h3_), A(A_), hl(h3_ ? h3_ : h2_) {}
    __host__ __device__ size_t oW1() const { return 0; }
    __host__ __device__ size_t ob1() const { return (size_t)RW_H1 * S; }
    __host__ __device__ size_t oW2() const { return ob1() + RW_H1; }
    __host__ __device__ size_t ob2() const { return oW2() + (size_t)h2 * RW_H1; }
    __host__ __device__ size_t oW3() const { return ob2() + h2; }
    __host__ __device__ size_t ob3() const { return oW3() + (size_t)h3 * h2; }
    __host__ __device__ size_t oWo() const { return h3 ? ob3() + h3 : ob2() + h2; }
    __host__ __device__ size_t obo() const { return oWo() + (size_t)A * hl; }
    __host__ __device__ size_t oStd() const { return obo() + A; }
};

template <bool VEC, bool L3>
__global__ __launch_bounds__(512) void rollout_wide_kernel(RwArgs g)
{
    __shared__ __attribute__((aligned(16))) u8 T1[RW_TBYTES];
    __shared__ __attribute__((aligned(16))) u8 T2[L3 ? RB_TBYTES : 16];
    __shared__ __attribute__((aligned(16))) float PS[8 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const RwOff d(g.S, g.h2, L3 ? g.h3 : 0, g.A);
    const int S = d.S, A = d.A, n2 = d.h2 >> 4, nl = d.hl >> 4;
    const int ks_s = (S + 31) >> 5;                        // k-steps of 32 state columns (S <= 64: 1 or 2)
    const bool onl = wave < nl;
    const float *std_log = g.P + d.oStd();

    const int64_t env = (int64_t)blockIdx.x * 16 + l15;
    const bool valid = env < g.rows;
    const int64_t row = valid ? env : g.rows - 1;

    // ---- every global load of the step, issued back to back: the lane's 8-column groups k = 32 ks + 8 q .. + 7 ----
    float4 XR[4], w1r[2][4], w2r[16];
    {
        const float *srow = g.states + row * S;
        const float *r1a = g.P + d.oW1() + (size_t)(16 * wave + l15) * S, *r1b = r1a + (size_t)128 * S;
        const float *r2 = g.P + d.oW2() + (size_t)min(16 * wave + l15, d.h2 - 1) * RW_H1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k0 = 32 * ks + 8 * q;
            if (ks < ks_s) {
                XR[2 * ks] = load4<VEC>(srow, k0, S);
                XR[2 * ks + 1] = load4<VEC>(srow, k0 + 4, S);
                w1r[0][2 * ks] = load4<VEC>(r1a, k0, S);
                w1r[0][2 * ks + 1] = load4<VEC>(r1a, k0 + 4, S);
                w1r[1][2 * ks] = load4<VEC>(r1b, k0, S);
                w1r[1][2 * ks + 1] = load4<VEC>(r1b, k0 + 4, S);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            w2r[2 * ks] = load4<true>(r2, 32 * ks + 8 * q, RW_H1);
            w2r[2 * ks + 1] = load4<true>(r2, 32 * ks + 8 * q + 4, RW_H1);
        }
    }
    const int kt = min(wave, nl - 1);                      // this wave's k-tile of the output layer
    float4 wo = load4<VEC>(g.P + d.oWo() + (size_t)min(l15, A - 1) * d.hl, 16 * kt + 4 * q, d.hl);
    if (l15 >= A || !onl) wo = zero4();
    const float4 b1a = load4<true>(g.P + d.ob1(), 16 * wave + 4 * q, RW_H1), b1b = load4<true>(g.P + d.ob1(), 128 + 16 * wave + 4 * q, RW_H1);
    const float4 b2 = load4<VEC>(g.P + d.ob2(), 16 * min(wave, n2 - 1) + 4 * q, d.h2);
    float4 w3r[8], b3h = zero4();                          // third hidden layer: rows 16 w .. of W3 (h3 x h2), K = h2 <= 128
    if (L3) {
        const float *r3 = g.P + d.oW3() + (size_t)min(16 * wave + l15, d.h3 - 1) * d.h2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            w3r[2 * ks] = load4<VEC>(r3, 32 * ks + 8 * q, d.h2);
            w3r[2 * ks + 1] = load4<VEC>(r3, 32 * ks + 8 * q + 4, d.h2);
        }
        b3h = load4<VEC>(g.P + d.ob3(), 16 * kt + 4 * q, d.h3);
    }
    float eps[4], sl[4], b3[4];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * q + r, ac = min(a, A - 1);
            sl[r] = std_log[ac];
            b3[r] = g.P[d.obo() + ac];
            eps[r] = g.noise ? g.noise[row * A + ac] : philox_normal(g.seed, g.counter, (uint32_t)row, (uint32_t)ac);
        }
    }
    if (wave == 7 && g.o_state && valid) {   // states[t] = state: raw rows, the lane's 4-float groups
#pragma unroll
        for (int t = 0; t < 4; ++t) {
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
    // (s - avg) / (std + 1e-4), AgentPPO.py:360-361; columns >= S and rows past the end are zero
    Parts X[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
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

    // ---- L1: this wave's two feature tiles of H1^T ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        RbAcc acc;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            if (ks < ks_s) rb_mma6(rb_split8(w1r[j][2 * ks], w1r[j][2 * ks + 1]), X[ks], acc);
        const float4 b1 = j ? b1b : b1a;
        const float bb[4] = {b1.x, b1.y, b1.z, b1.w};
        float h[4], gd;
#pragma unroll
        for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
        rb_tile_put(T1, RW_TLD, l15, 128 * j + 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
    }
    lds_barrier();

    // ---- L2: all of H1 back as B operands, this wave's feature tile of H2^T stays in registers ----
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    float h[4] = {0.f, 0.f, 0.f, 0.f}, gd;
    if (__builtin_amdgcn_readfirstlane(wave) < n2) {      // (waves beyond h2 / 16 hold no rows)
        RbAcc acc;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) rb_mma6(rb_split8(w2r[2 * ks], w2r[2 * ks + 1]), rb_tile_get(T1, RW_TLD, l15, ks, q), acc);
        const float bb[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
        if (L3) rb_tile_put(T2, RB_TLD, l15, 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
    }
    if (L3) {
        // ---- L3: H2 crosses the waves like H1 did; this wave's feature tile of H3^T ----
        lds_barrier();
        if (__builtin_amdgcn_readfirstlane(wave) < nl) {
            RbAcc acc;
            const int ks_2 = d.h2 >> 5;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < ks_2) rb_mma6(rb_split8(w3r[2 * ks], w3r[2 * ks + 1]), rb_tile_get(T2, RB_TLD, l15, ks, q), acc);
            const float bb[4] = {b3h.x, b3h.y, b3h.z, b3h.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
        }
    }
    if (__builtin_amdgcn_readfirstlane(wave) < nl) {
        // ---- output layer, k-slice 16 w + 4 q + r: the B operand is the tile just computed ----
        part = mfma16(wo.x, h[0], part);
        part = mfma16(wo.y, h[1], part);
        part = mfma16(wo.z, h[2], part);
        part = mfma16(wo.w, h[3], part);
    }
    *reinterpret_cast<float4 *>(PS + (wave * 64 + lane) * 4) = make_float4(part[0], part[1], part[2], part[3]);
    lds_barrier();
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
}

// ---------------------------------------------------------------------------------------------------------
// The value pre-pass of the same shapes: CriticPPO.forward over the rollout's H x N rows (elegantrl/agents/AgentPPO.py:141-143, :219-220,
// :435-441) as ONE launch instead of the layered path's four (normalise + three GEMMs through memory: ~150 us at 131 072 rows).  Persistent
// workgroups walk 16-row tiles; the wave's weight rows (first-layer tiles w and w + 8, second-layer tile w) are split into their bf16 parts
// ONCE and stay in registers (48 + 96 per lane; a third layer's rows are re-read and re-split per tile), a tile costs three LDS barriers.
// ---------------------------------------------------------------------------------------------------------
struct VwArgs {
    const float *P, *avg, *sd;
    int S, h2, h3;
    const float *states;
    int64_t rows;
    float *values;
};

template <bool VEC, bool L3>
__global__ __launch_bounds__(512) void value_wide_kernel(VwArgs g)
{
    __shared__ __attribute__((aligned(16))) u8 T1[RW_TBYTES];
    __shared__ __attribute__((aligned(16))) u8 T2[L3 ? RB_TBYTES : 16];
    __shared__ __attribute__((aligned(16))) float PS[8 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const RwOff d(g.S, g.h2, L3 ? g.h3 : 0, 1);
    const int S = d.S, n2 = d.h2 >> 4, nl = d.hl >> 4;
    const int ks_s = (S + 31) >> 5;

    Parts w1[2][2], w2[8];          // (three hidden layers: the first layer's rows are re-read and re-split per tile too -- 256 registers)
    float4 b3h = zero4();
    const float *r1a = g.P + d.oW1() + (size_t)(16 * wave + l15) * S, *r1b = r1a + (size_t)128 * S;
    {
        const float *r2 = g.P + d.oW2() + (size_t)min(16 * wave + l15, d.h2 - 1) * RW_H1;
        if (!L3) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks < ks_s) {
                    w1[0][ks] = rb_load_w<VEC>(r1a, ks, q, S);
                    w1[1][ks] = rb_load_w<VEC>(r1b, ks, q, S);
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w2[ks] = rb_load_w<true>(r2, ks, q, RW_H1);
    }
    const int kt = min(wave, nl - 1);
    float4 wo = load4<VEC>(g.P + d.oWo(), 16 * kt + 4 * q, d.hl);            // the value head: ONE output row
    if (l15 >= 1 || wave >= nl) wo = zero4();
    const float4 b1a = load4<true>(g.P + d.ob1(), 16 * wave + 4 * q, RW_H1), b1b = load4<true>(g.P + d.ob1(), 128 + 16 * wave + 4 * q, RW_H1);
    const float4 b2 = load4<VEC>(g.P + d.ob2(), 16 * min(wave, n2 - 1) + 4 * q, d.h2);
    const float *r3 = g.P + d.oW3() + (size_t)min(16 * wave + l15, (L3 ? d.h3 : 1) - 1) * d.h2;      // (third layer: re-read and re-split per tile)
    if (L3) b3h = load4<VEC>(g.P + d.ob3(), 16 * kt + 4 * q, d.h3);
    const float bo = g.P[d.obo()];

    const int64_t ntiles = (g.rows + 15) / 16;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r_ = tile * 16 + l15;
        const bool valid = r_ < g.rows;
        const float *srow = g.states + (valid ? r_ : g.rows - 1) * S;
        // (loop-invariant loads that are meant to be re-issued per tile -- hoisted out of the loop they are 40-70 registers held across it,
        // which spill: the offset below is opaque to the compiler)
        int zoff = 0;
        asm volatile("" : "+s"(zoff));
        float4 w3r[8];
        if (L3) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                w3r[2 * ks] = load4<VEC>(r3 + zoff, 32 * ks + 8 * q, d.h2);
                w3r[2 * ks + 1] = load4<VEC>(r3 + zoff, 32 * ks + 8 * q + 4, d.h2);
            }
        }
        Parts X[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < ks_s) {
                float4 xn[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int k0 = 32 * ks + 8 * q + 4 * hh;
                    const float4 x4 = load4<VEC>(srow, k0, S), a4 = load4<VEC>(g.avg + zoff, k0, S), s4 = load4<VEC>(g.sd + zoff, k0, S);
                    xn[hh].x = (valid && k0 + 0 < S) ? (x4.x - a4.x) / (s4.x + 1e-4f) : 0.f;      // (s - avg) / (std + 1e-4), AgentPPO.py:440-441
                    xn[hh].y = (valid && k0 + 1 < S) ? (x4.y - a4.y) / (s4.y + 1e-4f) : 0.f;
                    xn[hh].z = (valid && k0 + 2 < S) ? (x4.z - a4.z) / (s4.z + 1e-4f) : 0.f;
                    xn[hh].w = (valid && k0 + 3 < S) ? (x4.w - a4.w) / (s4.w + 1e-4f) : 0.f;
                }
                X[ks] = rb_split8(xn[0], xn[1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            RbAcc acc;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                if (ks < ks_s) rb_mma6(L3 ? rb_load_w<VEC>((j ? r1b : r1a) + zoff, ks, q, S) : w1[j][ks], X[ks], acc);
            const float4 b1 = j ? b1b : b1a;
            const float bb[4] = {b1.x, b1.y, b1.z, b1.w};
            float h[4], gd;
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
            rb_tile_put(T1, RW_TLD, l15, 128 * j + 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
        }
        lds_barrier();
        f32x4 part = {0.f, 0.f, 0.f, 0.f};
        float h[4] = {0.f, 0.f, 0.f, 0.f}, gd;
        if (__builtin_amdgcn_readfirstlane(wave) < n2) {
            RbAcc acc;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) rb_mma6(w2[ks], rb_tile_get(T1, RW_TLD, l15, ks, q), acc);
            const float bb[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
            if (L3) rb_tile_put(T2, RB_TLD, l15, 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
        }
        if (L3) {
            lds_barrier();
            if (__builtin_amdgcn_readfirstlane(wave) < nl) {
                RbAcc acc;
                const int ks_2 = d.h2 >> 5;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < ks_2) rb_mma6(rb_split8(w3r[2 * ks], w3r[2 * ks + 1]), rb_tile_get(T2, RB_TLD, l15, ks, q), acc);
                const float bb[4] = {b3h.x, b3h.y, b3h.z, b3h.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(acc, r) + bb[r], h[r], gd);
            }
        }
        if (__builtin_amdgcn_readfirstlane(wave) < nl) {
            part = mfma16(wo.x, h[0], part);
            part = mfma16(wo.y, h[1], part);
            part = mfma16(wo.z, h[2], part);
            part = mfma16(wo.w, h[3], part);
        }
        if (q == 0) PS[wave * 16 + l15] = part[0];           // output row 0 of sample l15 (lanes q = 0 hold rows 0..3)
        lds_barrier();
        if (wave == 0 && q == 0 && valid) {
            float p[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) p[w] = PS[w * 16 + l15];
            g.values[r_] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + bo;
        }
        lds_barrier();                                         // (PS and T1 are rewritten by the next tile)
    }
}

constexpr int64_t kRwMaxEnvs = 16384;      // beyond: the weight re-reads (196 KB per 16-env tile) outweigh the launches saved

}  // namespace

// dims = [S, 256, h2, A] or [S, 256, h2, h3, A]: S <= 64, h2 / h3 in 32..128 (steps of 32), A <= 16, N <= 16 384 envs; ERL_WIDE_FUSED=0 turns
// the kernel off
int erl_rollout_wide_supported(const int *dims, int n_dims, int64_t N)
{
    static const bool on = [] { const char *e = getenv("ERL_WIDE_FUSED"); return !(e && atoi(e) == 0); }();
    if (!on || !dims || (n_dims != 4 && n_dims != 5) || N < 1 || N > kRwMaxEnvs) return 0;
    auto mid = [](int h) { return h >= 32 && h <= 128 && h % 32 == 0; };
    const int A = dims[n_dims - 1];
    return dims[0] >= 1 && dims[0] <= 64 && dims[1] == RW_H1 && mid(dims[2]) && (n_dims == 4 || mid(dims[3])) && A >= 1 && A <= 16;
}

int erl_rollout_wide_step(const float *actor_params, const float *state_avg, const float *state_std, const int *dims, int n_dims, const float *state, int64_t N,
                          const float *noise, uint64_t seed, uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                          float *out_action_env, hipStream_t stream)
{
    RwArgs g{};
    g.P = actor_params; g.avg = state_avg; g.sd = state_std;
    g.S = dims[0]; g.h2 = dims[2]; g.h3 = n_dims == 5 ? dims[3] : 0; g.A = dims[n_dims - 1];
    g.states = state; g.rows = N; g.noise = noise; g.seed = seed; g.counter = counter;
    g.o_state = out_state_row; g.o_action = out_action_row; g.o_logprob = out_logprob_row; g.o_env = out_action_env;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (g.S % 4 == 0) && al(g.P) && al(g.states) && al(g.avg) && al(g.sd) && (!g.o_state || al(g.o_state));
    const dim3 grid((unsigned)erl_cdiv(N, 16)), block(512);
    if (g.h3) {
        if (vec) hipLaunchKernelGGL((rollout_wide_kernel<true, true>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((rollout_wide_kernel<false, true>), grid, block, 0, stream, g);
    } else {
        if (vec) hipLaunchKernelGGL((rollout_wide_kernel<true, false>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((rollout_wide_kernel<false, false>), grid, block, 0, stream, g);
    }
    return erl_hip_status(hipGetLastError(), "erl_mlpn_rollout_step_f32 (wide latency form)");
}

// dims = [S, 256, h2, 1] or [S, 256, h2, h3, 1]; any number of rows (persistent workgroups); ERL_WIDE_FUSED=0 turns the kernel off
int erl_value_wide_supported(const int *dims, int n_dims)
{
    return dims && (n_dims == 4 || n_dims == 5) && dims[n_dims - 1] == 1 && erl_rollout_wide_supported(dims, n_dims, 1);
}

int erl_value_wide_forward(const float *params, const float *state_avg, const float *state_std, const int *dims, int n_dims, const float *states,
                           int64_t rows, float *values, hipStream_t stream)
{
    VwArgs g{};
    g.P = params; g.avg = state_avg; g.sd = state_std;
    g.S = dims[0]; g.h2 = dims[2]; g.h3 = n_dims == 5 ? dims[3] : 0;
    g.states = states; g.rows = rows; g.values = values;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (g.S % 4 == 0) && al(g.P) && al(g.states) && al(g.avg) && al(g.sd);
    const int64_t tiles = erl_cdiv(rows, 16);
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256)), block(512);
    if (g.h3) {
        if (vec) hipLaunchKernelGGL((value_wide_kernel<true, true>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((value_wide_kernel<false, true>), grid, block, 0, stream, g);
    } else {
        if (vec) hipLaunchKernelGGL((value_wide_kernel<true, false>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((value_wide_kernel<false, false>), grid, block, 0, stream, g);
    }
    return erl_hip_status(hipGetLastError(), "erl_mlpn_value_forward_f32 (wide form)");
}
