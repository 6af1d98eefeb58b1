// Persistent H-step rollout for device-resident environments: ONE launch per AgentPPO.explore_env.  gfx950; hidden layers on the
// bf16 matrix pipe (three-way operand split, rollout_bf16.h), output layers and the environment on the fp32 MFMA.
//
// Replaces the whole loop of AgentPPO._explore_vec_env (elegantrl/agents/AgentPPO.py:87-129): for t in range(H):
// ActorPPO.get_action (:368-376), the three buffer stores (:115-117), convert_action_for_env (:388-390), env.step, the
// reward / flag stores (:121-123); then `rewards *= reward_scale` and the two logical_not (:126-128).  It also evaluates
// CriticPPO (:435-441) on every state it visits -- the value pre-pass of update_net (:141-143) and the bootstrap value
// cri(last_state) (:219-220) -- so that update_net finds them ready (same critic weights: nothing trains during a rollout).
//
// Envs are independent, so a workgroup (8 waves) owns a 16-env tile for all H steps; nothing crosses workgroups and there
// is no launch, no weight re-read and no HBM round trip between steps:
//   * wave w holds rows 16 w .. 16 w + 15 of W2 of BOTH networks and of the actor's W1 in registers for the whole rollout, split
//     once into their three bf16 parts (A operands of v_mfma_f32_16x16x32_bf16: 2 x 48 + 24 VGPRs; the layout of the latency-form
//     step kernel, mlp.hip rollout_split_kernel), and its k-slice of the output layers (fp32); the critic's W1 (split), the
//     biases and -- SynVecEnv -- Ws^T / Wa^T sit in LDS, loaded once per launch: 256 registers per lane are all a wave has at two
//     waves per SIMD, and with a fourth weight block in them the step loop spilled;
//   * the state tile lives in LDS (XS, and normalised + split per network: XA / XC); per step: L1 of both nets -> H1 tiles (split) to LDS -> barrier -> L2 + output-layer
//     partials -> LDS -> barrier -> waves < ceil(S/16) finish the policy head in registers (action, log-prob, tanh) and
//     step the env on the matrix cores, wave 7 finishes the value, waves 4..7 draw the next step's N(0,1) (injected or
//     Philox4x32-10) -> barrier (SynVecEnv only) -> done flags, auto-reset, new state tile -> barrier.  Four (Pendulum: three) LDS-only barriers per step; the
//     per-step chain is MFMA-bound (~100 kFLOP per env-step);
//   * every buffer row is written straight from the kernel: states / actions (pre-tanh) / logprobs / rewards (already
//     multiplied by reward_scale) / undones = !terminal / unmasks = !truncate / values, time-major (H, N, .).
// The arithmetic of a step is instruction-for-instruction that of erl_rollout_step_f32's latency form followed by
// erl_synenv_step_f32's tile form (or erl_pendulum_step_f32), so the six rollout buffers are bit-identical to the per-step
// path under the same Philox keys / injected noise (tests/test_rollout_fused_gpu.py).
#include <type_traits>

#include "erl_common.h"
#include "gae_step.h"
#include "mlp_chain.h"
#include "rollout_bf16.h"

namespace {

constexpr int RF_NSM = 4;      // state k-tiles of 16 held per lane: state_dim <= 64
constexpr int RF_XLD = 68;     // row stride of the state tile XS[16][.]

struct RfArgs {
    const float *Pa, *Pc;                          // actor / critic flat parameter blocks (include/erl_hip.h)
    const float *avg_a, *std_a, *avg_c, *std_c;
    int S, h1, h2, A;
    int64_t N;
    int H;
    const float *noise;                            // (H, N, A) or NULL
    uint64_t seed, counter0;
    float reward_scale;
    float *o_states, *o_actions, *o_logprobs, *o_rewards;
    uint8_t *o_undones, *o_unmasks;
    float *o_values, *o_next_value;                // may be NULL
    // epilogue (round 4; all may be NULL): a private copy of the final state, and get_advantages over the rollout just written --
    // raw advantages, reward sums, the raw sums of the advantage normalisation (erl_gae_scan_f32's `stats` block)
    float *o_last_state;
    float *o_adv, *o_ret;
    double *gae_stats, *gae_ws;                    // ws: [3 x workgroups] fp64 partial sums (erl_adv_stats_fold_f32 folds them into the 5 sums)
    float gamma, lam;
    int vtrace;
    int gae_lds;                                   // 1: the epilogue's inputs are kept in LDS during the rollout (H <= kRfGaeLdsSteps)
    // environment
    float *env_state;                              // (N, S) live state (SynVecEnv.state / PendulumVecEnv.state = obs)
    float *phys;                                   // Pendulum: (N, 2) theta, theta_dot
    const float *Ws, *Wa;                          // SynVecEnv
    int32_t *step_count, *episode;
    int max_step;
    uint64_t env_seed;
    long long *prof;                               // ERL_PROFILE builds only: [wave][16] s_memtime stamps of workgroup 0, step 5
};

#ifdef ERL_PROFILE
long long *g_rf_prof = nullptr;
#define RFPROF(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (g.prof && blockIdx.x == 0 && lane == 0 && t == 5) g.prof[wave * 16 + (i)] = (long long)t_;  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define RFPROF(i) do { } while (0)
#endif

enum { ENV_SYN = 0, ENV_PENDULUM = 1 };

constexpr int RF_WLD = 68;     // row stride of the LDS operand images (W1 rows, Ws^T rows): 16-byte rows, 4 banks apart
// dynamic LDS layout (floats)
constexpr int RF_O_XS = 0;                              // [16][RF_XLD]   state tile
constexpr int RF_O_XA = RF_O_XS + 16 * RF_XLD;          // [3][16][RB_XLD bytes]  state tile normalised for the actor, split (rollout_bf16.h)
constexpr int RF_O_XC = RF_O_XA + RB_XBYTES / 4;        //                ... for the critic
constexpr int RF_O_NRM = RF_O_XC + RB_XBYTES / 4;       // [4][64]        avg_a | den_a | avg_c | den_c  (den = std + 1e-4)
constexpr int RF_O_T1A = RF_O_NRM + 4 * 64;             // [3][16][RB_TLD bytes]  actor H1 tile, split
constexpr int RF_O_T1C = RF_O_T1A + RB_TBYTES / 4;      //                critic H1 tile
constexpr int RF_O_PSA = RF_O_T1C + RB_TBYTES / 4;      // [8][64][4]     actor output-layer partials
constexpr int RF_O_PSC = RF_O_PSA + 8 * 64 * 4;         // [8][16]        critic output-layer partials
constexpr int RF_O_EPS = RF_O_PSC + 8 * 16;             // [2][16][16]    N(0,1) draws of step t (t & 1) and t + 1, produced a step ahead
constexpr int RF_O_RED = RF_O_EPS + 2 * 16 * 16;        // [8][16][2]     env reductions
constexpr int RF_O_BIA = RF_O_RED + 8 * 16 * 2;         // [4][128]       b1 | b2 of the actor, b1 | b2 of the critic (zero beyond h)
constexpr int RF_O_HEAD = RF_O_BIA + 4 * 128;           // [3][16]        action_std_log | b3 of the actor (index clamped to A - 1) | exp(action_std_log)
constexpr int RF_O_WST = RF_O_HEAD + 48;                // [64][RF_WLD]   Ws^T: WST[j][k] = Ws[k][j]
constexpr int RF_O_WAT = RF_O_WST + 64 * RF_WLD;        // [64][16]       Wa^T
constexpr int RF_O_W1C = RF_O_WAT + 64 * 16;            // [128][RF_W1LD bytes]  critic W1, split: [row][3 parts][64 bf16] + 16 bytes (A operands of layer 1)
constexpr int RF_W1LD = 3 * 128 + 16;                   //                rows 100 dwords apart: the 16 rows of a ds_read_b128 lane group on 16 distinct 4-bank groups
constexpr int RF_FLOATS = RF_O_W1C + 128 * RF_W1LD / 4;
constexpr size_t kRfLdsBytes = (size_t)RF_FLOATS * sizeof(float);
// the advantage epilogue keeps its inputs -- [t][reward | value | flags][16 envs] + the bootstrap values -- behind the layout above while
// they fit (192 bytes per step); longer horizons read them back from the rollout buffers
constexpr int kRfGaeLdsSteps = 128;
constexpr size_t rf_gae_lds_bytes(int H) { return ((size_t)H * 3 + 1) * 16 * sizeof(float); }
constexpr size_t kRfRsExtraBytes = 4096;      // role split: one 4 KB tile of W2 small parts behind the layout
static_assert(kRfLdsBytes + kRfRsExtraBytes + rf_gae_lds_bytes(kRfGaeLdsSteps) <= 160 * 1024, "LDS budget of the rollout kernel with the epilogue's tile");
static_assert(128 * RF_W1LD >= 12 * 4096 && RB_TBYTES >= 3 * 4096, "role split: homes of the W2 small-part tiles");

// NS_ / N1_ / N2_: k-tiles of the state / hidden layers as compile-time constants for the tuned shapes (0 = read them from
// the arguments): with them the step body is straight-line code between barriers, which lets the scheduler interleave the
// LDS operand reads, the divisions of the normalisation and the GELUs with the MFMA chains.
// RS_ (round 6, "role split"): waves 0..3 run the ACTOR (two 16-row tiles of every layer each) and step the environment, waves 4..7 run
// the CRITIC on the same state a half step later -- layer 1 while the actor's waves finish the policy head and step the env, layer 2 while
// they write the new state tile, the value sum at the start of the next step -- so that the critic's ~3k cycles per step leave the step's
// dependent chain (tools/r06_gpu_f.sh: the chain without the critic's layers is 10.7k cycles of 13.6k).  Every dot product, GELU and sum
// keeps its operands and its order: the buffers stay bit-identical to the per-step path.  Needs h1 = 128 and h2 in {64, 128} at compile time.
template <int ENV, bool VEC, int NS_, int N1_, int N2_, bool RS_ = false>
__global__ __launch_bounds__(512) void rollout_fused_kernel(RfArgs g)
{
    static_assert(!RS_ || (N1_ == 8 && (N2_ == 4 || N2_ == 8) && NS_ >= 1), "role split: compile-time shapes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *XS = smem + RF_O_XS, *NRM = smem + RF_O_NRM, *PSA = smem + RF_O_PSA;
    u8 *XA = reinterpret_cast<u8 *>(smem + RF_O_XA), *XC = reinterpret_cast<u8 *>(smem + RF_O_XC);
    u8 *T1A = reinterpret_cast<u8 *>(smem + RF_O_T1A), *T1C = reinterpret_cast<u8 *>(smem + RF_O_T1C);
    float *PSC = smem + RF_O_PSC, *EPS = smem + RF_O_EPS, *RED = smem + RF_O_RED;
    float *WST = smem + RF_O_WST, *WAT = smem + RF_O_WAT, *BIA = smem + RF_O_BIA;
    u8 *W1C = reinterpret_cast<u8 *>(smem + RF_O_W1C);
    float *GAE = smem + RF_FLOATS + (RS_ ? 1024 : 0);     // [H][3][16] + [16] (only with g.gae_lds); role split: behind the last W2 small-part tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const Dims da{g.S, g.h1, g.h2, g.A}, dc{g.S, g.h1, g.h2, 1};
    const int S = g.S, A = g.A, ns = NS_ ? NS_ : (S + 15) >> 4, n1 = N1_ ? N1_ : g.h1 >> 4, n2 = N2_ ? N2_ : g.h2 >> 4;
    const int ks_s = NS_ ? (NS_ + 1) >> 1 : (S + 31) >> 5, ks_1 = N1_ ? N1_ >> 1 : g.h1 >> 5;      // k-steps of 32 (rollout_bf16.h)
    const bool on1 = wave < n1, on2 = wave < n2;
    const int64_t env0 = (int64_t)blockIdx.x * 16;
    const int64_t env = env0 + l15;
    const bool valid = env < g.N;
    const int64_t row = valid ? env : g.N - 1;          // rows past N replay env N - 1 (never stored)
    const int H = g.H;
    const size_t N = (size_t)g.N;

    // ---- state tile (raw, and normalised for either network: (s - avg) / (std + 1e-4), AgentPPO.py:360-361, :440-441) and
    // the normalisation constants into LDS; columns >= S stay 0 for the whole rollout.  The normalised tiles are written by
    // whoever produces a state (here, then the env waves): 8 divides per producing lane per step instead of 32 per lane in
    // every one of the 8 waves that consume the tile as an MFMA B operand.
    {
        const int i = tid >> 5, k = 2 * (tid & 31);          // 512 threads x 2 columns = the 16 x 64 tile
        const int64_t r_ = min(env0 + i, g.N - 1);
        float xs[2], xa[2], xc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int kc = min(k + c, S - 1);
            const float x = g.env_state[r_ * S + kc];
            const bool in = k + c < S;
            xs[c] = in ? x : 0.f;
            xa[c] = in ? (x - g.avg_a[kc]) / (g.std_a[kc] + 1e-4f) : 0.f;
            xc[c] = in ? (x - g.avg_c[kc]) / (g.std_c[kc] + 1e-4f) : 0.f;
        }
        XS[i * RF_XLD + k] = xs[0];
        XS[i * RF_XLD + k + 1] = xs[1];
        rb_tile_put2(XA, RB_XLD, i, k, xa[0], xa[1]);
        rb_tile_put2(XC, RB_XLD, i, k, xc[0], xc[1]);
    }
    if (tid < 256) {
        const int which = tid >> 6, k = tid & 63, kc = min(k, S - 1);
        float v;
        if (which == 0) v = g.avg_a[kc];
        else if (which == 1) v = g.std_a[kc] + 1e-4f;
        else if (which == 2) v = g.avg_c[kc];
        else v = g.std_c[kc] + 1e-4f;
        if (k >= S) v = (which & 1) ? 1.f : 0.f;
        NRM[tid] = v;
    }
    if (ENV == ENV_SYN) {
        for (int e = tid; e < 64 * 64; e += 512) {           // WST[j][k] = Ws[k][j]  (coalesced along j)
            const int k = e >> 6, jj = e & 63;
            WST[jj * RF_WLD + k] = (k < S && jj < S) ? g.Ws[(size_t)k * S + jj] : 0.f;
        }
        for (int e = tid; e < 16 * 64; e += 512) {
            const int k = e >> 6, jj = e & 63;
            WAT[jj * 16 + k] = (k < A && jj < S) ? g.Wa[(size_t)k * S + jj] : 0.f;
        }
    }

    // ---- the wave's rows of W2 of both networks, held in registers for the whole rollout
    // (and of W1: rows clamped like the step kernel's loads, columns >= S zero), split into their bf16 parts once
    // (the critic's W1 waits in LDS in its split form: 256 registers hold the three weight blocks below, not four)
    Parts w1a[2], w2a[4], w2c[4];
    constexpr int TP2 = RS_ ? N2_ / 4 : 1;                 // (role split) layer-2 tiles per wave; layer 1: two
    // (role split) this wave's rows of ITS network's W1 / W2, split once: W1 and the two large parts of W2 in registers (112), the small
    // part of W2 in LDS (16 bytes per lane and k-step, lane-contiguous: [tile][k-step][lane]) -- with all three parts of two layer-2 tiles
    // in registers (144) the step loop spilled 33 registers
    Parts W1[2][2];
    u32x4 W2h[TP2][4], W2m[TP2][4];
    const u8 *w2l_at[TP2];
    float4 w3r[TP2];                                       //              ... and its k-slices of the output layer (fp32)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);              // (scalar: role branches are scalar branches)
    const int role = RS_ ? wave_u >> 2 : 0, rw = wave_u & 3;   // (role split) 0: actor, 1: critic; rank among the network's four waves
    const u8 *w1c_at = nullptr;
    float4 w3a = zero4(), w3c = zero4();
    const int kt = min(wave, n2 - 1);                      // this wave's k-tile of the output layers
    if constexpr (RS_) {
        const float *P = role ? g.Pc : g.Pa;
        const Dims &d = role ? dc : da;
        const int outs = role ? 1 : A;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float *r1 = P + d.oW1() + (size_t)(16 * (2 * rw + j) + l15) * S;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                if (ks < ks_s) W1[j][ks] = rb_load_w<VEC>(r1, ks, q, S);
        }
#pragma unroll
        for (int j = 0; j < TP2; ++j) {
            const int tile = TP2 * rw + j;
            const float *r2 = P + d.oW2() + (size_t)(16 * tile + l15) * d.h1;
            // the small parts' home: the actor's eight tiles and the critic's first four where the all-waves mapping keeps the critic's W1,
            // the critic's tiles 4..6 in the second H1 tile (one H1 tile serves both networks here: T1A), tile 7 behind the layout
            u8 *home = role == 0 ? W1C + 4096 * tile : tile < 4 ? W1C + 4096 * (8 + tile) : tile < 7 ? T1C + 4096 * (tile - 4)
                                                                                                     : reinterpret_cast<u8 *>(smem + RF_FLOATS);
            w2l_at[j] = home + 16 * lane;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const Parts p = rb_load_w<VEC>(r2, ks, q, d.h1);
                W2h[j][ks] = p.h;
                W2m[j][ks] = p.m;
                *reinterpret_cast<u32x4 *>(home + 1024 * ks + 16 * lane) = p.l;
            }
            w3r[j] = load4<VEC>(P + d.oW3() + (size_t)min(l15, outs - 1) * d.h2, 16 * tile + 4 * q, d.h2);
            if (l15 >= outs) w3r[j] = zero4();
        }
        // output-layer partials of tiles nobody owns (h2 = 64: tiles 4..7) stay zero for the whole rollout
        for (int e = tid; e < 8 * 64 * 4; e += 512) PSA[e] = 0.f;
        if (tid < 128) PSC[tid] = 0.f;
        if (role == 0) __builtin_amdgcn_s_setprio(2);      // the actor's waves carry the step's dependent chain
    } else {
    {
        const int i = tid >> 2, c = tid & 3;                 // row i, columns 16 c .. 16 c + 15
        const float *rc1 = g.Pc + dc.oW1() + (size_t)min(i, dc.h1 - 1) * S;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const Parts p = rb_split8(load4<VEC>(rc1, 16 * c + 8 * hh, S), load4<VEC>(rc1, 16 * c + 8 * hh + 4, S));
            u8 *dst = W1C + i * RF_W1LD + 32 * c + 16 * hh;
            *reinterpret_cast<u32x4 *>(dst) = p.h;
            *reinterpret_cast<u32x4 *>(dst + 128) = p.m;
            *reinterpret_cast<u32x4 *>(dst + 256) = p.l;
        }
    }
    w1c_at = W1C + (16 * wave + l15) * RF_W1LD + 16 * q;
    {
        const float *ra1 = g.Pa + da.oW1() + (size_t)min(16 * wave + l15, da.h1 - 1) * S;
        const float *ra2 = g.Pa + da.oW2() + (size_t)min(16 * wave + l15, da.h2 - 1) * da.h1;
        const float *rc2 = g.Pc + dc.oW2() + (size_t)min(16 * wave + l15, dc.h2 - 1) * dc.h1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < ks_s) {
                w1a[ks] = rb_load_w<VEC>(ra1, ks, q, S);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < ks_1) {
                w2a[ks] = rb_load_w<VEC>(ra2, ks, q, da.h1);
                w2c[ks] = rb_load_w<VEC>(rc2, ks, q, dc.h1);
            }
        }
    }
    w3a = load4<VEC>(g.Pa + da.oW3() + (size_t)min(l15, A - 1) * da.h2, 16 * kt + 4 * q, da.h2);
    if (l15 >= A || !on2) w3a = zero4();
    w3c = load4<VEC>(g.Pc + dc.oW3(), 16 * kt + 4 * q, dc.h2);
    if (l15 >= 1 || !on2) w3c = zero4();
    }
    {   // the biases wait in LDS (a ds_read_b128 per layer and network per step; 16 registers less across the MFMA chains)
        const int which = tid >> 7, k = tid & 127;
        const float *src = (which & 2 ? g.Pc : g.Pa) + (which & 2 ? (which & 1 ? dc.ob2() : dc.ob1()) : (which & 1 ? da.ob2() : da.ob1()));
        const int hk = (which & 1) ? g.h2 : g.h1;
        const float v = src[min(k, hk - 1)];
        BIA[tid] = k < hk ? v : 0.f;
    }
    const float *b1_at = BIA + 16 * min(wave, n1 - 1) + 4 * q, *b2_at = BIA + 128 + 16 * kt + 4 * q;
    float *HEAD = smem + RF_O_HEAD;
    if (tid < 32) HEAD[tid] = g.Pa[(tid < 16 ? da.oStd() : da.ob3()) + min(tid & 15, A - 1)];
    // the policy's standard deviation is constant over a rollout: expf once per launch instead of four times per step in every env wave
    // (the same function of the same input: the same bits)
    if (tid >= 32 && tid < 48) HEAD[tid] = expf(g.Pa[da.oStd() + min(tid & 15, A - 1)]);
    const float b3c = g.Pc[dc.ob3()];

    // ---- the environment's per-lane constants
    const int nt = (ENV == ENV_SYN) ? ns : 1;                // waves that step the environment (<= 4: state_dim <= 64)
    const float *wst_row = WST + ((16 * wave + l15) & 63) * RF_WLD + 4 * q, *wat_row = WAT + ((16 * wave + l15) & 63) * 16 + 4 * q;
    int sc = g.step_count[row], ep = g.episode[row];
    float th = 0.f, thdot = 0.f;
    if (ENV == ENV_PENDULUM) { th = g.phys[2 * row]; thdot = g.phys[2 * row + 1]; }

    // The N(0,1) draws of step t are produced one step ahead by waves 4..7 (idle while waves < nt step the environment):
    // wave 4 + r, lane (m, q) owns eps[m][4 q + r] -- injected noise[t] or Philox4x32-10 + Box-Muller keyed by
    // (seed, counter0 + t, env, action-dim) -- so the ~600-cycle draw never sits on the step's dependent chain.
    auto draw = [&](int t) {
        if (wave >= 4 && t < H) {
            const int r = wave - 4, ac = min(4 * q + r, A - 1);
            const float e = g.noise ? g.noise[((size_t)t * N + row) * A + ac]
                                    : philox_normal(g.seed, g.counter0 + (uint64_t)t, (uint32_t)row, (uint32_t)ac);
            EPS[(t & 1) * 256 + l15 * 16 + 4 * q + r] = e;
        }
    };
    draw(0);
    __syncthreads();

    // ---- (role split) a network's layers on this wave's tiles.  Operands and order of every sum are those of the all-waves mapping:
    // rb_mma6 over the k-steps in order, rb_sum + bias, GELU, the output layer's partial of ONE 16-feature tile from four fp32 MFMAs;
    // the tiles' partials are summed in tile order by whoever finishes the head (below)
    auto rs_layer1 = [&](const u8 *X, u8 *T1, const float *b1) {
        RbAcc c0, c1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < ks_s) {
                const Parts b = rb_tile_get(X, RB_XLD, l15, ks, q);
                rb_mma6(W1[0][ks], b, c0);
                rb_mma6(W1[1][ks], b, c1);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f0 = 16 * (2 * rw + j) + 4 * q;
            const float4 bv = *reinterpret_cast<const float4 *>(b1 + f0);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            float h[4], gd;
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(j ? c1 : c0, r) + bb[r], h[r], gd);
            rb_tile_put(T1, RB_TLD, l15, f0, h[0], h[1], h[2], h[3]);
        }
    };
    auto rs_layer2 = [&](const u8 *T1, const float *b2, auto critic_c) {
        constexpr bool critic = decltype(critic_c)::value;
        RbAcc c[TP2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const Parts b = rb_tile_get(T1, RB_TLD, l15, ks, q);
#pragma unroll
            for (int j = 0; j < TP2; ++j) {
                Parts a;
                a.h = W2h[j][ks];
                a.m = W2m[j][ks];
                a.l = *reinterpret_cast<const u32x4 *>(w2l_at[j] + 1024 * ks);
                rb_mma6(a, b, c[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < TP2; ++j) {
            const int tile = TP2 * rw + j;
            const float4 bv = *reinterpret_cast<const float4 *>(b2 + 16 * tile + 4 * q);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
            float h[4], gd;
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(c[j], r) + bb[r], h[r], gd);
            f32x4 part = {0.f, 0.f, 0.f, 0.f};
            part = mfma16(w3r[j].x, h[0], part);
            part = mfma16(w3r[j].y, h[1], part);
            part = mfma16(w3r[j].z, h[2], part);
            part = mfma16(w3r[j].w, h[3], part);
            if (critic) { if (q == 0) PSC[tile * 16 + l15] = part[0]; }
            else *reinterpret_cast<float4 *>(PSA + (tile * 64 + lane) * 4) = make_float4(part[0], part[1], part[2], part[3]);
        }
    };
    // the value of the state whose critic partials are in PSC: the tiles' partials in tile order + b3 (wave 7)
    auto value_out = [&](int tv, bool boot) {
        float p[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) p[w] = PSC[w * 16 + l15];
        const float v = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + b3c;
        if (valid && q == 0) {
            if (boot) { if (g.o_next_value) g.o_next_value[row] = v; }
            else if (g.o_values) g.o_values[(size_t)tv * N + row] = v;
        }
        if (g.gae_lds && q == 0) GAE[boot ? H * 48 + l15 : (tv * 3 + 1) * 16 + l15] = v;
    };

    // one step; LAST = the extra pass after the horizon that only evaluates the critic on the final state (bootstrap value).
    // Compile-time so that the H regular steps carry no `last` branches between their MFMA groups.
    auto step = [&](int t, auto last_c) {
        constexpr bool last = decltype(last_c)::value;
        if constexpr (RS_ && last) {
            // (role split) the pass after the horizon: the last step's value, then the critic alone on the final state
            if (wave == 7 && t > 0) value_out(t - 1, false);
            if (role == 1) rs_layer1(XC, T1A, BIA + 256);
            lds_barrier();
            if (role == 1) rs_layer2(T1A, BIA + 256 + 128, std::true_type{});
            lds_barrier();
            if (wave == 7) value_out(0, true);
            return;
        }
        // ================= phase 0: state tile -> registers; states[t]; layer 1 of both networks =================
        RFPROF(0);
        if (wave == 7 && !last && valid) {   // states[t] = state (AgentPPO.py:115)
            float4 R[RF_NSM];
#pragma unroll
            for (int tt = 0; tt < RF_NSM; ++tt)
                R[tt] = (tt < ns) ? *reinterpret_cast<const float4 *>(XS + l15 * RF_XLD + 16 * tt + 4 * q) : zero4();
            float *dst0 = g.o_states + ((size_t)t * N + row) * S;
#pragma unroll
            for (int tt = 0; tt < RF_NSM; ++tt) {
                if (tt < ns) {
                    const int k0 = 16 * tt + 4 * q;
                    if (VEC) { if (k0 < S) *reinterpret_cast<float4 *>(dst0 + k0) = R[tt]; }
                    else {
                        const float xr[4] = {R[tt].x, R[tt].y, R[tt].z, R[tt].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (k0 + c < S) dst0[k0 + c] = xr[c];
                    }
                }
            }
        }
        if constexpr (RS_) {
            // (role split) wave 7 finishes the PREVIOUS step's value from the partials the critic's waves left before barrier (4);
            // the actor's waves run layer 1
            if (wave == 7 && t > 0) value_out(t - 1, false);
            if (role == 0) rs_layer1(XA, T1A, BIA);
        } else
        {
            RbAcc ca, cc;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks < ks_s) {
                    if (!last) rb_mma6(w1a[ks], rb_tile_get(XA, RB_XLD, l15, ks, q), ca);
#ifndef ERL_RF_NO_CRITIC      // (timing experiment: what the critic's layers cost the step's dependent chain; values are garbage without them)
                    Parts w1c;
                    w1c.h = *reinterpret_cast<const u32x4 *>(w1c_at + 64 * ks);
                    w1c.m = *reinterpret_cast<const u32x4 *>(w1c_at + 64 * ks + 128);
                    w1c.l = *reinterpret_cast<const u32x4 *>(w1c_at + 64 * ks + 256);
                    rb_mma6(w1c, rb_tile_get(XC, RB_XLD, l15, ks, q), cc);
#endif
                }
            }
            if (on1) {
                float h[4], gd;
                if (!last) {
                    const float4 b1a = *reinterpret_cast<const float4 *>(b1_at);
                    const float bb[4] = {b1a.x, b1a.y, b1a.z, b1a.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(ca, r) + bb[r], h[r], gd);
                    rb_tile_put(T1A, RB_TLD, l15, 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
                }
#ifndef ERL_RF_NO_CRITIC
                const float4 b1c = *reinterpret_cast<const float4 *>(b1_at + 256);
                const float bc[4] = {b1c.x, b1c.y, b1c.z, b1c.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(cc, r) + bc[r], h[r], gd);
                rb_tile_put(T1C, RB_TLD, l15, 16 * wave + 4 * q, h[0], h[1], h[2], h[3]);
#endif
            }
        }
        RFPROF(1);
        lds_barrier();                                                                               // (1) H1 tiles
        RFPROF(2);

        // ================= phase 1: layer 2 + the output-layer partials of this wave's k-slice =================
        // (a wave whose 16 rows of W2 lie beyond h2 -- waves 4..7 of the Pendulum demo's [128, 64] -- holds zero operands: it skips the
        // chain and leaves its zero partials; the four waves with rows then have the SIMDs' matrix pipes to themselves.  Scalar branch.)
        if constexpr (RS_) {
            // (role split) the actor's layer 2 + output partials; the critic's waves draw the next step's N(0,1) meanwhile
            if (role == 0) rs_layer2(T1A, BIA + 128, std::false_type{});
            else draw(t + 1);
        } else
        if (__builtin_amdgcn_readfirstlane(wave) < n2)
        {
            RbAcc ca, cc;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < ks_1) {
                    if (!last) rb_mma6(w2a[ks], rb_tile_get(T1A, RB_TLD, l15, ks, q), ca);
#ifndef ERL_RF_NO_CRITIC
                    rb_mma6(w2c[ks], rb_tile_get(T1C, RB_TLD, l15, ks, q), cc);
#endif
                }
            }
            float h[4], gd;
            if (!last) {
                const float4 b2a = *reinterpret_cast<const float4 *>(b2_at);
                const float bb[4] = {b2a.x, b2a.y, b2a.z, b2a.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(ca, r) + bb[r], h[r], gd);
                f32x4 part = {0.f, 0.f, 0.f, 0.f};
                part = mfma16(w3a.x, h[0], part);
                part = mfma16(w3a.y, h[1], part);
                part = mfma16(w3a.z, h[2], part);
                part = mfma16(w3a.w, h[3], part);
                *reinterpret_cast<float4 *>(PSA + (wave * 64 + lane) * 4) =
                    on2 ? make_float4(part[0], part[1], part[2], part[3]) : zero4();
            }
            const float4 b2c = *reinterpret_cast<const float4 *>(b2_at + 256);
            const float bc[4] = {b2c.x, b2c.y, b2c.z, b2c.w};
#ifndef ERL_RF_NO_CRITIC
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_and_grad_fast(rb_sum(cc, r) + bc[r], h[r], gd);
#endif
            f32x4 pc = {0.f, 0.f, 0.f, 0.f};
            pc = mfma16(w3c.x, h[0], pc);
            pc = mfma16(w3c.y, h[1], pc);
            pc = mfma16(w3c.z, h[2], pc);
            pc = mfma16(w3c.w, h[3], pc);
            if (q == 0) PSC[wave * 16 + l15] = on2 ? pc[0] : 0.f;
        }
        else {
            if (!last) *reinterpret_cast<float4 *>(PSA + (wave * 64 + lane) * 4) = zero4();
            if (q == 0) PSC[wave * 16 + l15] = 0.f;
        }
        RFPROF(3);
        lds_barrier();                                                                               // (2) partials
        RFPROF(4);

        // ================= phase 2: value (wave 7); policy head + environment step (waves < nt); next draws (waves 4..7) ====
        if constexpr (!RS_) {
            if (wave == 7) value_out(t, last);
            if (last) return;
            draw(t + 1);
        } else {
            // (role split) the critic's layer 1 on the state of THIS step (XC is rewritten after barrier (3)), beside the policy head and the env step
            if (role == 1) rs_layer1(XC, T1A, BIA + 256);
        }
        float out[4] = {0.f, 0.f, 0.f, 0.f}, a2 = 0.f, pend_cost = 0.f;
        const int j0 = 16 * wave + 4 * q;
        if (wave < nt) {
            // the raw state tile: B operand of the env step (read here, not before the layers: 16 registers less across their MFMA chains;
            // XS is not written before barrier (3))
            float4 R[RF_NSM];
            if (ENV == ENV_SYN) {
#pragma unroll
                for (int tt = 0; tt < RF_NSM; ++tt)
                    R[tt] = (tt < ns) ? *reinterpret_cast<const float4 *>(XS + l15 * RF_XLD + 16 * tt + 4 * q) : zero4();
            }
            // every env wave finishes the policy head for its own lanes (same fixed-order sum, same draws: bit-identical in
            // all of them), so tanh(action) reaches the env's MFMA B operand -- k = 4 q + r -- without an LDS round trip
            float Y[4], sl[4], b3a[4], sd[4];
            {
                const float4 s4 = *reinterpret_cast<const float4 *>(HEAD + 4 * q), b4 = *reinterpret_cast<const float4 *>(HEAD + 16 + 4 * q);
                const float4 d4 = *reinterpret_cast<const float4 *>(HEAD + 32 + 4 * q);
                sl[0] = s4.x; sl[1] = s4.y; sl[2] = s4.z; sl[3] = s4.w;
                b3a[0] = b4.x; b3a[1] = b4.y; b3a[2] = b4.z; b3a[3] = b4.w;
                sd[0] = d4.x; sd[1] = d4.y; sd[2] = d4.z; sd[3] = d4.w;
            }
            {
                float4 p[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) p[w] = *reinterpret_cast<const float4 *>(PSA + (w * 64 + lane) * 4);
                Y[0] = ((p[0].x + p[1].x) + (p[2].x + p[3].x)) + ((p[4].x + p[5].x) + (p[6].x + p[7].x)) + b3a[0];
                Y[1] = ((p[0].y + p[1].y) + (p[2].y + p[3].y)) + ((p[4].y + p[5].y) + (p[6].y + p[7].y)) + b3a[1];
                Y[2] = ((p[0].z + p[1].z) + (p[2].z + p[3].z)) + ((p[4].z + p[5].z) + (p[6].z + p[7].z)) + b3a[2];
                Y[3] = ((p[0].w + p[1].w) + (p[2].w + p[3].w)) + ((p[4].w + p[5].w) + (p[6].w + p[7].w)) + b3a[3];
            }
            const float4 e4 = *reinterpret_cast<const float4 *>(EPS + (t & 1) * 256 + l15 * 16 + 4 * q);
            const float eps[4] = {e4.x, e4.y, e4.z, e4.w};
            // a = mean + std * eps (torch.normal(mean, std)); Normal.log_prob summed over the action dims (:373-376)
            float lp = 0.f, te[4], act[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool on = 4 * q + r < A;
                const float sdv = sd[r], var = sdv * sdv;
                act[r] = Y[r] + sdv * eps[r];
                const float diff = act[r] - Y[r];
                const float term = -(diff * diff) / (2.f * var) - sl[r] - kLogSqrt2PiF;
                lp += on ? term : 0.f;
                te[r] = on ? fast_tanh(act[r]) : 0.f;                       // convert_action_for_env (:388-390)
            }
            if (wave == 0) {
                lp += __shfl_xor(lp, 16, 64);
                lp += __shfl_xor(lp, 32, 64);
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * q + r < A) g.o_actions[((size_t)t * N + row) * A + 4 * q + r] = act[r];
                    if (q == 0) g.o_logprobs[(size_t)t * N + row] = lp;
                }
            }
            if (ENV == ENV_SYN) {
                // s' = s Ws + a Wa on the matrix cores: wave w < nt owns features 16 w .. 16 w + 15 (envs.hip synenv_tile_kernel)
                a2 = (te[0] * te[0] + te[1] * te[1]) + (te[2] * te[2] + te[3] * te[3]);
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
                const float4 wb = *reinterpret_cast<const float4 *>(wat_row);
                c0 = mfma16(wb.x, te[0], c0);
                c1 = mfma16(wb.y, te[1], c1);
                c0 = mfma16(wb.z, te[2], c0);
                c1 = mfma16(wb.w, te[3], c1);
#pragma unroll
                for (int tt = 0; tt < RF_NSM; ++tt) {
                    if (tt < ns) {
                        const float4 wa = *reinterpret_cast<const float4 *>(wst_row + 16 * tt);
                        c0 = mfma16(wa.x, R[tt].x, c0);
                        c1 = mfma16(wa.y, R[tt].y, c1);
                        c0 = mfma16(wa.z, R[tt].z, c0);
                        c1 = mfma16(wa.w, R[tt].w, c1);
                    }
                }
                float sq = 0.f, mx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    out[r] = c0[r] + c1[r];
                    if (j0 + r < S) {
                        sq += out[r] * out[r];
                        mx = fmaxf(mx, fabsf(out[r]));
                    }
                }
                {   // three independent cross-lane reductions issued together (each hop is an LDS-latency ds_bpermute)
                    const float a2x = __shfl_xor(a2, 16, 64), sqx = __shfl_xor(sq, 16, 64), mxx = __shfl_xor(mx, 16, 64);
                    a2 += a2x; sq += sqx; mx = fmaxf(mx, mxx);
                    const float a2y = __shfl_xor(a2, 32, 64), sqy = __shfl_xor(sq, 32, 64), mxy = __shfl_xor(mx, 32, 64);
                    a2 += a2y; sq += sqy; mx = fmaxf(mx, mxy);
                }
                if (q == 0) { RED[(wave * 16 + l15) * 2] = sq; RED[(wave * 16 + l15) * 2 + 1] = mx; }
            } else {
                // Pendulum-v1 behind the reference wrapper's scaling (envs.hip pendulum_step_kernel); lanes q > 0 mirror q = 0
                const float PI = 3.14159265358979323846f;
                const float a_env = __shfl(te[0], l15, 64);              // action 0 lives in lane group q = 0
                float u = 2.f * a_env;
                u = fminf(fmaxf(u, -2.f), 2.f);
                const float two_pi = 2.f * PI;
                float ang = fmodf(th + PI, two_pi);
                if (ang < 0.f) ang += two_pi;
                ang -= PI;
                pend_cost = ang * ang + 0.1f * thdot * thdot + 0.001f * u * u;
                float nthdot = thdot + (3.f * 10.f / 2.f * sinf(th) + 3.f * u) * 0.05f;
                nthdot = fminf(fmaxf(nthdot, -8.f), 8.f);
                out[0] = th + nthdot * 0.05f;                              // new theta
                out[1] = nthdot;
            }
        }
        RFPROF(5);
        // (3) the env's cross-wave reductions (SynVecEnv: S features over nt waves).  Pendulum is stepped by wave 0 alone, which goes straight
        // on: nothing it reads below was written by another wave since barrier (2), and what it writes is read after barrier (4)
        if (ENV == ENV_SYN || RS_) lds_barrier();
        RFPROF(6);
        if constexpr (RS_) {
            if (role == 1) rs_layer2(T1A, BIA + 256 + 128, std::true_type{});      // (role split) the critic's layer 2 beside the flags / new state tile
        }
        if (wave < nt) {
            if (ENV == ENV_SYN) {
                float sq = 0.f, mx = 0.f;
                for (int w = 0; w < nt; ++w) { sq += RED[(w * 16 + l15) * 2]; mx = fmaxf(mx, RED[(w * 16 + l15) * 2 + 1]); }
                const int sc1 = sc + 1;
                const bool term = mx > 10.f;
                const bool trunc = (sc1 >= g.max_step) && !term;
                const bool done = term || trunc;
                if (j0 < S) {
                    if (done) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out[r] = philox_normal(g.env_seed, (uint64_t)(ep + 1), (uint32_t)row, (uint32_t)(j0 + r));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (j0 + r >= S) out[r] = 0.f;
                    *reinterpret_cast<float4 *>(XS + l15 * RF_XLD + j0) = make_float4(out[0], out[1], out[2], out[3]);
                    const float4 aa = *reinterpret_cast<const float4 *>(NRM + j0), ad = *reinterpret_cast<const float4 *>(NRM + 64 + j0);
                    const float4 ca = *reinterpret_cast<const float4 *>(NRM + 128 + j0), cd = *reinterpret_cast<const float4 *>(NRM + 192 + j0);
                    rb_tile_put(XA, RB_XLD, l15, j0, (out[0] - aa.x) / ad.x, (out[1] - aa.y) / ad.y, (out[2] - aa.z) / ad.z, (out[3] - aa.w) / ad.w);
                    rb_tile_put(XC, RB_XLD, l15, j0, (out[0] - ca.x) / cd.x, (out[1] - ca.y) / cd.y, (out[2] - ca.z) / cd.z, (out[3] - ca.w) / cd.w);
                }
                if (wave == 0 && q == 0) {
                    const float rew = -(sq / (float)S) - 0.01f * (a2 / (float)A);
                    const float rws = g.reward_scale == 1.0f ? rew : rew * g.reward_scale;                   // rewards *= reward_scale (:126)
                    if (valid) {
                        g.o_rewards[(size_t)t * N + row] = rws;
                        g.o_undones[(size_t)t * N + row] = term ? 0 : 1;                                      // logical_not (:127-128)
                        g.o_unmasks[(size_t)t * N + row] = trunc ? 0 : 1;
                    }
                    if (g.gae_lds) {
                        GAE[(t * 3 + 0) * 16 + l15] = rws;
                        GAE[(t * 3 + 2) * 16 + l15] = __int_as_float((term ? 0 : 1) | (trunc ? 0 : 2));
                    }
                }
                sc = done ? 0 : sc1;
                if (done) ep = ep + 1;
            } else {
                const float PI = 3.14159265358979323846f;
                float nth = out[0], nthdot = out[1];
                const int sc1 = sc + 1;
                const bool trunc = sc1 >= g.max_step;
                if (trunc) {       // reset: theta ~ U(-pi, pi), theta_dot ~ U(-1, 1)
                    ep = ep + 1;
                    const Philox4 p = philox4x32_10((uint32_t)row, 0u, (uint32_t)ep, 0x50454e44u, (uint32_t)g.env_seed,
                                                    (uint32_t)(g.env_seed >> 32));
                    nth = ((float)(p.x >> 8) * (1.f / 16777216.f) * 2.f - 1.f) * PI;
                    nthdot = (float)(p.y >> 8) * (1.f / 16777216.f) * 2.f - 1.f;
                }
                th = nth;
                thdot = nthdot;
                sc = trunc ? 0 : sc1;
                if (q == 0) {
                    const float ob[3] = {cosf(nth), sinf(nth), nthdot};
#pragma unroll
                    for (int k = 0; k < 3; ++k) XS[l15 * RF_XLD + k] = ob[k];
                    rb_tile_put(XA, RB_XLD, l15, 0, (ob[0] - NRM[0]) / NRM[64], (ob[1] - NRM[1]) / NRM[65], (ob[2] - NRM[2]) / NRM[66], 0.f);
                    rb_tile_put(XC, RB_XLD, l15, 0, (ob[0] - NRM[128]) / NRM[192], (ob[1] - NRM[129]) / NRM[193],
                                (ob[2] - NRM[130]) / NRM[194], 0.f);
                    const float rew = -0.5f * pend_cost;
                    const float rws = g.reward_scale == 1.0f ? rew : rew * g.reward_scale;
                    if (valid) {
                        g.o_rewards[(size_t)t * N + row] = rws;
                        g.o_undones[(size_t)t * N + row] = 1;
                        g.o_unmasks[(size_t)t * N + row] = trunc ? 0 : 1;
                    }
                    if (g.gae_lds) {
                        GAE[(t * 3 + 0) * 16 + l15] = rws;
                        GAE[(t * 3 + 2) * 16 + l15] = __int_as_float(1 | (trunc ? 0 : 2));
                    }
                }
            }
        }
        RFPROF(7);
        lds_barrier();                                                                               // (4) new state tile visible
        RFPROF(8);
    };
    for (int t = 0; t < H; ++t) step(t, std::false_type{});
    step(H, std::true_type{});

    // ---- hand the environment back: live state, counters (the per-step kernels keep them in global memory); the agent's own
    // copy of the final state (AgentPPO.py:125 `self.last_state = state`: a tensor of its own, not the env's live buffer)
    for (int e = tid; e < 16 * 64; e += 512) {
        const int i = e >> 6, k = e & 63;
        if (env0 + i < g.N && k < S) {
            const float x = XS[i * RF_XLD + k];
            g.env_state[(env0 + i) * S + k] = x;
            if (g.o_last_state) g.o_last_state[(env0 + i) * S + k] = x;
        }
    }
    if (wave == 0 && q == 0 && valid) {
        g.step_count[row] = sc;
        g.episode[row] = ep;
        if (ENV == ENV_PENDULUM) { g.phys[2 * row] = th; g.phys[2 * row + 1] = thdot; }
    }

    // ---- epilogue: AgentPPO.get_advantages (elegantrl/agents/AgentPPO.py:207-232) + reward_sums (:146) + the sums of the advantage
    // normalisation (:149) for the 16 envs of this workgroup, straight from the rows it has just written (its own stores: visible
    // to the workgroup after the barrier) -- three launches less per iteration than scan + statistics fold + normalisation, which
    // at the benchmark's 32 x 4096 are launch-sized (9.3 + 4.6 + 4.8 us).  The exact scan's arithmetic (gae_step.h): bit-identical to
    // erl_gae_scan_f32(EXACT).  The caller's rewards / undones are NOT touched here (explore_env returns them as the reference
    // does; the truncation fix-up of get_advantages is applied by erl_ppo_finish_f32 at the end of update_net).
    if (g.o_adv) {
        __syncthreads();                                   // (drains this workgroup's stores: vmcnt(0), then the barrier)
        double s_all = 0, s_sub = 0, q_sub = 0;
        if (wave == 0 && q == 0 && valid && g.gae_lds) {
            // inputs from the LDS tile the steps filled (no round trip through memory)
            float nv = GAE[H * 48 + l15], a = 0.f;
            const bool sub_col = (row & 3) == 0;
            for (int t = H - 1; t >= 0; --t) {
                const float r = GAE[(t * 3 + 0) * 16 + l15], v = GAE[(t * 3 + 1) * 16 + l15];
                const int fl = __float_as_int(GAE[(t * 3 + 2) * 16 + l15]);
                const size_t i = (size_t)t * N + row;
                float r_eff;
                uint8_t ud_eff;
                const float out = g.vtrace ? erl_gae_step<true>(r, v, (uint8_t)(fl & 1), (uint8_t)((fl >> 1) & 1), g.gamma, g.lam, nv, a, r_eff, ud_eff)
                                           : erl_gae_step<false>(r, v, (uint8_t)(fl & 1), (uint8_t)((fl >> 1) & 1), g.gamma, g.lam, nv, a, r_eff, ud_eff);
                g.o_adv[i] = out;
                g.o_ret[i] = erl_add_rn(out, v);
                s_all += out;
                if (sub_col && (t & 3) == 0) {
                    s_sub += out;
                    q_sub += (double)out * out;
                }
            }
        } else if (wave == 0 && q == 0 && valid) {
            float nv = g.o_next_value[row], a = 0.f;
            const bool sub_col = (row & 3) == 0;
            constexpr int U = 8;
            for (int tb = H - 1; tb >= 0; tb -= U) {
                float r_[U], v_[U];
                uint8_t ud_[U], um_[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int t = max(tb - j, 0);
                    const size_t i = (size_t)t * N + row;
                    r_[j] = g.o_rewards[i]; v_[j] = g.o_values[i]; ud_[j] = g.o_undones[i]; um_[j] = g.o_unmasks[i];
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int t = tb - j;
                    if (t < 0) break;
                    const size_t i = (size_t)t * N + row;
                    float r_eff;
                    uint8_t ud_eff;
                    const float out = g.vtrace ? erl_gae_step<true>(r_[j], v_[j], ud_[j], um_[j], g.gamma, g.lam, nv, a, r_eff, ud_eff)
                                               : erl_gae_step<false>(r_[j], v_[j], ud_[j], um_[j], g.gamma, g.lam, nv, a, r_eff, ud_eff);
                    g.o_adv[i] = out;
                    g.o_ret[i] = erl_add_rn(out, v_[j]);
                    s_all += out;
                    if (sub_col && (t & 3) == 0) {
                        s_sub += out;
                        q_sub += (double)out * out;
                    }
                }
            }
        }
        // the workgroup's three fp64 partial sums go to gae_ws[3 b ..]; they are folded in index order by the consumer's first
        // launch (erl_adv_stats_fold_f32, or inside the update loop's weight-image kernel).  A fold by the last workgroup to arrive
        // was built first and measured: +22-26 us per rollout -- its arrival counter, its partial loads and its result are three
        // dependent memory round trips AFTER the slowest workgroup has finished, while 40 MB of rollout buffers drain.
        if (wave == 0) {
            const double w0 = wave_sum(s_all), w1 = wave_sum(s_sub), w2 = wave_sum(q_sub);
            if (lane == 0) {
                g.gae_ws[3 * (size_t)blockIdx.x + 0] = w0;
                g.gae_ws[3 * (size_t)blockIdx.x + 1] = w1;
                g.gae_ws[3 * (size_t)blockIdx.x + 2] = w2;
            }
        }
    }
}

bool rf_dims_ok(int S, int h1, int h2, int A)
{
    return mlp_dims_ok(S, h1, h2, A) && S <= 16 * RF_NSM;
}

int rf_launch(RfArgs &g, int env_kind, hipStream_t stream)
{
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (g.S % 4 == 0) && al(g.Pa) && al(g.Pc) && al(g.o_states);
    const dim3 grid((unsigned)erl_cdiv(g.N, 16)), block(512);
    g.gae_lds = (g.o_adv && g.H <= kRfGaeLdsSteps) ? 1 : 0;
    const size_t lds_bytes = kRfLdsBytes + kRfRsExtraBytes + (g.gae_lds ? rf_gae_lds_bytes(g.H) : 0);
    static bool attr[8] = {false, false, false, false, false, false, false, false};
    // role split (the critic off the step's dependent chain; see the kernel): the default for the tuned shapes, ERL_RF_ROLE_SPLIT=0 keeps
    // every wave on both networks (read per launch: A/B in one process)
    const bool rs = [] { const char *e = getenv("ERL_RF_ROLE_SPLIT"); return !e || atoi(e) != 0; }();
#define RF_LAUNCH(E, V, A_, B_, C_, SLOT)                                                                                  \
    do {                                                                                                                   \
        if (!attr[SLOT]) {                                                                                                 \
            int rc = erl_hip_status(hipFuncSetAttribute((const void *)rollout_fused_kernel<E, V, A_, B_, C_, (SLOT >= 6)>,  \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,                         \
                                                        (int)(kRfLdsBytes + kRfRsExtraBytes + rf_gae_lds_bytes(kRfGaeLdsSteps))), \
                                    "hipFuncSetAttribute(rollout_fused_kernel)");                                          \
            if (rc) return rc;                                                                                             \
            attr[SLOT] = true;                                                                                             \
        }                                                                                                                  \
        hipLaunchKernelGGL((rollout_fused_kernel<E, V, A_, B_, C_, (SLOT >= 6)>), grid, block, lds_bytes, stream, g);    \
    } while (0)
    const int ns = (g.S + 15) / 16;
    if (env_kind == ENV_SYN) {
        if (vec && ns == 4 && g.h1 == 128 && g.h2 == 128 && rs) RF_LAUNCH(ENV_SYN, true, 4, 8, 8, 6);  // configs 4 / 5
        else if (vec && ns == 4 && g.h1 == 128 && g.h2 == 128) RF_LAUNCH(ENV_SYN, true, 4, 8, 8, 0);
        else if (vec) RF_LAUNCH(ENV_SYN, true, 0, 0, 0, 1);
        else RF_LAUNCH(ENV_SYN, false, 0, 0, 0, 2);
    } else {
        if (g.h1 == 128 && g.h2 == 64 && rs) RF_LAUNCH(ENV_PENDULUM, false, 1, 8, 4, 7);              // config 2
        else if (g.h1 == 128 && g.h2 == 64) RF_LAUNCH(ENV_PENDULUM, false, 1, 8, 4, 3);
        else RF_LAUNCH(ENV_PENDULUM, false, 0, 0, 0, 4);
    }
#undef RF_LAUNCH
    return erl_hip_status(hipGetLastError(), "rollout_fused_kernel launch");
}

int rf_fill(RfArgs &g, const char *what, const float *actor_params, const float *critic_params, const float *act_avg,
            const float *act_std, const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, int64_t N, int64_t H,
            const float *noise, uint64_t seed, uint64_t counter0, float reward_scale, float *out_states, float *out_actions,
            float *out_logprobs, float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks, float *out_values,
            float *out_next_value, float *out_last_state, float *out_adv, float *out_ret, double *gae_stats, double *gae_ws,
            int64_t gae_ws_bytes, float gamma, float lam, int use_v_trace)
{
    ERL_REQUIRE(actor_params && critic_params && act_avg && act_std && cri_avg && cri_std, "%s: NULL network tensor", what);
    ERL_REQUIRE(out_states && out_actions && out_logprobs && out_rewards && out_undones && out_unmasks, "%s: NULL rollout buffer", what);
    ERL_REQUIRE(rf_dims_ok(S, h1, h2, A), "%s: unsupported dims S=%d net=[%d,%d] A=%d (fused rollout: state_dim <= %d, 2 hidden "
                "layers of 32..128 in steps of 32, action_dim <= 16)", what, S, h1, h2, A, 16 * RF_NSM);
    ERL_REQUIRE(N >= 1 && H >= 1 && H < (1LL << 30), "%s: bad shape N=%lld H=%lld", what, (long long)N, (long long)H);
    g.Pa = actor_params; g.Pc = critic_params;
    g.avg_a = act_avg; g.std_a = act_std; g.avg_c = cri_avg; g.std_c = cri_std;
    g.S = S; g.h1 = h1; g.h2 = h2; g.A = A; g.N = N; g.H = (int)H;
    g.noise = noise; g.seed = seed; g.counter0 = counter0; g.reward_scale = reward_scale;
    g.o_states = out_states; g.o_actions = out_actions; g.o_logprobs = out_logprobs; g.o_rewards = out_rewards;
    g.o_undones = out_undones; g.o_unmasks = out_unmasks; g.o_values = out_values; g.o_next_value = out_next_value;
    g.o_last_state = out_last_state;
    if (out_adv || out_ret) {
        ERL_REQUIRE(out_adv && out_ret && gae_ws && out_values && out_next_value,
                    "%s: the advantage epilogue needs out_advantages, out_reward_sums, gae_workspace, out_values and out_next_value", what);
        ERL_REQUIRE(gae_ws_bytes >= erl_rollout_gae_workspace_bytes(N), "%s: gae_workspace of %lld bytes, erl_rollout_gae_workspace_bytes(N) = %lld",
                    what, (long long)gae_ws_bytes, (long long)erl_rollout_gae_workspace_bytes(N));
    }
    g.o_adv = out_adv; g.o_ret = out_ret; g.gae_stats = gae_stats; g.gae_ws = gae_ws;
    g.gamma = gamma; g.lam = lam; g.vtrace = use_v_trace ? 1 : 0;
#ifdef ERL_PROFILE
    g.prof = g_rf_prof;
#endif
    return ERL_OK;
}

}  // namespace

#ifdef ERL_PROFILE
// profiling builds only (make EXTRA=-DERL_PROFILE): device buffer of 8 * 16 int64 cycle stamps
extern "C" __attribute__((visibility("default"))) void erl_debug_set_rollout_fused_profile(long long *dev_buf) { g_rf_prof = dev_buf; }
#endif

extern "C" int erl_rollout_fused_supported(int S, int h1, int h2, int A) { return rf_dims_ok(S, h1, h2, A) ? 1 : 0; }

// bytes of `gae_workspace` of the persistent rollouts for N envs: 3 fp64 partial sums per 16-env workgroup (erl_rollout_gae_partials of them)
extern "C" int64_t erl_rollout_gae_workspace_bytes(int64_t N) { return N >= 1 ? 3 * erl_cdiv(N, 16) * 8 : -1; }
extern "C" int erl_rollout_gae_partials(int64_t N) { return N >= 1 ? (int)erl_cdiv(N, 16) : -1; }

extern "C" int erl_rollout_synenv_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                                      const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, float *env_state,
                                      const float *Ws, const float *Wa, int32_t *step_count, int32_t *episode, int max_step,
                                      uint64_t env_seed, int64_t N, int64_t H, const float *noise, uint64_t seed, uint64_t counter0,
                                      float reward_scale, float *out_states, float *out_actions, float *out_logprobs,
                                      float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks, float *out_values,
                                      float *out_next_value, float *out_last_state, float *out_advantages, float *out_reward_sums,
                                      double *gae_stats, double *gae_workspace, int64_t gae_workspace_bytes, float gamma,
                                      float lambda_gae, int use_v_trace, void *stream)
{
    RfArgs g{};
    int rc = rf_fill(g, "erl_rollout_synenv_f32", actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, S, h1, h2, A, N, H,
                     noise, seed, counter0, reward_scale, out_states, out_actions, out_logprobs, out_rewards, out_undones, out_unmasks,
                     out_values, out_next_value, out_last_state, out_advantages, out_reward_sums, gae_stats, gae_workspace,
                     gae_workspace_bytes, gamma, lambda_gae, use_v_trace);
    if (rc) return rc;
    ERL_REQUIRE(env_state && Ws && Wa && step_count && episode && max_step >= 1, "erl_rollout_synenv_f32: bad environment argument");
    g.env_state = env_state; g.Ws = Ws; g.Wa = Wa; g.step_count = step_count; g.episode = episode;
    g.max_step = max_step; g.env_seed = env_seed;
    return rf_launch(g, ENV_SYN, (hipStream_t)stream);
}

extern "C" int erl_rollout_pendulum_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                                        const float *act_std, const float *cri_avg, const float *cri_std, int h1, int h2, float *phys,
                                        float *obs, int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N,
                                        int64_t H, const float *noise, uint64_t seed, uint64_t counter0, float reward_scale,
                                        float *out_states, float *out_actions, float *out_logprobs, float *out_rewards,
                                        uint8_t *out_undones, uint8_t *out_unmasks, float *out_values, float *out_next_value,
                                        float *out_last_state, float *out_advantages, float *out_reward_sums, double *gae_stats,
                                        double *gae_workspace, int64_t gae_workspace_bytes, float gamma, float lambda_gae,
                                        int use_v_trace, void *stream)
{
    RfArgs g{};
    int rc = rf_fill(g, "erl_rollout_pendulum_f32", actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, 3, h1, h2, 1, N, H,
                     noise, seed, counter0, reward_scale, out_states, out_actions, out_logprobs, out_rewards, out_undones, out_unmasks,
                     out_values, out_next_value, out_last_state, out_advantages, out_reward_sums, gae_stats, gae_workspace,
                     gae_workspace_bytes, gamma, lambda_gae, use_v_trace);
    if (rc) return rc;
    ERL_REQUIRE(phys && obs && step_count && episode && max_step >= 1, "erl_rollout_pendulum_f32: bad environment argument");
    g.env_state = obs; g.phys = phys; g.step_count = step_count; g.episode = episode;
    g.max_step = max_step; g.env_seed = env_seed;
    return rf_launch(g, ENV_PENDULUM, (hipStream_t)stream);
}
