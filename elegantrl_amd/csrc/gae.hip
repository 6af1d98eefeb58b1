// K3: GAE / lambda-return backward scan, K4: advantage statistics + normalisation.  gfx950.
//
// Reference semantics: elegantrl/agents/AgentPPO.py:207-232 (get_advantages), :146 (reward_sums),
// :149 (normalisation).  Data is time-major (H, N): x[t] is one contiguous N-vector, so a
// lane-per-env access is coalesced.  The recurrence is affine in the carried advantage,
//     adv_t = delta_t + c_t * adv_{t+1},   delta_t = (r_t + m_t v_{t+1}) - v_t,   c_t = m_t * lambda,
// which is what the time-parallel variants exploit (only N independent chains exist, far too few
// to fill 256 CUs on their own).
//
//  * EXACT   : one lane per env, reference op order, separate mul/add roundings (no FMA contraction)
//              -> bit-identical to oracle/gae_scan.c.  Right choice when H*N is launch-latency sized.
//  * CHUNKED : grid = env-groups x time-chunks.  Pass A reduces every chunk to its affine map
//              (A_k, P_k); pass B folds the later chunks' maps into the carry, re-scans its chunk and
//              writes adv/ret (+ statistics partials).  Reassociation error <= ~1e-6 at default gamma.
//  * LOOKBACK: single pass (18 B/elem of HBM traffic), gae_lookback.hip: chunk data stays in registers while
//              the slab's affine map is published as a nonce-tagged 8-byte granule and later slabs' granules
//              are walked (decoupled look-back; slabs are handed out by an atomic ticket, latest time first,
//              so a slab only waits for workgroups that are already running).  AUTO picks it for H >= 64.
#include "erl_common.h"
#include "gae_step.h"

#pragma clang fp contract(off)

namespace {

struct Step {
    float r, v;
    uint8_t ud, um;
};

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

// ----------------------------------------------------------------------------------------------
// EXACT: sequential in t, one lane per env.  Loads for U steps are issued together (they do not
// depend on the recurrence), the recurrence itself is 5 dependent flops per step.
// ----------------------------------------------------------------------------------------------
template <bool VTRACE, bool STATS>
__global__ __launch_bounds__(64) void gae_exact_kernel(float *__restrict__ rewards, uint8_t *__restrict__ undones,
                                                       const uint8_t *__restrict__ unmasks,
                                                       const float *__restrict__ values,
                                                       const float *__restrict__ next_value, float *__restrict__ adv,
                                                       float *__restrict__ ret, int H, int N, float gamma, float lam,
                                                       int mutate, double *__restrict__ partials, unsigned long long *span)
{
    const unsigned long long t_span = erl_span_in(span);
    const int n = blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    float nv = live ? next_value[n] : 0.f;
    float a = 0.f;
    double s_all = 0, s_sub = 0, q_sub = 0;
    const bool sub_col = (n & 3) == 0;
    constexpr int U = 8;
    if (live) {
        for (int tb = H - 1; tb >= 0; tb -= U) {
            Step st[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int t = tb - j;
                if (t >= 0) {
                    const size_t i = (size_t)t * N + n;
                    st[j].r = rewards[i];
                    st[j].v = values[i];
                    st[j].ud = undones[i];
                    st[j].um = unmasks[i];
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int t = tb - j;
                if (t < 0) break;
                const size_t i = (size_t)t * N + n;
                const float v = st[j].v;
                float r_eff;
                uint8_t ud_eff;
                const float out = erl_gae_step<VTRACE>(st[j].r, v, st[j].ud, st[j].um, gamma, lam, nv, a, r_eff, ud_eff);
                if (!st[j].um && mutate) {  // truncated: the bootstrapped reward and the cut flag go back to the caller (:211-214)
                    rewards[i] = r_eff;
                    undones[i] = 0;
                }
                adv[i] = out;
                if (ret) ret[i] = add_rn(out, v);
                if (STATS) {
                    s_all += out;
                    if (sub_col && (t & 3) == 0) {
                        s_sub += out;
                        q_sub += (double)out * out;
                    }
                }
            }
        }
    }
    if (STATS) {
        const double w0 = wave_sum(s_all), w1 = wave_sum(s_sub), w2 = wave_sum(q_sub);
        if (threadIdx.x == 0) {
            partials[(size_t)blockIdx.x * 3 + 0] = w0;
            partials[(size_t)blockIdx.x * 3 + 1] = w1;
            partials[(size_t)blockIdx.x * 3 + 2] = w2;
        }
    }
    erl_span_out(span, t_span);
}

// ----------------------------------------------------------------------------------------------
// CHUNKED two-pass.  Thread = VEC consecutive envs x one time chunk of length L.
// ----------------------------------------------------------------------------------------------
template <int VEC>
struct Row {
    float r[VEC], v[VEC];
    uint8_t ud[VEC], um[VEC];
};

template <int VEC>
__device__ __forceinline__ void load_row(Row<VEC> &x, const float *rewards, const uint8_t *undones,
                                         const uint8_t *unmasks, const float *values, size_t i)
{
    if (VEC == 4) {
        const float4 r4 = *reinterpret_cast<const float4 *>(rewards + i);
        const float4 v4 = *reinterpret_cast<const float4 *>(values + i);
        const uchar4 u4 = *reinterpret_cast<const uchar4 *>(undones + i);
        const uchar4 m4 = *reinterpret_cast<const uchar4 *>(unmasks + i);
        x.r[0] = r4.x; x.r[1 % VEC] = r4.y; x.r[2 % VEC] = r4.z; x.r[3 % VEC] = r4.w;
        x.v[0] = v4.x; x.v[1 % VEC] = v4.y; x.v[2 % VEC] = v4.z; x.v[3 % VEC] = v4.w;
        x.ud[0] = u4.x; x.ud[1 % VEC] = u4.y; x.ud[2 % VEC] = u4.z; x.ud[3 % VEC] = u4.w;
        x.um[0] = m4.x; x.um[1 % VEC] = m4.y; x.um[2 % VEC] = m4.z; x.um[3 % VEC] = m4.w;
    } else {
        x.r[0] = rewards[i];
        x.v[0] = values[i];
        x.ud[0] = undones[i];
        x.um[0] = unmasks[i];
    }
}

// one backward step of the affine recurrence for lane-slot e; returns adv_t
__device__ __forceinline__ float gae_step(float r, float v, uint8_t ud, uint8_t um, float gamma, float lam, float &vnext,
                                          float &a, float &prod, float &r_fixed, uint8_t &ud_fixed)
{
    if (!um) {
        r = r + v;
        ud = 0;
    }
    r_fixed = r;
    ud_fixed = ud;
    const float m = ud ? gamma : 0.f;
    const float c = m * lam;
    const float delta = (r + m * vnext) - v;
    a = delta + c * a;
    prod = c * prod;
    vnext = v;
    return a;
}

template <int VEC>
__global__ __launch_bounds__(256) void gae_chunk_aggregate_kernel(const float *__restrict__ rewards,
                                                                  const uint8_t *__restrict__ undones,
                                                                  const uint8_t *__restrict__ unmasks,
                                                                  const float *__restrict__ values,
                                                                  const float *__restrict__ next_value,
                                                                  float2 *__restrict__ agg, int H, int N, int L,
                                                                  float gamma, float lam, int vtrace)
{
    const int n0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (n0 >= N) return;
    const int k = blockIdx.y;
    const int t0 = k * L, t1 = min(H, t0 + L);
    float vnext[VEC], a[VEC], prod[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        vnext[e] = (t1 == H) ? (vtrace ? next_value[n0 + e] : 0.f) : values[(size_t)t1 * N + n0 + e];
        a[e] = 0.f;
        prod[e] = 1.f;
    }
#pragma unroll 8
    for (int t = t1 - 1; t >= t0; --t) {
        Row<VEC> x;
        load_row<VEC>(x, rewards, undones, unmasks, values, (size_t)t * N + n0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float rf;
            uint8_t uf;
            gae_step(x.r[e], x.v[e], x.ud[e], x.um[e], gamma, lam, vnext[e], a[e], prod[e], rf, uf);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) agg[(size_t)k * N + n0 + e] = make_float2(a[e], prod[e]);
}

template <int VEC, bool STATS>
__global__ __launch_bounds__(256) void gae_chunk_finalize_kernel(float *__restrict__ rewards,
                                                                 uint8_t *__restrict__ undones,
                                                                 const uint8_t *__restrict__ unmasks,
                                                                 const float *__restrict__ values,
                                                                 const float *__restrict__ next_value,
                                                                 const float2 *__restrict__ agg, float *__restrict__ adv,
                                                                 float *__restrict__ ret, int H, int N, int L, int K,
                                                                 float gamma, float lam, int vtrace, int mutate,
                                                                 double *__restrict__ partials)
{
    __shared__ double scratch[3][4];
    const int n0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const bool live = n0 < N;
    const int k = blockIdx.y;
    const int t0 = k * L, t1 = min(H, t0 + L);
    double s_all = 0, s_sub = 0, q_sub = 0;
    if (live) {
        float vnext[VEC], a[VEC], prod[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            vnext[e] = (t1 == H) ? (vtrace ? next_value[n0 + e] : 0.f) : values[(size_t)t1 * N + n0 + e];
            a[e] = 0.f;  // carry = adv at the first step of chunk k+1, folded from the far end
            prod[e] = 1.f;
        }
#pragma unroll 4
        for (int j = K - 1; j > k; --j) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float2 g = agg[(size_t)j * N + n0 + e];
                a[e] = g.x + g.y * a[e];
            }
        }
#pragma unroll 8
        for (int t = t1 - 1; t >= t0; --t) {
            const size_t i = (size_t)t * N + n0;
            Row<VEC> x;
            load_row<VEC>(x, rewards, undones, unmasks, values, i);
            float o[VEC], rt[VEC];
            bool any_fix = false;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float rf;
                uint8_t uf;
                o[e] = gae_step(x.r[e], x.v[e], x.ud[e], x.um[e], gamma, lam, vnext[e], a[e], prod[e], rf, uf);
                rt[e] = o[e] + x.v[e];
                if (!x.um[e]) {
                    any_fix = true;
                    x.r[e] = rf;
                    x.ud[e] = uf;
                }
                if (STATS) {
                    s_all += o[e];
                    if (((n0 + e) & 3) == 0 && (t & 3) == 0) {
                        s_sub += o[e];
                        q_sub += (double)o[e] * o[e];
                    }
                }
            }
            if (VEC == 4) {
                *reinterpret_cast<float4 *>(adv + i) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
                if (ret) *reinterpret_cast<float4 *>(ret + i) = make_float4(rt[0], rt[1 % VEC], rt[2 % VEC], rt[3 % VEC]);
            } else {
                adv[i] = o[0];
                if (ret) ret[i] = rt[0];
            }
            if (mutate && any_fix) {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (!x.um[e]) {
                        rewards[i + e] = x.r[e];
                        undones[i + e] = 0;
                    }
            }
        }
    }
    if (STATS) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const double w0 = wave_sum(s_all), w1 = wave_sum(s_sub), w2 = wave_sum(q_sub);
        if (lane == 0) {
            scratch[0][wave] = w0;
            scratch[1][wave] = w1;
            scratch[2][wave] = w2;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            partials[b * 3 + threadIdx.x] =
                scratch[threadIdx.x][0] + scratch[threadIdx.x][1] + scratch[threadIdx.x][2] + scratch[threadIdx.x][3];
        }
    }
}

// ----------------------------------------------------------------------------------------------
// statistics: standalone partials kernel (for callers that did not fuse them), and the 1-block fold.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adv_stats_partial_kernel(const float *__restrict__ adv, int H, int N,
                                                                double *__restrict__ partials)
{
    __shared__ double scratch[4];
    double s_all = 0, s_sub = 0, q_sub = 0;
    const size_t total = (size_t)H * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float x = adv[i];
        const int t = (int)(i / N), n = (int)(i - (size_t)t * N);
        s_all += x;
        if ((t & 3) == 0 && (n & 3) == 0) {
            s_sub += x;
            q_sub += (double)x * x;
        }
    }
    const double b0 = block_sum(s_all, scratch), b1 = block_sum(s_sub, scratch), b2 = block_sum(q_sub, scratch);
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * 3 + 0] = b0;
        partials[(size_t)blockIdx.x * 3 + 1] = b1;
        partials[(size_t)blockIdx.x * 3 + 2] = b2;
    }
}

__global__ __launch_bounds__(256) void adv_stats_fold_kernel(const double *__restrict__ partials, int nparts, int H,
                                                             int N, double *__restrict__ stats)
{
    __shared__ double scratch[4];
    erl_adv_stats_fold_block(partials, nparts, H, N, stats, scratch);
}

__global__ __launch_bounds__(256) void adv_normalize_kernel(const float *__restrict__ adv, float *__restrict__ out,
                                                            size_t total, const double *__restrict__ stats)
{
    const double mean_d = stats[0] / stats[1];
    const double cnt = stats[4];
    double var = (stats[3] - stats[2] * stats[2] / cnt) / (cnt - 1.0);  // unbiased (torch.std default)
    var = var > 0 ? var : 0;
    const float mean = (float)mean_d;
    const float denom = __fadd_rn((float)sqrt(var), 1e-5f);
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += stride) {
        if (i + 3 < total && ((reinterpret_cast<uintptr_t>(adv + i) | reinterpret_cast<uintptr_t>(out + i)) & 15) == 0) {
            float4 x = *reinterpret_cast<const float4 *>(adv + i);
            x.x = __fdiv_rn(__fsub_rn(x.x, mean), denom);
            x.y = __fdiv_rn(__fsub_rn(x.y, mean), denom);
            x.z = __fdiv_rn(__fsub_rn(x.z, mean), denom);
            x.w = __fdiv_rn(__fsub_rn(x.w, mean), denom);
            *reinterpret_cast<float4 *>(out + i) = x;
        } else {
            for (size_t j = i; j < total && j < i + 4; ++j) out[j] = __fdiv_rn(__fsub_rn(adv[j], mean), denom);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// n-step discounted return of the off-policy agents (AgentBase.get_cumulative_rewards, AgentBase.py:226-237):
//   masks = undones * gamma;  for t = H-1 .. 0:  cum[t] = next_value = rewards[t] + masks[t] * next_value
// One lane per sequence, the reference's op order with every product / sum rounded separately (this file is built
// with fp contraction off): bit-identical to the chain of ATen ops.  It is K3's recurrence with lambda = 1 and no
// value term; U steps of loads are issued together ahead of the dependent chain.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cum_rewards_kernel(const float *__restrict__ rewards, const float *__restrict__ undones,
                                                         const float *__restrict__ next_value, float *__restrict__ out, int H,
                                                         int N, float gamma)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float nv = next_value[n];
    constexpr int U = 8;
    for (int tb = H - 1; tb >= 0; tb -= U) {
        float r[U], u[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = tb - j;
            if (t >= 0) {
                r[j] = rewards[(size_t)t * N + n];
                u[j] = undones[(size_t)t * N + n];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = tb - j;
            if (t < 0) break;
            nv = add_rn(r[j], mul_rn(mul_rn(u[j], gamma), nv));
            out[(size_t)t * N + n] = nv;
        }
    }
}

inline int pick_chunk_len(int64_t H, int64_t N, int vec)
{
    // Pass B folds the (K - k - 1) later chunk maps per lane: that term is O(K^2 N), the chunk re-scan
    // O(H N).  K ~ L ~ sqrt(H) keeps the fold at a fraction of the streaming work while still giving
    // sqrt(H) * N / vec threads (4096 envs x 2048 steps -> 43 chunks of 48 steps).
    (void)N; (void)vec;
    int64_t L = 8;
    while (L * L < H) L += 4;
    if (L > H) L = H;
    return (int)L;
}

}  // namespace

// gae_lookback.hip
bool erl_gae_lookback_usable(const float *rewards, const uint8_t *undones, const uint8_t *unmasks, const float *values,
                             const float *next_value, const float *adv, const float *ret, int64_t N);
int erl_gae_lookback_launch(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                            const float *next_value, float *adv, float *ret, int64_t H, int64_t N, float gamma, float lam,
                            bool vtrace, bool mutate, bool want_stats, void *workspace, int64_t workspace_bytes,
                            double **partials, int *nparts, hipStream_t stream);

extern "C" int64_t erl_gae_workspace_bytes(int64_t H, int64_t N)
{
    if (H < 0 || N < 0) return 0;
    return H * N * 2 + (int64_t)(1 << 20);
}

extern "C" int erl_gae_scan_f32(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                                const float *next_value, float *adv, float *ret, int64_t H, int64_t N, float gamma,
                                float lam, int flags, double *stats, void *workspace, int64_t workspace_bytes,
                                void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    ERL_REQUIRE(rewards && undones && unmasks && values && next_value && adv, "erl_gae_scan_f32: NULL tensor");
    ERL_REQUIRE(H >= 1 && N >= 1 && H * N < (1LL << 40) && N < (1LL << 31) && H < (1LL << 31),
                "erl_gae_scan_f32: bad shape H=%lld N=%lld", (long long)H, (long long)N);
    const bool vtrace = flags & ERL_GAE_VTRACE, mutate = flags & ERL_GAE_MUTATE, want_stats = flags & ERL_GAE_STATS;
    ERL_REQUIRE(!want_stats || stats, "erl_gae_scan_f32: ERL_GAE_STATS needs stats");
    ERL_REQUIRE(workspace && workspace_bytes >= erl_gae_workspace_bytes(H, N), "erl_gae_scan_f32: workspace too small");
    int algo = flags & ERL_GAE_ALGO_MASK;
    const bool lb_ok = H >= 4 && erl_gae_lookback_usable(rewards, undones, unmasks, values, next_value, adv, ret, N);
    // AUTO: short horizons are launch-latency sized -> the bit-exact lane-per-env scan (N <= 8192: 64-128 workgroups walking 32 dependent
    // steps take 5.6 us at 32 x 4096, the one-slab form of the single-pass scan 4.8: not worth giving up bit-exactness; at 32 x 32768 it
    // is 11.9 against 5.5 us, tools/gae_lb_sweep.py); otherwise the single-pass scan
    if (algo == ERL_GAE_ALGO_AUTO)
        algo = (H < 64 && (N <= 8192 || !lb_ok)) ? ERL_GAE_ALGO_EXACT : (lb_ok ? ERL_GAE_ALGO_LOOKBACK : ERL_GAE_ALGO_CHUNKED);
    if (algo == ERL_GAE_ALGO_LOOKBACK && !lb_ok) algo = ERL_GAE_ALGO_CHUNKED;  // needs N % 4 == 0 and 16-byte aligned rows

    // workspace: [agg float2 K*N][partials double 3*nblk]
    char *ws = (char *)workspace;
    int nparts = 0;
    double *partials = nullptr;
    if (algo == ERL_GAE_ALGO_LOOKBACK) {
        int rc = erl_gae_lookback_launch(rewards, undones, unmasks, values, next_value, adv, ret, H, N, gamma, lam, vtrace,
                                         mutate, want_stats, workspace, workspace_bytes, &partials, &nparts, stream);
        if (rc) return rc;
    } else if (algo == ERL_GAE_ALGO_EXACT) {
        const int nblk = (int)erl_cdiv(N, 64);
        partials = (double *)ws;
        nparts = nblk;
#define LAUNCH_EXACT(VT, ST)                                                                                       \
    hipLaunchKernelGGL((gae_exact_kernel<VT, ST>), dim3(nblk), dim3(64), 0, stream, rewards, undones, unmasks, values, \
                       next_value, adv, ret, (int)H, (int)N, gamma, lam, (int)mutate, partials, erl_span_slot(ERL_SPAN_GAE, nblk))
        if (vtrace) { if (want_stats) LAUNCH_EXACT(true, true); else LAUNCH_EXACT(true, false); }
        else        { if (want_stats) LAUNCH_EXACT(false, true); else LAUNCH_EXACT(false, false); }
#undef LAUNCH_EXACT
    } else {
        const bool vec4 = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(rewards) | reinterpret_cast<uintptr_t>(values) |
                                            reinterpret_cast<uintptr_t>(adv) | reinterpret_cast<uintptr_t>(ret)) % 16 == 0) &&
                          ((reinterpret_cast<uintptr_t>(undones) | reinterpret_cast<uintptr_t>(unmasks)) % 4 == 0);
        const int vec = vec4 ? 4 : 1;
        const int L = pick_chunk_len(H, N, vec);
        const int K = (int)erl_cdiv(H, L);
        const int gx = (int)erl_cdiv(N, 256 * vec);
        float2 *agg = (float2 *)ws;
        const size_t agg_bytes = ((size_t)K * N * sizeof(float2) + 255) & ~(size_t)255;
        ERL_REQUIRE((int64_t)(agg_bytes + (size_t)gx * K * 24) <= workspace_bytes, "erl_gae_scan_f32: workspace layout");
        partials = (double *)(ws + agg_bytes);
        nparts = gx * K;
        const dim3 grid(gx, K), block(256);
        if (K > 1) {
            if (vec4)
                hipLaunchKernelGGL((gae_chunk_aggregate_kernel<4>), grid, block, 0, stream, rewards, undones, unmasks, values,
                                   next_value, agg, (int)H, (int)N, L, gamma, lam, (int)vtrace);
            else
                hipLaunchKernelGGL((gae_chunk_aggregate_kernel<1>), grid, block, 0, stream, rewards, undones, unmasks, values,
                                   next_value, agg, (int)H, (int)N, L, gamma, lam, (int)vtrace);
        }
#define LAUNCH_FIN(V, ST)                                                                                              \
    hipLaunchKernelGGL((gae_chunk_finalize_kernel<V, ST>), grid, block, 0, stream, rewards, undones, unmasks, values,   \
                       next_value, agg, adv, ret, (int)H, (int)N, L, K, gamma, lam, (int)vtrace, (int)mutate, partials)
        if (vec4) { if (want_stats) LAUNCH_FIN(4, true); else LAUNCH_FIN(4, false); }
        else      { if (want_stats) LAUNCH_FIN(1, true); else LAUNCH_FIN(1, false); }
#undef LAUNCH_FIN
    }
    if (want_stats)
        hipLaunchKernelGGL(adv_stats_fold_kernel, dim3(1), dim3(256), 0, stream, partials, nparts, (int)H, (int)N, stats);
    ERL_LAUNCH_CHECK("erl_gae_scan_f32");
}

extern "C" int erl_cum_rewards_f32(const float *rewards, const float *undones, const float *next_value, float *cum_rewards,
                                   int64_t H, int64_t N, float gamma, void *stream)
{
    ERL_REQUIRE(rewards && undones && next_value && cum_rewards, "erl_cum_rewards_f32: NULL tensor");
    ERL_REQUIRE(H >= 1 && N >= 1 && N < (1LL << 31) && H < (1LL << 31), "erl_cum_rewards_f32: bad shape H=%lld N=%lld",
                (long long)H, (long long)N);
    hipLaunchKernelGGL(cum_rewards_kernel, dim3((unsigned)erl_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, rewards, undones,
                       next_value, cum_rewards, (int)H, (int)N, gamma);
    ERL_LAUNCH_CHECK("erl_cum_rewards_f32");
}

extern "C" int erl_adv_stats_f32(const float *adv, int64_t H, int64_t N, double *stats, void *workspace,
                                 int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    ERL_REQUIRE(adv && stats && workspace, "erl_adv_stats_f32: NULL tensor");
    ERL_REQUIRE(H >= 1 && N >= 1 && N < (1LL << 31) && H < (1LL << 31), "erl_adv_stats_f32: bad shape");
    int nblk = (int)erl_cdiv(H * N, 256 * 8);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    ERL_REQUIRE(workspace_bytes >= (int64_t)nblk * 24, "erl_adv_stats_f32: workspace too small");
    hipLaunchKernelGGL(adv_stats_partial_kernel, dim3(nblk), dim3(256), 0, stream, adv, (int)H, (int)N, (double *)workspace);
    hipLaunchKernelGGL(adv_stats_fold_kernel, dim3(1), dim3(256), 0, stream, (const double *)workspace, nblk, (int)H, (int)N,
                       stats);
    ERL_LAUNCH_CHECK("erl_adv_stats_f32");
}

extern "C" int erl_adv_stats_fold_f32(const double *partials, int n_partials, int64_t H, int64_t N, double *stats, void *stream)
{
    ERL_REQUIRE(partials && stats && n_partials >= 1 && H >= 1 && N >= 1 && N < (1LL << 31) && H < (1LL << 31), "erl_adv_stats_fold_f32: bad argument");
    hipLaunchKernelGGL(adv_stats_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n_partials, (int)H, (int)N, stats);
    ERL_LAUNCH_CHECK("erl_adv_stats_fold_f32");
}

extern "C" int erl_adv_normalize_f32(const float *adv, float *out, int64_t H, int64_t N, const double *stats, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    ERL_REQUIRE(adv && out && stats, "erl_adv_normalize_f32: NULL tensor");
    ERL_REQUIRE(H >= 1 && N >= 1, "erl_adv_normalize_f32: bad shape");
    const size_t total = (size_t)H * (size_t)N;
    int nblk = (int)erl_cdiv((int64_t)total, 256 * 4);
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3(nblk), dim3(256), 0, stream, adv, out, total, stats);
    ERL_LAUNCH_CHECK("erl_adv_normalize_f32");
}
