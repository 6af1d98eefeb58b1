// Shared pieces of the layered paths (own fp32 MFMA GEMMs with fused epilogues, gemm_tiles.h): erl_mlpn_* (mlpn.hip) and
// erl_sac_* (sac.hip).
#pragma once

#include "gemm_tiles.h"
#include "ppo_objective.h"

namespace {

constexpr int MAXL = ERL_MAX_LAYERS;      // hidden layers
constexpr float kLogSqrt2PiN = 0.91893853320467274178f;

struct NetDims {
    int n;                 // number of dense layers = hidden + 1
    int n_act;             // layers 0 .. n_act - 1 are followed by GELU (default n - 1: every layer but the last; ActorFixSAC's encoder
                           // -- build_mlp([S, *net_dims]) with a RAW last layer, elegantrl/agents/AgentSAC.py:204 -- has n - 2)
    int d[MAXL + 2];       // d[0] = S, d[1..n-1] hidden, d[n] = out
    int64_t oW[MAXL + 1], ob[MAXL + 1], oStd, count;
};

bool make_dims(const int *dims, int n_dims, bool with_std, NetDims *nd)
{
    if (!dims || n_dims < 2 || n_dims > MAXL + 2) return false;
    nd->n = n_dims - 1;
    nd->n_act = nd->n - 1;
    int64_t o = 0;
    for (int i = 0; i < n_dims; ++i) {
        if (dims[i] < 1 || dims[i] > ERL_MAXN_WIDTH) return false;
        nd->d[i] = dims[i];
    }
    for (int l = 0; l < nd->n; ++l) {
        nd->oW[l] = o;
        o += (int64_t)dims[l + 1] * dims[l];
        nd->ob[l] = o;
        o += dims[l + 1];
    }
    nd->oStd = o;
    nd->count = o + (with_std ? dims[n_dims - 1] : 0);
    return true;
}


// sum `n_slabs` partial results of `stride` floats each (fixed order -> deterministic); defined in mlp.hip
extern "C" int erl_grad_reduce_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, void *stream);

// the weight-gradient contraction is split over chunks of DW_CHUNK batch rows when the batch is at least 4 chunks tall
constexpr int DW_CHUNK = 256;
// per-split partials of the widest layer: dW (NK floats) followed by db (at most ERL_MAXN_WIDTH floats)
inline int64_t dw_scratch_floats(int64_t M, int64_t NK) { return M >= 4 * DW_CHUNK ? ((M + DW_CHUNK - 1) / DW_CHUNK) * (NK + ERL_MAXN_WIDTH) : 0; }

// db[N] = column sums of dZ[M][N] (row-major): each block sums a slice of rows into its own partial (threads along the
// columns: coalesced), a fixed-order fold finishes.  (used for dL/dstd_log; the layers' bias gradients come out of the weight-gradient GEMM.)
constexpr int CS_ROWS = 128;   // rows per block
constexpr int CS_MAX_PART = 1024;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ dZ, int M, int N, float *__restrict__ part)
{
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    for (int c = threadIdx.x; c < N; c += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += dZ[(size_t)r * N + c];
            s1 += dZ[(size_t)(r + 1) * N + c];
            s2 += dZ[(size_t)(r + 2) * N + c];
            s3 += dZ[(size_t)(r + 3) * N + c];
        }
        for (; r < r1; ++r) s0 += dZ[(size_t)r * N + c];
        part[(size_t)blockIdx.x * N + c] = (s0 + s1) + (s2 + s3);
    }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float *__restrict__ part, int nparts, int N, float *__restrict__ db)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * N + c];
    db[c] = s;
}

// `part`: scratch of ceil(M / CS_ROWS) * N floats
int colsum(hipStream_t s, const float *dZ, float *part, float *db, int M, int N)
{
    const int nparts = (M + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nparts), dim3(256), 0, s, dZ, M, N, part);
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, nparts, N, db);
    return 0;
}
inline int64_t colsum_scratch_floats(int64_t M, int64_t maxN) { return ((M + CS_ROWS - 1) / CS_ROWS) * maxN; }

// ---------------------------------------------------------------------------------------------------------
// hand-written pieces
// ---------------------------------------------------------------------------------------------------------
// X[b][:] = (states[row(b)][:] - avg) / (std + 1e-4); row(b) = b (ids == NULL) or (id % H) * N + id // H
__global__ __launch_bounds__(256) void gather_norm_kernel(const float *__restrict__ states, const float *__restrict__ avg,
                                                          const float *__restrict__ sd, const int64_t *__restrict__ ids,
                                                          int64_t H, int64_t N, int S, int64_t rows, float *__restrict__ X,
                                                          float *__restrict__ raw_copy)
{
    const int64_t total = rows * S;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / S;
        const int c = (int)(e - b * S);
        int64_t row = b;
        if (ids) {
            const int64_t id = ids[b];
            const int64_t n = id / H, t = id - n * H;
            row = t * N + n;
        }
        const float raw = states[row * S + c];
        if (raw_copy) raw_copy[e] = raw;
        X[e] = (raw - avg[c]) / (sd[c] + 1e-4f);
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ p, float v, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) p[e] = v;
}

// rollout sampling: a = mean + std * eps, log-prob, tanh (AgentPPO.py:368-376, :388-390); one thread per env
__global__ __launch_bounds__(256) void sample_kernel(const float *__restrict__ Y, const float *__restrict__ std_log, int A,
                                                     int64_t N, const float *__restrict__ noise, uint64_t seed, uint64_t counter,
                                                     float *__restrict__ o_action, float *__restrict__ o_logprob,
                                                     float *__restrict__ o_env)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
        const float eps = noise ? noise[n * A + a] : philox_normal(seed, counter, (uint32_t)n, (uint32_t)a);
        const float sdv = expf(std_log[a]), var = sdv * sdv;
        const float mean = Y[n * A + a];
        const float act = mean + sdv * eps;
        const float diff = act - mean;
        lp += -(diff * diff) / (2.f * var) - logf(sdv) - kLogSqrt2PiN;
        if (o_action) o_action[n * A + a] = act;
        if (o_env) o_env[n * A + a] = tanhf(act);
    }
    if (o_logprob) o_logprob[n] = lp;
}

// PPO objective on gathered rows (AgentPPO.py:189-204).  ACTOR: Y holds the means (B, A) on entry and dL/dmean on exit;
// DSL (B, A) receives the per-row dL/dstd_log terms.  CRITIC: Y (B, 1) holds values on entry, dL/dv on exit.
// Per-block partial sums of the logged objectives go to part[block][2].
template <bool ACTOR>
__global__ __launch_bounds__(256) void objective_kernel(float *__restrict__ Y, float *__restrict__ DSL, const int64_t *__restrict__ ids,
                                                        int64_t H, int64_t N, int A, int64_t B, const float *__restrict__ actions,
                                                        const uint8_t *__restrict__ unmasks, const float *__restrict__ xa_src,
                                                        const float *__restrict__ xb_src, const float *__restrict__ std_log,
                                                        float ratio_clip, float lambda_entropy, float inv_batch, int objective,
                                                        float *__restrict__ part)
{
    // part[block][2 (+ A)]: the block's sums of the two logged values and (actor) of dL/dstd_log per action -- folded in a
    // fixed order by fold_logs_kernel (the std_log gradient used to be a (B, A) tensor and two more launches)
    __shared__ float red[4];
    const int pstride = ACTOR ? 2 + A : 2;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l0 = 0.f, l1 = 0.f;
    if (ACTOR) {
        const bool valid = b < B;
        int64_t row = 0;
        float dlp = 0.f, ent_term = 0.f;
        if (valid) {
            const int64_t id = ids[b];
            const int64_t n = id / H, t = id - n * H;
            row = t * N + n;
            const float um = unmasks[row] ? 1.f : 0.f;
            float lp = 0.f;
            for (int a = 0; a < A; ++a) {
                const float sdv = expf(std_log[a]), var = sdv * sdv;
                const float diff = actions[row * A + a] - Y[b * A + a];
                lp += -(diff * diff) / (2.f * var) - logf(sdv) - kLogSqrt2PiN;
            }
            const PpoActorTerms o = ppo_actor_terms(objective, xb_src[row], lp, xa_src[row], ratio_clip, lambda_entropy, um, A, false);
            l0 = o.logged;
            l1 = o.ent_mask;
            dlp = o.dlp * inv_batch;
            ent_term = o.ent_w * inv_batch;
        }
        for (int a = 0; a < A; ++a) {                     // uniform trip count: every thread takes part in the block sums
            float dsl = 0.f;
            if (valid) {
                const float sdv = expf(std_log[a]), var = sdv * sdv;
                const float diff = actions[row * A + a] - Y[b * A + a];
                Y[b * A + a] = dlp * (diff / var);
                dsl = dlp * (diff * diff / var - 1.f) + ent_term;
                if (DSL) DSL[b * A + a] = dsl;
            }
            const float ts = block_sum(dsl, red);
            if (threadIdx.x == 0) part[(size_t)blockIdx.x * pstride + 2 + a] = ts;
        }
    } else if (b < B) {
        const int64_t id = ids[b];
        const int64_t n = id / H, t = id - n * H;
        const int64_t row = t * N + n;
        const float um = unmasks[row] ? 1.f : 0.f;
        const float diff = Y[b] - xa_src[row];
        l0 = diff * diff * um;
        Y[b] = 2.f * diff * um * inv_batch;
    }
    const float t0 = block_sum(l0, red), t1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * pstride + 0] = t0;
        part[(size_t)blockIdx.x * pstride + 1] = t1;
    }
}

// ---- discrete policy (ActorDiscretePPO, elegantrl/agents/AgentPPO.py:393-422) ------------------------------------
// torch.distributions.Categorical(probs = softmax(z)) works on logits = log(clamp(p, eps, 1 - eps)) with
// eps = float32 machine epsilon: log_prob(a) = logits[a], entropy = -sum p logits; the clamp has zero gradient outside.
constexpr int kMaxDiscrete = 64;             // action_dim of the discrete path
constexpr float kCatEps = 1.1920928955078125e-07f;

__device__ __forceinline__ void softmax_row(const float *__restrict__ z, int A, float *p)
{
    float mx = z[0];
    for (int a = 1; a < A; ++a) mx = fmaxf(mx, z[a]);
    float sum = 0.f;
    for (int a = 0; a < A; ++a) { p[a] = expf(z[a] - mx); sum += p[a]; }
    const float inv = 1.f / sum;
    for (int a = 0; a < A; ++a) p[a] *= inv;
}

// rollout sampling: inverse-CDF draw from softmax(logits) with u in [0, 1) (injected or Philox), log-prob of the draw
__global__ __launch_bounds__(256) void sample_categorical_kernel(const float *__restrict__ Y, int A, int64_t N,
                                                                 const float *__restrict__ uniform, uint64_t seed, uint64_t counter,
                                                                 int32_t *__restrict__ o_action, float *__restrict__ o_logprob,
                                                                 int64_t *__restrict__ o_env)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float p[kMaxDiscrete];
    softmax_row(Y + n * A, A, p);
    const float u = uniform ? uniform[n] : philox_uniform(seed, counter, (uint32_t)n);
    int act = A - 1;
    float c = 0.f;
    for (int a = 0; a < A; ++a) {
        c += p[a];
        if (u < c) { act = a; break; }
    }
    if (o_action) o_action[n] = act;
    if (o_env) o_env[n] = act;                                      // convert_action_for_env: action.long()
    if (o_logprob) o_logprob[n] = logf(fminf(fmaxf(p[act], kCatEps), 1.f - kCatEps));
}

// PPO objective of the discrete actor on gathered rows (AgentPPO.py:189-204 with get_logprob_entropy of :413-418):
// Y (B, A) holds the logits on entry and dL/dlogits on exit; part[block] = (sum surr * unmask, sum entropy * unmask).
__global__ __launch_bounds__(256) void objective_discrete_kernel(float *__restrict__ Y, const int64_t *__restrict__ ids, int64_t H,
                                                                 int64_t N, int A, int64_t B, const int32_t *__restrict__ actions,
                                                                 const uint8_t *__restrict__ unmasks, const float *__restrict__ logp_old,
                                                                 const float *__restrict__ advantages, float ratio_clip,
                                                                 float lambda_entropy, float inv_batch, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l0 = 0.f, l1 = 0.f;
    if (b < B) {
        const int64_t id = ids[b];
        const int64_t n = id / H, t = id - n * H;
        const int64_t row = t * N + n;
        const float um = unmasks[row] ? 1.f : 0.f;
        float p[kMaxDiscrete];
        float *z = Y + b * A;
        softmax_row(z, A, p);
        int act = actions[row];
        act = act < 0 ? 0 : (act >= A ? A - 1 : act);
        float ent = 0.f, ph = 0.f;                                  // entropy, sum_k p_k h_k with h_k = dH/dp_k
        for (int a = 0; a < A; ++a) {
            const bool inside = p[a] > kCatEps && p[a] < 1.f - kCatEps;
            const float L = logf(fminf(fmaxf(p[a], kCatEps), 1.f - kCatEps));
            ent -= p[a] * L;
            ph += p[a] * -(L + (inside ? 1.f : 0.f));
        }
        const bool a_inside = p[act] > kCatEps && p[act] < 1.f - kCatEps;
        const float lp = logf(fminf(fmaxf(p[act], kCatEps), 1.f - kCatEps));
        const float adv = advantages[row];
        const float ratio = expf(lp - logp_old[row]);
        const float w = adv > 0.f ? 1.f - ratio_clip : 1.f + ratio_clip;
        const float surr = adv * ratio * w;
        l0 = surr * um;
        l1 = ent * um;
        // loss = -(mean(surr um) - lambda mean(ent um)):  dL/dlp = -surr um / B,  dL/dent = lambda um / B
        const float dlp = a_inside ? -(surr * um) * inv_batch : 0.f;
        const float dent = lambda_entropy * um * inv_batch;
        for (int a = 0; a < A; ++a) {
            const bool inside = p[a] > kCatEps && p[a] < 1.f - kCatEps;
            const float L = logf(fminf(fmaxf(p[a], kCatEps), 1.f - kCatEps));
            const float h = -(L + (inside ? 1.f : 0.f));
            z[a] = dlp * ((a == act ? 1.f : 0.f) - p[a]) + dent * p[a] * (h - ph);
        }
    }
    const float t0 = block_sum(l0, red), t1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 2 + 0] = t0;
        part[(size_t)blockIdx.x * 2 + 1] = t1;
    }
}

// logs: fold the per-block partials part[nparts][pstride] in a fixed order (is_actor: 1 = Gaussian head, 2 = categorical head;
// columns 0, 1: the logged values; 2 + a: dL/dstd_log of action a -> dstd[a]).
// 256 threads = 32 column slots x 8 row groups: a thread adds every 8th partial of its column (8x fewer dependent loads than one
// thread per column: the fold was 7.8 us of a ~4.5 us launch floor), the 8 group sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void fold_logs_kernel(const float *__restrict__ part, int nparts, int pstride,
                                                        const float *__restrict__ std_log, int A, float inv_batch, int is_actor,
                                                        float *__restrict__ logs, float *__restrict__ dstd)
{
    __shared__ float s01[2];
    __shared__ float grp[8][32];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    for (int c0 = 0; c0 < pstride; c0 += 32) {
        const int c = c0 + col;
        float s = 0.f;
        if (c < pstride)
            for (int i = rg; i < nparts; i += 8) s += part[(size_t)i * pstride + c];
        grp[rg][col] = s;
        __syncthreads();
        if (rg == 0 && c < pstride) {
            float t = grp[0][col];
#pragma unroll
            for (int r = 1; r < 8; ++r) t += grp[r][col];
            if (c < 2) s01[c] = t;
            else if (dstd) dstd[c - 2] = t;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float s0 = s01[0], s1 = s01[1];
    if (is_actor == 2) {
        logs[1] = s0 * inv_batch;
        logs[2] = s1 * inv_batch;
    } else if (is_actor) {
        float ent = 0.f;
        for (int a = 0; a < A; ++a) ent += 1.4189385332046727418f + logf(expf(std_log[a]));
        logs[1] = s0 * inv_batch;
        logs[2] = ent * s1 * inv_batch;
    } else {
        logs[0] = s0 * inv_batch;
        logs[3] = 0.f;
    }
}

inline int grid1d(int64_t total)
{
    int64_t g = erl_cdiv(total, 256);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

struct Ws {
    char *base;
    int64_t used, cap;
    float *take(int64_t floats)
    {
        float *p = (float *)(base + used);
        used += ((floats * 4 + 255) / 256) * 256;
        return used <= cap ? p : nullptr;
    }
};

int64_t ws_floats_forward(const NetDims &nd, int64_t rows)
{
    int64_t f = rows * nd.d[0] + 64;
    for (int l = 1; l <= nd.n; ++l) f += rows * nd.d[l] + 64;
    return f;
}

// forward pass into workspace buffers; act[l] = activation after layer l (act[0] = X), gd: store GELU' per hidden layer.
// One launch per dense layer (bias and GELU are the GEMM's epilogue).
int forward(hipStream_t s, const NetDims &nd, const float *P, int64_t rows, float **act, float **gd)
{
    for (int l = 0; l < nd.n; ++l) {
        const bool hidden = l < nd.n_act;
        int rc = dense_forward(s, act[l], P + nd.oW[l], P + nd.ob[l], act[l + 1], hidden && gd ? gd[l + 1] : nullptr, (int)rows,
                               nd.d[l + 1], nd.d[l], hidden);
        if (rc) return rc;
    }
    return 0;
}


// dW[Nw][K] = dZ^T . X and db[Nw] = column sums of dZ, one launch: the reduction over the `rows` samples is split in
// chunks of DW_CHUNK rows over blockIdx.z when scratch is available; each split writes its own partial (dW partials to
// dw_scratch, db partials to cs_scratch) and the partials are summed in a fixed order (deterministic).
int dense_weight_grad(hipStream_t s, const float *dZ, const float *X, float *dW, float *db, int rows, int Nw, int K, float *dw_scratch,
                      float *cs_scratch)
{
    GemmArgs g{};
    g.A = dZ; g.lda = Nw; g.B = X; g.ldb = K; g.M = Nw; g.N = K; g.K = rows; g.ldc = K;
    const bool split = dw_scratch && cs_scratch && rows >= 4 * DW_CHUNK;
    const int nsplit = split ? (int)erl_cdiv(rows, DW_CHUNK) : 1;
    // the parameter block keeps b right behind W: each split's db partial is then laid out right behind its dW partial and
    // ONE fixed-order sum produces both (a dependent launch costs ~7 us whatever it computes)
    const bool joint = split && db == dW + (size_t)Nw * K;
    g.kchunk = split ? DW_CHUNK : rows;
    g.C = split ? dw_scratch : dW;
    g.c_split = (int64_t)Nw * K + (joint ? Nw : 0);
    g.rowsum = joint ? dw_scratch + (size_t)Nw * K : (split ? cs_scratch : db);
    g.rs_split = joint ? g.c_split : Nw;
    int rc = gemm_launch<OP_OC, OP_OC, EPI_PARTIAL>(s, g, nsplit, "dense_weight_grad");
    if (rc || !split) return rc;
    if (joint) return erl_grad_reduce_f32(dw_scratch, nsplit, g.c_split, dW, (void *)s);
    if ((rc = erl_grad_reduce_f32(dw_scratch, nsplit, (int64_t)Nw * K, dW, (void *)s))) return rc;
    return erl_grad_reduce_f32(cs_scratch, nsplit, Nw, db, (void *)s);
}

// backward through an MLP whose forward was run by forward(): dZ = dL/d(output).  Writes weight / bias gradients into
// G (same layout as the parameter block) when G != nullptr, and dL/d(input) into dX0 when dX0 != nullptr
// (accumulating into it when acc_dx0).  tmpA / tmpB: two scratch buffers of rows * max-width floats; cs_scratch:
// colsum_scratch_floats(rows, max width) floats (bias-gradient partials); dw_scratch: dw_scratch_floats(rows, max W).
// Per layer: one launch for dW + db (+ two fixed-order sums when the batch is split), one for dX with GELU' applied.
int backward(hipStream_t s, const NetDims &nd, const float *P, int64_t rows, float *const *act, float *const *gd,
             const float *dZ, float *G, float *cs_scratch, float *dX0, bool acc_dx0, float *tmpA, float *tmpB,
             float *dw_scratch = nullptr)
{
    int rc;
    for (int l = nd.n - 1; l >= 0; --l) {
        const int K = nd.d[l], Nw = nd.d[l + 1];
        if (G && (rc = dense_weight_grad(s, dZ, act[l], G + nd.oW[l], G + nd.ob[l], (int)rows, Nw, K, dw_scratch, cs_scratch))) return rc;
        if (l > 0) {
            float *dH = (dZ == tmpA) ? tmpB : tmpA;
            if ((rc = dense_backward_input(s, dZ, P + nd.oW[l], dH, l - 1 < nd.n_act ? gd[l] : nullptr, false, (int)rows, Nw, K))) return rc;
            dZ = dH;
        } else if (dX0) {
            if ((rc = dense_backward_input(s, dZ, P + nd.oW[0], dX0, nullptr, acc_dx0, (int)rows, Nw, K))) return rc;
        }
    }
    return 0;
}

}  // namespace
