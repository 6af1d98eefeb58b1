// SAC update step: the consumer on the other side of ReplayBuffer.sample.  gfx950.
//
// Replaces AgentSAC.update_objectives (elegantrl/agents/AgentSAC.py:42-86) after the sample: tanh-Gaussian actor
// (ActorSAC :167-199), critic ensemble with a shared (state, action) encoder (CriticEnsemble :243-259), temperature
// update, soft target update (AgentBase.py:270-278) and the three clip + Adam steps (AgentBase.py:239-248).  Quirks are
// reproduced: Normal.log_prob is evaluated at the MEAN (:197), tanh correction log(1 - tanh^2 + 1e-6) (:198), the actor is
// trained against the TARGET ensemble's mean (:83), alpha is read after its own Adam step and only then clamped (:79-81).
//
// Batches are small (256..1024 rows): the step is launch-latency bound, so the win over ~300 ATen dispatches is that
// ONE C call enqueues everything.  Dense layers are the layered path's own fp32 MFMA GEMMs (gemm_tiles.h).
//
// Parameter blocks (flat fp32):
//   actor   build_mlp([S, h0..hL-1]) with GELU after every layer, then Linear(hL-1, 2A):  W b ... Whead bhead
//   critic  encoder Linear(S + A, h0) (no activation), then E decoders build_mlp([h0, h1, .., hL-1, 1]):  We be | dec0 | dec1 ...
#include <stdlib.h>

#include "mlpn_common.h"

extern "C" int erl_clip_adam_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                                 const int64_t *group_len, int n_groups, const int32_t *step_base, int32_t step_offset, float lr,
                                 float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream);

namespace {

constexpr int MAXE = 16;
constexpr float kLogSqrt2PiS = 0.91893853320467274178f;

struct SacDims {
    int S, A, E, L;
    NetDims actor, enc, dec;
    int64_t Pa, Pc;
};

bool make_sac_dims(int S, int A, const int *hidden, int n_hidden, int E, SacDims *d, int variant = ERL_SAC_ACTOR_SAC)
{
    if (S < 1 || A < 1 || !hidden || n_hidden < 1 || n_hidden > MAXL || E < 1 || E > MAXE) return false;
    int dims[MAXL + 2];
    dims[0] = S;
    for (int i = 0; i < n_hidden; ++i) dims[i + 1] = hidden[i];
    dims[n_hidden + 1] = 2 * A;
    if (!make_dims(dims, n_hidden + 2, false, &d->actor)) return false;
    // ActorFixSAC (AgentSAC.py:201-243): encoder_s = build_mlp([S, *net_dims]) keeps its last layer RAW; the two one-layer decoders
    // (mean | log_std) are the rows [0, A) | [A, 2A) of one Linear(h, 2A) -- the same parameter count and layout as ActorSAC's net_a
    if (variant == ERL_SAC_ACTOR_FIX) d->actor.n_act = n_hidden - 1;
    int e[2] = {S + A, hidden[0]};
    if (!make_dims(e, 2, false, &d->enc)) return false;
    int dd[MAXL + 2];
    for (int i = 0; i < n_hidden; ++i) dd[i] = hidden[i];
    dd[n_hidden] = 1;
    if (!make_dims(dd, n_hidden + 1, false, &d->dec)) return false;
    d->S = S; d->A = A; d->E = E; d->L = n_hidden;
    d->Pa = d->actor.count;
    d->Pc = d->enc.count + (int64_t)E * d->dec.count;
    return true;
}

__global__ __launch_bounds__(256) void concat_kernel(const float *__restrict__ s, const float *__restrict__ a, int S, int A, int64_t B,
                                                     float *__restrict__ out)
{
    const int W = S + A;
    const int64_t total = B * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / W;
        const int c = (int)(e - b * W);
        out[e] = c < S ? s[b * S + c] : a[b * A + (c - S)];
    }
}

// ActorSAC.get_action_logprob head: Y (B, 2A) = [mean | log_std] -> tanh action, log-prob; keeps eps for the backward
__global__ __launch_bounds__(256) void head_forward_kernel(const float *__restrict__ Y, const float *__restrict__ noise, uint64_t seed,
                                                           uint64_t counter, int A, int64_t B, float *__restrict__ act_t,
                                                           float *__restrict__ logprob, float *__restrict__ eps_out,
                                                           const float *__restrict__ xs, int S, float *__restrict__ xa, int variant)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    // the critic's input row [state | action] is written here as well (one launch less than a separate concat)
    if (xa)
        for (int c = 0; c < S; ++c) xa[b * (S + A) + c] = xs[b * S + c];
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
        const float mean = Y[b * 2 * A + a], ls = Y[b * 2 * A + A + a];
        const float lsc = fminf(fmaxf(ls, variant == ERL_SAC_ACTOR_FIX ? -20.f : -16.f), 2.f);
        const float sd = expf(lsc);
        const float eps = noise ? noise[b * A + a] : philox_normal(seed, counter, (uint32_t)b, (uint32_t)a);
        const float u = mean + sd * eps;
        const float t = tanhf(u);
        act_t[b * A + a] = t;
        if (xa) xa[b * (S + A) + S + a] = t;
        if (eps_out) eps_out[b * A + a] = eps;
        if (variant == ERL_SAC_ACTOR_FIX) {
            // ActorFixSAC.get_action_logprob (AgentSAC.py:226-243): the log-prob AT the sample, -log_std - eps^2 / 2 - log sqrt(2 pi), and
            // the tanh correction in its softplus form, -(log 2 - u - softplus(-2 u)) * 2 (nn.Softplus: beta 1, threshold 20)
            const float x = -2.f * u;
            const float sp = x > 20.f ? x : log1pf(expf(x));
            lp += (-lsc - (eps * eps) * 0.5f - kLogSqrt2PiS) - (0.69314718055994530942f - u - sp) * 2.f;
        } else {
            lp += (-logf(sd) - kLogSqrt2PiS) - logf(-(t * t) + 1.000001f);
        }
    }
    logprob[b] = lp;
}

// q_label = reward + undone * gamma * (min_e q_target - next_logprob * alpha)     (AgentSAC.py:52-55)
__global__ __launch_bounds__(256) void q_label_kernel(const float *__restrict__ qt, int E, int64_t B, const float *__restrict__ reward,
                                                      const float *__restrict__ undone, const float *__restrict__ next_lp,
                                                      const float *__restrict__ alpha_log, float gamma, float *__restrict__ label)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float m = qt[b];
    for (int e = 1; e < E; ++e) m = fminf(m, qt[(size_t)e * B + b]);
    const float alpha = expf(alpha_log[0]);
    label[b] = reward[b] + (undone[b] * gamma) * (m - next_lp[b] * alpha);
}

// critic objective: td = mean_e (q - label)^2 * unmask; obj = mean_b (td w); dq[e][b] = 2 (q - label) unmask w / (E B), with the
// importance-sampling weights w of prioritised replay (AgentSAC.py:60-62; w = 1 without) and td written out for the priorities
__global__ __launch_bounds__(256) void critic_loss_kernel(const float *__restrict__ q, const float *__restrict__ label,
                                                          const float *__restrict__ unmask, const float *__restrict__ is_weight,
                                                          const float *__restrict__ fit, int E, int64_t B, float *__restrict__ dq,
                                                          float *__restrict__ td_out, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float td = 0.f;
    if (b < B) {
        const float l = label[b], um = unmask[b], w = is_weight ? is_weight[b] : 1.f;
        float s = 0.f;
        for (int e = 0; e < E; ++e) {
            const float diff = q[(size_t)e * B + b] - l;
            s += diff * diff;
            float d = 2.f * diff * um * w / ((float)E * (float)B);
            if (fit) d += fit[e];                      // the lambda_fit_cum_r term's gradient (fit_cum_r_kernel)
            dq[(size_t)e * B + b] = d;
        }
        td = (s / (float)E) * um;
        if (td_out) td_out[b] = td;
        td *= w;
    }
    const float t = block_sum(td, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// `lambda_fit_cum_r` term of the critic objective (AgentSAC.py:66-68):
//   obj += lambda * mean_e ( mean_b cum_reward[b] - mean_b q[e][b] )^2
// One workgroup: the E + 1 batch means, then fit[e] = d term / d q[e][b] = 2 lambda (qbar_e - cbar) / (E B) and fit[E] = the term.
__global__ __launch_bounds__(256) void fit_cum_r_kernel(const float *__restrict__ q, const float *__restrict__ cum_reward, int E, int64_t B,
                                                        float lambda_fit, float *__restrict__ fit)
{
    __shared__ float red[4];
    float c = 0.f;
    for (int64_t i = threadIdx.x; i < B; i += 256) c += cum_reward[i];
    const float cbar = block_sum(c, red) / (float)B;
    float term = 0.f;
    for (int e = 0; e < E; ++e) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < B; i += 256) s += q[(size_t)e * B + i];
        const float qbar = block_sum(s, red) / (float)B;
        const float diff = cbar - qbar;
        term += diff * diff;
        if (threadIdx.x == 0) fit[e] = 2.f * lambda_fit * (qbar - cbar) / ((float)E * (float)B);
    }
    if (threadIdx.x == 0) fit[E] = lambda_fit * term / (float)E;
}

// single-block deterministic reductions of small vectors:  out = scale * sum_i x[i] (+ bias) (+ *add)
__global__ __launch_bounds__(256) void sum_kernel(const float *__restrict__ x, int64_t n, float scale, float bias, float *__restrict__ out,
                                                  const float *__restrict__ add = nullptr)
{
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = t * scale + bias + (add ? add[0] : 0.f);
}

// temperature step in one launch: g = target_entropy - mean(logprob) (sum_kernel's reduction), then clip_adam_kernel's
// arithmetic for a one-element group (AgentSAC.py:76-79 through AgentBase.optimizer_backward :239-248); g is left in g_out
__global__ __launch_bounds__(256) void alpha_step_kernel(const float *__restrict__ lp, int64_t n, float target_entropy, float *__restrict__ g_out,
                                                         float *__restrict__ alpha_log, float *__restrict__ m1, float *__restrict__ m2,
                                                         float beta1, float beta2, float eps, float max_norm, float step_size, float bc2_sqrt)
{
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += lp[i];
    const float t = block_sum(s, red);
    if (threadIdx.x != 0) return;
    const float g = t * (-1.0f / (float)n) + target_entropy;
    g_out[0] = g;
    const float total_norm = (float)sqrt((double)g * (double)g);
    float coef = max_norm / (total_norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    float e_m1 = m1[0], e_m2 = m2[0], e_p = alpha_log[0];
    erl_adam_update(erl_mul_rn(g, erl_mul_rn(1.0f, coef)), e_m1, e_m2, e_p, beta1, beta2, eps, step_size, bc2_sqrt);   // the library's one Adam
    m1[0] = e_m1;
    m2[0] = e_m2;
    alpha_log[0] = e_p;
}

__global__ __launch_bounds__(256) void fillk_kernel(float *__restrict__ p, float v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

// actor objective value and dL/d(head output) for L = -(mean q_pg - alpha * mean logprob)   (AgentSAC.py:74-85)
//   action = tanh(u), u = mean + std eps, std = exp(clamp(ls, -16, 2));  logprob = sum_a [-log std - c - log(1 - t^2 + 1e-6)]
__global__ __launch_bounds__(256) void head_backward_kernel(const float *__restrict__ Y, const float *__restrict__ act_t,
                                                            const float *__restrict__ eps, const float *__restrict__ dA, int ldA,
                                                            const float *__restrict__ alpha_log, int A, int64_t B,
                                                            float *__restrict__ dY, int variant)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float alpha = expf(alpha_log[0]);
    const float dlp = alpha / (float)B;                 // dL/dlogprob_b
    for (int a = 0; a < A; ++a) {
        const float ls = Y[b * 2 * A + A + a];
        const float lo = variant == ERL_SAC_ACTOR_FIX ? -20.f : -16.f;
        const float lsc = fminf(fmaxf(ls, lo), 2.f);
        const float sd = expf(lsc);
        const float t = act_t[b * A + a];
        const float one_m = 1.f - t * t;
        // d logprob / du: -d/du log(1 - t^2 + 1e-6) for ActorSAC; ActorFixSAC: -2 d/du (log 2 - u - softplus(-2 u)) = 2 tanh(u)
        const float dlp_du = variant == ERL_SAC_ACTOR_FIX ? 2.f * t : 2.f * t * one_m / (one_m + 1e-6f);
        const float du = dA[b * ldA + a] * one_m + dlp * dlp_du;
        const bool inside = ls >= lo && ls <= 2.f;      // clamp passes its gradient inside [min, max]
        dY[b * 2 * A + a] = du;
        dY[b * 2 * A + A + a] = inside ? du * sd * eps[b * A + a] - dlp : 0.f;
    }
}

// obj_actor = mean_b mean_e q_pg - alpha * mean_b logprob
__global__ __launch_bounds__(256) void actor_obj_kernel(const float *__restrict__ qpg, int E, int64_t B, const float *__restrict__ lp,
                                                        const float *__restrict__ alpha_log, float *__restrict__ out)
{
    __shared__ float red[4];
    float sq = 0.f, sl = 0.f;
    for (int64_t i = threadIdx.x; i < (int64_t)E * B; i += 256) sq += qpg[i];
    for (int64_t i = threadIdx.x; i < B; i += 256) sl += lp[i];
    const float tq = block_sum(sq, red), tl = block_sum(sl, red);
    if (threadIdx.x == 0) out[0] = tq / ((float)E * (float)B) - expf(alpha_log[0]) * (tl / (float)B);
}

__global__ __launch_bounds__(256) void soft_update_kernel(float *__restrict__ tar, const float *__restrict__ cur, float tau, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        tar[i] = erl_soft_update(cur[i], tar[i], tau);
}

__global__ void clamp_alpha_kernel(float *alpha_log)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) alpha_log[0] = fminf(fmaxf(alpha_log[0], -16.f), 2.f);
}

struct CriticWs {
    float *enc;                       // (B, h0)
    float *act[MAXE][MAXL + 2];       // decoder activations; act[e][0] = enc
    float *gd[MAXE][MAXL + 2];
    float *q;                         // [E][B]
    int64_t st[MAXL + 2];             // floats between decoder e and e + 1 of act[.][l] / gd[.][l] (0: the shared encoder output)
};

int64_t critic_ws_floats(const SacDims &d, int64_t B)
{
    int64_t f = B * d.enc.d[1] + 64 + (int64_t)d.E * B + 64;
    for (int l = 1; l < d.dec.n; ++l) f += (int64_t)d.E * 2 * (B * d.dec.d[l] + 64);
    return f;
}

bool carve_critic(Ws &ws, const SacDims &d, int64_t B, CriticWs *c)
{
    c->enc = ws.take(B * d.enc.d[1]);
    c->q = ws.take((int64_t)d.E * B);
    c->st[0] = 0;
    c->st[d.dec.n] = B;
    bool ok = c->enc && c->q;
    for (int l = 1; l < d.dec.n; ++l) {               // layer l of every decoder in one block (the batched GEMMs step by st[l])
        c->st[l] = ((B * d.dec.d[l] + 63) / 64) * 64;
        float *a = ws.take((int64_t)d.E * c->st[l]), *g = ws.take((int64_t)d.E * c->st[l]);
        ok = ok && a && g;
        for (int e = 0; e < d.E; ++e) {
            c->act[e][l] = a + (size_t)e * c->st[l];
            c->gd[e][l] = g + (size_t)e * c->st[l];
        }
    }
    for (int e = 0; e < d.E; ++e) {
        c->act[e][0] = c->enc;
        c->gd[e][0] = nullptr;
        c->act[e][d.dec.n] = c->q + (size_t)e * B;   // (B, 1) output column of decoder e
        c->gd[e][d.dec.n] = nullptr;
    }
    return ok;
}

// The E decoders are E same-shaped MLPs on their own parameter blocks (dec.count apart): at the off-policy batch sizes each of
// their layers is a 5-8 us launch-bound GEMM, so the ensemble runs as ONE batched launch per layer (blockIdx.z = decoder).
constexpr int64_t kBatchedRows = 1024;   // above this the per-decoder launches fill the device on their own (64 x 64 tiles)

int decoders_forward(hipStream_t s, const SacDims &d, const float *Pdec, int64_t B, CriticWs &c, bool keep)
{
    const NetDims &nd = d.dec;
    for (int l = 0; l < nd.n; ++l) {
        const bool hidden = l + 1 < nd.n;
        GemmArgs g{};
        g.A = c.act[0][l]; g.sA = c.st[l]; g.lda = nd.d[l];
        g.B = Pdec + nd.oW[l]; g.sB = nd.count; g.ldb = nd.d[l];
        g.M = (int)B; g.N = nd.d[l + 1]; g.K = nd.d[l];
        g.C = c.act[0][l + 1]; g.sC = c.st[l + 1]; g.ldc = nd.d[l + 1];
        g.bias = Pdec + nd.ob[l]; g.sBias = nd.count;
        g.G = hidden && keep ? c.gd[0][l + 1] : nullptr; g.sG = c.st[l + 1];
        g.nbatch = d.E;
        int rc = hidden ? gemm_launch<OP_RC, OP_RC, EPI_BIAS_GELU>(s, g, 1, "decoders_forward") : gemm_launch<OP_RC, OP_RC, EPI_BIAS>(s, g, 1, "decoders_forward");
        if (rc) return rc;
    }
    return 0;
}

// dst = ((src_0 + src_1) + src_2) + ...: the order in which the per-decoder launches accumulated into the encoder gradient
__global__ __launch_bounds__(256) void sum_batches_kernel(const float *__restrict__ src, int E, int64_t stride, int64_t n, float *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = src[i];
        for (int e = 1; e < E; ++e) v += src[(size_t)e * stride + i];
        dst[i] = v;
    }
}

// backward through all E decoders: dq [E][B] -> parameter gradients Gdec (E blocks, dec.count apart; may be NULL) and the
// encoder-output gradient dEnc (B, h0) = sum_e.  tmpA / tmpB: E x tmp_stride floats each; dEncE: E x enc_stride floats.
int decoders_backward(hipStream_t s, const SacDims &d, const float *Pdec, int64_t B, CriticWs &c, const float *dq, float *Gdec,
                      float *dEnc, float *dEncE, int64_t enc_stride, float *tmpA, float *tmpB, int64_t tmp_stride)
{
    const NetDims &nd = d.dec;
    const float *dZ = dq;
    int64_t sdZ = B;
    int rc;
    for (int l = nd.n - 1; l >= 0; --l) {
        const int K = nd.d[l], Nw = nd.d[l + 1];
        if (Gdec) {                                   // dW = dZ^T . X, db = column sums of dZ (the GEMM's row sums)
            GemmArgs g{};
            g.A = dZ; g.sA = sdZ; g.lda = Nw;
            g.B = c.act[0][l]; g.sB = c.st[l]; g.ldb = K;
            g.M = Nw; g.N = K; g.K = (int)B; g.ldc = K;
            g.kchunk = (int)B;
            g.C = Gdec + nd.oW[l]; g.sC = nd.count; g.c_split = 0;
            g.rowsum = Gdec + nd.ob[l]; g.sRS = nd.count; g.rs_split = 0;
            g.nbatch = d.E;
            if ((rc = gemm_launch<OP_OC, OP_OC, EPI_PARTIAL>(s, g, 1, "decoders_backward(dW)"))) return rc;
        }
        GemmArgs g{};
        g.A = dZ; g.sA = sdZ; g.lda = Nw;
        g.B = Pdec + nd.oW[l]; g.sB = nd.count; g.ldb = K;
        g.M = (int)B; g.N = K; g.K = Nw; g.ldc = K;
        g.nbatch = d.E;
        if (l > 0) {
            float *dH = (dZ == tmpA) ? tmpB : tmpA;
            g.C = dH; g.sC = tmp_stride;
            g.G = c.gd[0][l]; g.sG = c.st[l];
            if ((rc = gemm_launch<OP_RC, OP_OC, EPI_MUL>(s, g, 1, "decoders_backward(dX)"))) return rc;
            dZ = dH;
            sdZ = tmp_stride;
        } else {
            g.C = dEncE; g.sC = enc_stride;
            if ((rc = gemm_launch<OP_RC, OP_OC, EPI_STORE>(s, g, 1, "decoders_backward(dEnc)"))) return rc;
            const int64_t n = B * K;
            hipLaunchKernelGGL(sum_batches_kernel, dim3(grid1d(n)), dim3(256), 0, s, dEncE, d.E, enc_stride, n, dEnc);
        }
    }
    return erl_hip_status(hipGetLastError(), "decoders_backward");
}

int critic_forward(hipStream_t s, const SacDims &d, const float *P, int64_t B, float *xa, CriticWs &c, bool keep)
{
    float *ea[2] = {xa, c.enc};
    int rc = forward(s, d.enc, P, B, ea, nullptr);      // one raw linear layer
    if (rc) return rc;
    if (B <= kBatchedRows) return decoders_forward(s, d, P + d.enc.count, B, c, keep);
    for (int e = 0; e < d.E; ++e)
        if ((rc = forward(s, d.dec, P + d.enc.count + (int64_t)e * d.dec.count, B, c.act[e], keep ? c.gd[e] : nullptr))) return rc;
    return 0;
}

}  // namespace

extern "C" int erl_sac_param_counts(int S, int A, const int *hidden, int n_hidden, int E, int64_t *actor_count, int64_t *critic_count)
{
    SacDims d;
    ERL_REQUIRE(make_sac_dims(S, A, hidden, n_hidden, E, &d), "erl_sac_param_counts: unsupported dims");
    if (actor_count) *actor_count = d.Pa;
    if (critic_count) *critic_count = d.Pc;
    return ERL_OK;
}

extern "C" int64_t erl_sac_workspace_bytes(int S, int A, const int *hidden, int n_hidden, int E, int64_t B)
{
    SacDims d;
    if (!make_sac_dims(S, A, hidden, n_hidden, E, &d) || B < 1) return -1;
    int maxd = S + A;
    for (int i = 0; i < n_hidden; ++i) maxd = hidden[i] > maxd ? hidden[i] : maxd;
    maxd = 2 * A > maxd ? 2 * A : maxd;
    int64_t f = 0;
    f += 2 * (B * (S + A) + 64);                                     // xa, dxa
    for (int l = 0; l <= d.actor.n; ++l) f += 2 * (B * d.actor.d[l] + 64);   // actor activations + GELU'
    f += critic_ws_floats(d, B);
    f += 4 * (B * A + 64) + 6 * (B + 64) + (int64_t)d.E * B + 64;    // actions, eps, dA, t | logprobs, label, ... | dq
    f += colsum_scratch_floats(B, maxd) + 64;                        // bias-gradient partials
    f += (2 * (int64_t)E + 1) * (B * maxd + 64) + (int64_t)E * (B * hidden[0] + 64) + B * 2 * A + 64;   // tmpA, tmpB (per decoder), dEnc, per-decoder dEnc, dHead
    f += d.Pa + d.Pc + 64 + 1024 + E + 64;                            // gradients, partials, lambda_fit_cum_r scratch
    if (erl_sac_fused_supported(S, A, hidden, n_hidden, E, B)) {      // the fused step (sac_fused.hip) carves its own layout
        const int64_t ff = erl_sac_fused_ws_floats(S, A, hidden[0], hidden[1], E, B, d.Pa, d.Pc);
        f = ff > f ? ff : f;
    }
    return f * 4 + 8192;
}

static int sac_update_impl(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                           float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                           const int *hidden, int n_hidden, int E, const float *state, const float *action,
                           const float *reward, const float *undone, const float *unmask, const float *next_state,
                           const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r, int64_t B,
                           const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter, float gamma,
                           float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                           int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes, const ErlRingSample *ring,
                           const ErlSacOptions *opt, void *stream);

extern "C" int erl_sac_update_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                                  float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                                  const int *hidden, int n_hidden, int E, const float *state, const float *action,
                                  const float *reward, const float *undone, const float *unmask, const float *next_state,
                                  const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r, int64_t B,
                                  const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter, float gamma,
                                  float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                                  int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    return sac_update_impl(actor_params, critic_params, target_params, alpha_log, actor_m, actor_v, critic_m, critic_v, alpha_m, alpha_v, S, A, hidden,
                           n_hidden, E, state, action, reward, undone, unmask, next_state, is_weight, td_error_out, cum_reward, lambda_fit_cum_r, B,
                           eps_next, eps_cur, seed, counter, gamma, target_entropy, tau, lr, beta1, beta2, eps_adam, max_norm, step, objs_out,
                           workspace, workspace_bytes, nullptr, nullptr, stream);
}

// erl_sac_update_f32 with the options of AgentModSAC (include/erl_hip.h, ErlSacOptions): ActorFixSAC's head, the actor step skipped by
// the two-time-scale rule, the actor's own Adam step count, the actor target's soft update
extern "C" int erl_sac_update_opt_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                                      float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                                      const int *hidden, int n_hidden, int E, const float *state, const float *action,
                                      const float *reward, const float *undone, const float *unmask, const float *next_state,
                                      const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r, int64_t B,
                                      const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter, float gamma,
                                      float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                                      int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes, const ErlSacOptions *opt,
                                      void *stream)
{
    return sac_update_impl(actor_params, critic_params, target_params, alpha_log, actor_m, actor_v, critic_m, critic_v, alpha_m, alpha_v, S, A, hidden,
                           n_hidden, E, state, action, reward, undone, unmask, next_state, is_weight, td_error_out, cum_reward, lambda_fit_cum_r, B,
                           eps_next, eps_cur, seed, counter, gamma, target_entropy, tau, lr, beta1, beta2, eps_adam, max_norm, step, objs_out,
                           workspace, workspace_bytes, nullptr, opt, stream);
}

// ReplayBuffer.sample + the step from one call (include/erl_hip.h): in the fused step the gather rides in the first launch
extern "C" int erl_sac_update_ring_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                                       float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                                       const int *hidden, int n_hidden, int E, const ErlRingSample *ring, float *state, float *action,
                                       float *reward, float *undone, float *unmask, float *next_state, int64_t B, const float *eps_next,
                                       const float *eps_cur, uint64_t seed, uint64_t counter, float gamma, float target_entropy, float tau,
                                       float lr, float beta1, float beta2, float eps_adam, float max_norm, int32_t step, float *objs_out,
                                       void *workspace, int64_t workspace_bytes, void *stream)
{
    ERL_REQUIRE(ring && ring->buf_states && ring->ids && (ring->row_floats || (ring->buf_actions && ring->buf_rewards && ring->buf_undones && ring->buf_unmasks)),
                "erl_sac_update_ring_f32: NULL ring tensor");
    ERL_REQUIRE(ring->row_floats == 0 || (ring->row_floats == erl_replay_row_floats(S, A) && ring->sample_len < ring->max_size &&
                                          (reinterpret_cast<uintptr_t>(ring->buf_states) & 15) == 0),
                "erl_sac_update_ring_f32: interleaved ring with row_floats=%lld (expected %lld), sample_len=%lld", (long long)ring->row_floats,
                (long long)erl_replay_row_floats(S, A), (long long)ring->sample_len);
    ERL_REQUIRE(ring->num_seqs >= 1 && ring->sample_len >= 1 && ring->sample_len <= ring->max_size,
                "erl_sac_update_ring_f32: bad ring shape max_size=%lld num_seqs=%lld sample_len=%lld", (long long)ring->max_size,
                (long long)ring->num_seqs, (long long)ring->sample_len);
    return sac_update_impl(actor_params, critic_params, target_params, alpha_log, actor_m, actor_v, critic_m, critic_v, alpha_m, alpha_v, S, A, hidden,
                           n_hidden, E, state, action, reward, undone, unmask, next_state, nullptr, nullptr, nullptr, 0.f, B, eps_next, eps_cur, seed,
                           counter, gamma, target_entropy, tau, lr, beta1, beta2, eps_adam, max_norm, step, objs_out, workspace, workspace_bytes, ring,
                           nullptr, stream);
}

// AgentBase.update_net's loop (elegantrl/agents/AgentBase.py:172-189) for the ring form: n_steps x erl_sac_update_ring_f32 from ONE call --
// step t samples with ids_all[t], counts as optimiser step step0 + t, keys its noise with counter0 + t and logs into objs_all[2 t ..].
// The interpreter is off the launch path: at config 3 a step is ~12 launches + 3 event operations, and the per-step Python / ctypes
// work (~35 us) had become longer than what the GPU waits between launches.
extern "C" int erl_sac_update_ring_loop_f32(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                                            float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                                            const int *hidden, int n_hidden, int E, const ErlRingSample *ring, const int64_t *ids_all,
                                            int64_t n_steps, float *state, float *action, float *reward, float *undone, float *unmask,
                                            float *next_state, int64_t B, uint64_t seed, uint64_t counter0, float gamma, float target_entropy,
                                            float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm, int32_t step0,
                                            float *objs_all, void *workspace, int64_t workspace_bytes, void *stream)
{
    ERL_REQUIRE(ring && ids_all && objs_all && n_steps >= 0 && B >= 1, "erl_sac_update_ring_loop_f32: bad argument");
    for (int64_t t = 0; t < n_steps; ++t) {
        ErlRingSample r = *ring;
        r.ids = ids_all + t * B;
        const int rc = erl_sac_update_ring_f32(actor_params, critic_params, target_params, alpha_log, actor_m, actor_v, critic_m, critic_v, alpha_m,
                                               alpha_v, S, A, hidden, n_hidden, E, &r, state, action, reward, undone, unmask, next_state, B, nullptr,
                                               nullptr, seed, counter0 + (uint64_t)t, gamma, target_entropy, tau, lr, beta1, beta2, eps_adam,
                                               max_norm, step0 + (int32_t)t, objs_all + 2 * t, workspace, workspace_bytes, stream);
        if (rc) return rc;
    }
    return ERL_OK;
}

static int sac_update_impl(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m,
                           float *actor_v, float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A,
                           const int *hidden, int n_hidden, int E, const float *state, const float *action,
                           const float *reward, const float *undone, const float *unmask, const float *next_state,
                           const float *is_weight, float *td_error_out, const float *cum_reward, float lambda_fit_cum_r, int64_t B,
                           const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter, float gamma,
                           float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                           int32_t step, float *objs_out, void *workspace, int64_t workspace_bytes, const ErlRingSample *ring,
                           const ErlSacOptions *opt, void *stream)
{
    ERL_REQUIRE(actor_params && critic_params && target_params && alpha_log && actor_m && actor_v && critic_m && critic_v && alpha_m &&
                    alpha_v && state && action && reward && undone && unmask && next_state && objs_out && workspace,
                "erl_sac_update_f32: NULL tensor");
    const int variant = opt ? opt->actor_variant : ERL_SAC_ACTOR_SAC;
    const bool update_actor = opt ? opt->update_actor != 0 : true;
    const int32_t actor_step = opt && opt->actor_step > 0 ? opt->actor_step : step;
    float *actor_target = opt ? opt->actor_target_params : nullptr;
    ERL_REQUIRE(variant == ERL_SAC_ACTOR_SAC || variant == ERL_SAC_ACTOR_FIX, "erl_sac_update_opt_f32: unknown actor_variant %d", variant);
    SacDims d;
    ERL_REQUIRE(make_sac_dims(S, A, hidden, n_hidden, E, &d, variant), "erl_sac_update_f32: unsupported dims");
    ERL_REQUIRE(B >= 1 && B < (1LL << 24) && step >= 1, "erl_sac_update_f32: bad argument");
    ERL_REQUIRE(lambda_fit_cum_r == 0.f || cum_reward, "erl_sac_update_f32: lambda_fit_cum_r != 0 needs the batch's cum_reward");
    ERL_REQUIRE(workspace_bytes >= erl_sac_workspace_bytes(S, A, hidden, n_hidden, E, B), "erl_sac_update_f32: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int rc;

    // Off-policy batch sizes with two hidden layers up to 256 wide (config 3): the fused step, 12 launches (sac_fused.hip).
    // ERL_SAC_FUSED=0 keeps the layered step below (A/B runs); lambda_fit_cum_r != 0 (off by default) is layered only.
    static const bool fused_on = [] { const char *e = getenv("ERL_SAC_FUSED"); return !(e && atoi(e) == 0); }();
    // (the fused step implements ActorSAC's head with every optimiser on one step count: AgentModSAC's options take the layered step)
    const bool plain = variant == ERL_SAC_ACTOR_SAC && update_actor && actor_step == step && !actor_target;
    if (fused_on && plain && lambda_fit_cum_r == 0.f && erl_sac_fused_supported(S, A, hidden, n_hidden, E, B)) {
        const int64_t aoff[6] = {d.actor.oW[0], d.actor.ob[0], d.actor.oW[1], d.actor.ob[1], d.actor.oW[2], d.actor.ob[2]};
        const int64_t coff[8] = {d.enc.oW[0], d.enc.ob[0], d.enc.count, d.dec.oW[0], d.dec.ob[0], d.dec.oW[1], d.dec.ob[1], d.dec.count};
        return erl_sac_update_fused(actor_params, critic_params, target_params, alpha_log, actor_m, actor_v, critic_m, critic_v, alpha_m, alpha_v, S,
                                    A, hidden[0], hidden[1], E, aoff, coff, d.Pa, d.Pc, state, action, reward, undone, unmask, next_state,
                                    is_weight, td_error_out, B, eps_next, eps_cur, seed, counter, gamma, target_entropy, tau, lr, beta1, beta2,
                                    eps_adam, max_norm, step, objs_out, (float *)workspace, ring, s);
    }
    if (ring) {        // the layered step reads a finished batch: the sample as a launch of its own
        rc = ring->row_floats
                 ? erl_replay_sample_rows_f32(ring->buf_states, ring->max_size, ring->num_seqs, S, A, ring->ids, B, ring->sample_len,
                                              const_cast<float *>(state), const_cast<float *>(action), const_cast<float *>(reward),
                                              const_cast<float *>(undone), const_cast<float *>(unmask), const_cast<float *>(next_state),
                                              ring->out_ids0, ring->out_ids1, stream)
                 : erl_replay_sample_f32(ring->buf_states, ring->buf_actions, ring->buf_rewards, ring->buf_undones, ring->buf_unmasks, ring->max_size, ring->num_seqs, S,
                                   A, ring->ids, B, ring->sample_len, const_cast<float *>(state), const_cast<float *>(action),
                                   const_cast<float *>(reward), const_cast<float *>(undone), const_cast<float *>(unmask),
                                   const_cast<float *>(next_state), ring->out_ids0, ring->out_ids1, stream);
        if (rc) return rc;
    }

    Ws ws{(char *)workspace, 0, workspace_bytes};
    int maxd = S + A;
    for (int i = 0; i < n_hidden; ++i) maxd = hidden[i] > maxd ? hidden[i] : maxd;
    maxd = 2 * A > maxd ? 2 * A : maxd;
    float *xa = ws.take(B * (S + A)), *dxa = ws.take(B * (S + A));
    float *aact[MAXL + 2], *agd[MAXL + 2];
    for (int l = 0; l <= d.actor.n; ++l) {
        aact[l] = ws.take(B * d.actor.d[l]);
        agd[l] = (l >= 1 && l <= d.actor.n_act) ? ws.take(B * d.actor.d[l]) : nullptr;      // GELU' of the layers that have one
    }
    CriticWs cw;
    ERL_REQUIRE(carve_critic(ws, d, B, &cw), "erl_sac_update_f32: workspace layout");
    float *act_t = ws.take(B * A), *eps_used = ws.take(B * A);
    float *lp_next = ws.take(B), *lp_cur = ws.take(B), *label = ws.take(B), *cs_scr = ws.take(colsum_scratch_floats(B, maxd));   // bias-gradient partials
    float *dq = ws.take((int64_t)E * B);
    const int64_t tmp_stride = ((B * maxd + 63) / 64) * 64, enc_stride = ((B * d.enc.d[1] + 63) / 64) * 64;
    float *tmpA = ws.take((int64_t)E * tmp_stride), *tmpB = ws.take((int64_t)E * tmp_stride), *dEnc = ws.take(B * d.enc.d[1]);
    float *dEncE = ws.take((int64_t)E * enc_stride);
    const bool batched = B <= kBatchedRows;
    float *dHead = ws.take(B * 2 * A);
    float *g_actor = ws.take(d.Pa), *g_critic = ws.take(d.Pc), *g_alpha = ws.take(4);
    const int nparts = (int)erl_cdiv(B, 256);
    float *part = ws.take(nparts);
    float *fit = ws.take(E + 4);
    ERL_REQUIRE(part != nullptr && fit != nullptr, "erl_sac_update_f32: workspace layout (tail)");
    const dim3 rows_grid((unsigned)erl_cdiv(B, 256)), blk(256);

    // ---- (1) targets: next action / log-prob from the actor, min over the TARGET ensemble          (:50-55)
    aact[0] = const_cast<float *>(next_state);                      // the input layer reads the sample in place
    if ((rc = forward(s, d.actor, actor_params, B, aact, nullptr))) return rc;
    hipLaunchKernelGGL(head_forward_kernel, rows_grid, blk, 0, s, aact[d.actor.n], eps_next, seed, 2 * counter, A, B, act_t, lp_next,
                       (float *)nullptr, next_state, S, xa, variant);
    if ((rc = critic_forward(s, d, target_params, B, xa, cw, false))) return rc;
    hipLaunchKernelGGL(q_label_kernel, rows_grid, blk, 0, s, cw.q, E, B, reward, undone, lp_next, alpha_log, gamma, label);

    // ---- (2) critic objective, backward, clip + Adam, soft target update                          (:57-70)
    hipLaunchKernelGGL(concat_kernel, dim3(grid1d(B * (S + A))), blk, 0, s, state, action, S, A, B, xa);
    if ((rc = critic_forward(s, d, critic_params, B, xa, cw, true))) return rc;
    const bool fit_on = lambda_fit_cum_r != 0.f;                     // AgentSAC.py:66-68 (off by default: config.py:52)
    if (fit_on) hipLaunchKernelGGL(fit_cum_r_kernel, dim3(1), blk, 0, s, cw.q, cum_reward, E, B, lambda_fit_cum_r, fit);
    hipLaunchKernelGGL(critic_loss_kernel, rows_grid, blk, 0, s, cw.q, label, unmask, is_weight, fit_on ? fit : (const float *)nullptr, E, B, dq,
                       td_error_out, part);
    hipLaunchKernelGGL(sum_kernel, dim3(1), blk, 0, s, part, (int64_t)nparts, 1.0f / (float)B, 0.f, objs_out,
                       fit_on ? fit + E : (const float *)nullptr);
    if (batched) {
        if ((rc = decoders_backward(s, d, critic_params + d.enc.count, B, cw, dq, g_critic + d.enc.count, dEnc, dEncE, enc_stride, tmpA, tmpB,
                                    tmp_stride)))
            return rc;
    } else {
        for (int e = 0; e < E; ++e) {
            float *Gdec = g_critic + d.enc.count + (int64_t)e * d.dec.count;
            if ((rc = backward(s, d.dec, critic_params + d.enc.count + (int64_t)e * d.dec.count, B, cw.act[e], cw.gd[e], dq + (size_t)e * B,
                               Gdec, cs_scr, dEnc, e > 0, tmpA, tmpB)))
                return rc;
        }
    }
    {   // encoder: one raw linear layer, input xa
        float *ea[2] = {xa, cw.enc};
        if ((rc = backward(s, d.enc, critic_params, B, ea, nullptr, dEnc, g_critic, cs_scr, nullptr, false, tmpA, tmpB))) return rc;
    }
    {
        const int64_t off = 0, len = d.Pc;
        if ((rc = erl_clip_adam_f32(critic_params, g_critic, critic_m, critic_v, &off, &len, 1, nullptr, step, lr, beta1, beta2, eps_adam,
                                    max_norm, 1.0f, stream)))
            return rc;
    }
    hipLaunchKernelGGL(soft_update_kernel, dim3(grid1d(d.Pc)), blk, 0, s, target_params, critic_params, tau, d.Pc);

    // ---- (3) policy-gradient sample, temperature step                                              (:72-81)
    aact[0] = const_cast<float *>(state);
    if ((rc = forward(s, d.actor, actor_params, B, aact, agd))) return rc;
    hipLaunchKernelGGL(head_forward_kernel, rows_grid, blk, 0, s, aact[d.actor.n], eps_cur, seed, 2 * counter + 1, A, B, act_t, lp_cur,
                       eps_used, state, S, xa, variant);        // xa = [state | action_pg] for step (4)
    // obj_alpha = mean(alpha_log * (target_entropy - logprob)):  d/dalpha_log = target_entropy - mean(logprob)
    {
        const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        hipLaunchKernelGGL(alpha_step_kernel, dim3(1), blk, 0, s, lp_cur, B, target_entropy, g_alpha, alpha_log, alpha_m, alpha_v, beta1, beta2,
                           eps_adam, max_norm, (float)((double)lr / bc1), (float)sqrt(bc2));
    }

    if (!update_actor) {
        // AgentModSAC's two-time-scale rule skipped the actor this step (AgentSAC.py:152-158): alpha is clamped as always (:146-147),
        // obj_actor is nan (:158), actor / actor target / their Adam state stay as they are
        hipLaunchKernelGGL(clamp_alpha_kernel, dim3(1), dim3(64), 0, s, alpha_log);
        hipLaunchKernelGGL(fillk_kernel, dim3(1), blk, 0, s, objs_out + 1, __builtin_nanf(""), (int64_t)1);
        ERL_LAUNCH_CHECK("erl_sac_update_opt_f32");
    }
    // ---- (4) actor objective against the TARGET ensemble's mean, backward into the action, head, actor   (:82-85)
    if ((rc = critic_forward(s, d, target_params, B, xa, cw, true))) return rc;
    hipLaunchKernelGGL(actor_obj_kernel, dim3(1), blk, 0, s, cw.q, E, B, lp_cur, alpha_log, objs_out + 1);
    hipLaunchKernelGGL(fillk_kernel, dim3(grid1d((int64_t)E * B)), blk, 0, s, dq, -1.0f / ((float)E * (float)B), (int64_t)E * B);
    if (batched) {
        if ((rc = decoders_backward(s, d, target_params + d.enc.count, B, cw, dq, nullptr, dEnc, dEncE, enc_stride, tmpA, tmpB, tmp_stride)))
            return rc;
    } else {
        for (int e = 0; e < E; ++e)
            if ((rc = backward(s, d.dec, target_params + d.enc.count + (int64_t)e * d.dec.count, B, cw.act[e], cw.gd[e], dq + (size_t)e * B,
                               nullptr, cs_scr, dEnc, e > 0, tmpA, tmpB)))
                return rc;
    }
    if ((rc = dense_backward_input(s, dEnc, target_params, dxa, nullptr, false, (int)B, d.enc.d[1], S + A))) return rc;   // dL/d[state | action]
    // dL/daction = the action columns of dxa, read in place (row stride S + A)
    hipLaunchKernelGGL(head_backward_kernel, rows_grid, blk, 0, s, aact[d.actor.n], act_t, eps_used, dxa + S, S + A, alpha_log, A, B, dHead, variant);
    hipLaunchKernelGGL(clamp_alpha_kernel, dim3(1), dim3(64), 0, s, alpha_log);                  // after alpha was read (:80-81)
    if ((rc = backward(s, d.actor, actor_params, B, aact, agd, dHead, g_actor, cs_scr, nullptr, false, tmpA, tmpB))) return rc;
    {
        const int64_t off = 0, len = d.Pa;
        if ((rc = erl_clip_adam_f32(actor_params, g_actor, actor_m, actor_v, &off, &len, 1, nullptr, actor_step, lr, beta1, beta2, eps_adam,
                                    max_norm, 1.0f, stream)))
            return rc;
    }
    if (actor_target)      // AgentModSAC: soft_update(act_target, act, tau) after the actor's step (AgentSAC.py:156)
        hipLaunchKernelGGL(soft_update_kernel, dim3(grid1d(d.Pa)), blk, 0, s, actor_target, actor_params, tau, d.Pa);
    ERL_LAUNCH_CHECK("erl_sac_update_f32");
}

// The whole off-policy rollout of AgentBase._explore_vec_env (AgentBase.py:130-170) on the device-resident SynVecEnv as ONE launch
// (sac_fused.hip sac_rollout_synenv_kernel); needs erl_sac_rollout_synenv_supported(...)
extern "C" int erl_sac_rollout_synenv_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden, float *env_state,
                                          const float *Ws, const float *Wa, int32_t *step_count, int32_t *episode, int max_step,
                                          uint64_t env_seed, int64_t N, int64_t H, const float *noise, uint64_t seed, uint64_t counter0,
                                          float reward_scale, float *out_states, float *out_actions, float *out_rewards,
                                          uint8_t *out_undones, uint8_t *out_unmasks, float *out_last_state, void *stream)
{
    ERL_REQUIRE(actor_params && env_state && Ws && Wa && step_count && episode && out_states && out_actions && out_rewards && out_undones &&
                out_unmasks, "erl_sac_rollout_synenv_f32: NULL tensor");
    ERL_REQUIRE(erl_sac_rollout_synenv_supported(S, A, hidden, n_hidden, N), "erl_sac_rollout_synenv_f32: unsupported dims S=%d A=%d N=%lld (two "
                "hidden layers <= 256 wide in steps of 16, S + A <= 64, A <= 8, N <= 4096; ERL_SAC_FUSED=0 turns it off)", S, A, (long long)N);
    ERL_REQUIRE(H >= 1 && H < (1LL << 30) && max_step >= 1, "erl_sac_rollout_synenv_f32: bad H=%lld max_step=%d", (long long)H, max_step);
    SacDims d;
    ERL_REQUIRE(make_sac_dims(S, A, hidden, n_hidden, 1, &d), "erl_sac_rollout_synenv_f32: unsupported dims");
    const int64_t aoff[6] = {d.actor.oW[0], d.actor.ob[0], d.actor.oW[1], d.actor.ob[1], d.actor.oW[2], d.actor.ob[2]};
    return erl_sac_rollout_fused(actor_params, S, A, hidden[0], hidden[1], aoff, env_state, Ws, Wa, nullptr, step_count, episode, max_step, env_seed,
                                 N, H, noise, seed, counter0, reward_scale, out_states, out_actions, out_rewards, out_undones, out_unmasks,
                                 out_last_state, (hipStream_t)stream);
}

// the same loop on the device-resident PendulumVecEnv (S = 3, A = 1; phys: (N, 2) theta, theta_dot; obs: (N, 3) the live observation)
extern "C" int erl_sac_rollout_pendulum_f32(const float *actor_params, const int *hidden, int n_hidden, float *phys, float *obs,
                                            int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N, int64_t H,
                                            const float *noise, uint64_t seed, uint64_t counter0, float reward_scale, float *out_states,
                                            float *out_actions, float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks,
                                            float *out_last_state, void *stream)
{
    ERL_REQUIRE(actor_params && phys && obs && step_count && episode && out_states && out_actions && out_rewards && out_undones && out_unmasks,
                "erl_sac_rollout_pendulum_f32: NULL tensor");
    ERL_REQUIRE(erl_sac_rollout_synenv_supported(3, 1, hidden, n_hidden, N), "erl_sac_rollout_pendulum_f32: unsupported dims N=%lld (two hidden "
                "layers <= 256 wide in steps of 16, N <= 4096; ERL_SAC_FUSED=0 turns it off)", (long long)N);
    ERL_REQUIRE(H >= 1 && H < (1LL << 30) && max_step >= 1, "erl_sac_rollout_pendulum_f32: bad H=%lld max_step=%d", (long long)H, max_step);
    SacDims d;
    ERL_REQUIRE(make_sac_dims(3, 1, hidden, n_hidden, 1, &d), "erl_sac_rollout_pendulum_f32: unsupported dims");
    const int64_t aoff[6] = {d.actor.oW[0], d.actor.ob[0], d.actor.oW[1], d.actor.ob[1], d.actor.oW[2], d.actor.ob[2]};
    return erl_sac_rollout_fused(actor_params, 3, 1, hidden[0], hidden[1], aoff, obs, nullptr, nullptr, phys, step_count, episode, max_step, env_seed,
                                 N, H, noise, seed, counter0, reward_scale, out_states, out_actions, out_rewards, out_undones, out_unmasks,
                                 out_last_state, (hipStream_t)stream);
}

// ActorSAC.get_action for the off-policy rollout (AgentSAC.py:179-185): action = tanh(mean + std * eps); state_out (may be NULL):
// the rollout's `states[t] = state` (AgentBase.py:145) written by the same launch
static int sac_explore_impl(const float *actor_params, int S, int A, const int *hidden, int n_hidden, const float *state, int64_t N,
                            const float *noise, uint64_t seed, uint64_t counter, float *action_out, float *state_out, void *workspace,
                            int64_t workspace_bytes, int variant, void *stream);

extern "C" int erl_sac_explore_action_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden, const float *state,
                                          int64_t N, const float *noise, uint64_t seed, uint64_t counter, float *action_out,
                                          float *state_out, void *workspace, int64_t workspace_bytes, void *stream)
{
    return sac_explore_impl(actor_params, S, A, hidden, n_hidden, state, N, noise, seed, counter, action_out, state_out, workspace, workspace_bytes,
                            ERL_SAC_ACTOR_SAC, stream);
}

// ... with the actor variant named: ERL_SAC_ACTOR_FIX = ActorFixSAC.get_action (AgentSAC.py:217-224: raw last encoder layer, log_std
// clamped to [-20, 2]) for AgentModSAC's rollout
extern "C" int erl_sac_explore_action_opt_f32(const float *actor_params, int S, int A, const int *hidden, int n_hidden, const float *state,
                                              int64_t N, const float *noise, uint64_t seed, uint64_t counter, float *action_out,
                                              float *state_out, void *workspace, int64_t workspace_bytes, int actor_variant, void *stream)
{
    ERL_REQUIRE(actor_variant == ERL_SAC_ACTOR_SAC || actor_variant == ERL_SAC_ACTOR_FIX, "erl_sac_explore_action_opt_f32: unknown actor_variant %d",
                actor_variant);
    return sac_explore_impl(actor_params, S, A, hidden, n_hidden, state, N, noise, seed, counter, action_out, state_out, workspace, workspace_bytes,
                            actor_variant, stream);
}

static int sac_explore_impl(const float *actor_params, int S, int A, const int *hidden, int n_hidden, const float *state, int64_t N,
                            const float *noise, uint64_t seed, uint64_t counter, float *action_out, float *state_out, void *workspace,
                            int64_t workspace_bytes, int variant, void *stream)
{
    ERL_REQUIRE(actor_params && state && action_out && workspace, "erl_sac_explore_action_f32: NULL tensor");
    SacDims d;
    ERL_REQUIRE(make_sac_dims(S, A, hidden, n_hidden, 1, &d, variant), "erl_sac_explore_action_f32: unsupported dims");
    ERL_REQUIRE(N >= 1 && N < (1LL << 31), "erl_sac_explore_action_f32: bad N");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    Ws ws{(char *)workspace, 0, workspace_bytes};
    // two hidden layers up to 256 wide, N <= 4096 rows (config 3: 64 envs): one launch, the fused step's actor kernel (sac_fused.hip);
    // ERL_SAC_FUSED=0 keeps the layered form below
    static const bool fused_on = [] { const char *e = getenv("ERL_SAC_FUSED"); return !(e && atoi(e) == 0); }();
    if (fused_on && variant == ERL_SAC_ACTOR_SAC && erl_sac_fused_supported(S, A, hidden, n_hidden, 1, N)) {
        float *lp_s = ws.take(N);
        ERL_REQUIRE(lp_s != nullptr, "erl_sac_explore_action_f32: workspace too small");
        const int64_t aoff[6] = {d.actor.oW[0], d.actor.ob[0], d.actor.oW[1], d.actor.ob[1], d.actor.oW[2], d.actor.ob[2]};
        return erl_sac_explore_fused(actor_params, S, A, hidden[0], hidden[1], aoff, state, N, noise, seed, counter, action_out, state_out, lp_s, s);
    }
    float *aact[MAXL + 2];
    aact[0] = const_cast<float *>(state);
    for (int l = 1; l <= d.actor.n; ++l) aact[l] = ws.take(N * d.actor.d[l]);
    float *lp = ws.take(N);
    ERL_REQUIRE(lp != nullptr, "erl_sac_explore_action_f32: workspace too small");
    if (state_out && (rc = erl_hip_status(hipMemcpyAsync(state_out, state, (size_t)N * S * sizeof(float), hipMemcpyDeviceToDevice, s),
                                          "erl_sac_explore_action_f32: hipMemcpyAsync(state row)")))
        return rc;
    if ((rc = forward(s, d.actor, actor_params, N, aact, nullptr))) return rc;
    hipLaunchKernelGGL(head_forward_kernel, dim3((unsigned)erl_cdiv(N, 256)), dim3(256), 0, s, aact[d.actor.n], noise, seed, counter, A, N,
                       action_out, lp, (float *)nullptr, (const float *)nullptr, S, (float *)nullptr, variant);
    ERL_LAUNCH_CHECK("erl_sac_explore_action_f32");
}
