// Generic-shape MLP path ("mlpn"): any number of hidden layers, any widths.  gfx950.
//
// The fused register-chained kernels (ppo_step.hip, mlp.hip) cover 2 hidden layers of width <= 128 -- the reference's
// default Config.net_dims = [128, 128] and its Pendulum demos.  Reference demos also use (256, 128), (256, 128, 64),
// (256, 128, 128) (examples/demo_A2C_PPO.py:117,171,224): those shapes take this layered path, so that
// AgentPPO.explore_env / update_net stay drop-in for every build_mlp() the reference can construct
// (elegantrl/agents/AgentBase.py:345-360).  Dense layers are plain library GEMMs (rocBLAS sgemm, fp32, atomics
// off => deterministic); everything around them is hand-written HIP: gather + normalise, bias + exact-erf GELU (+ its
// derivative), the PPO objective with its analytic dL/dY, Gaussian sampling / log-prob for the rollout.
// Activations live in a caller-provided workspace (row-major [rows][width]).
//
// Parameter block (one flat fp32 buffer per network, same convention as the fused kernels):
//   W1[d1][d0] b1[d1] ... WL[dL][dL-1] bL[dL] Wout[out][dL] bout[out] (+ action_std_log[out] for the actor)
#include <rocblas/rocblas.h>

#include "mlp_chain.h"

namespace {

constexpr int MAXL = ERL_MAX_LAYERS;      // hidden layers
constexpr float kLogSqrt2PiN = 0.91893853320467274178f;

struct NetDims {
    int n;                 // number of dense layers = hidden + 1
    int d[MAXL + 2];       // d[0] = S, d[1..n-1] hidden, d[n] = out
    int64_t oW[MAXL + 1], ob[MAXL + 1], oStd, count;
};

bool make_dims(const int *dims, int n_dims, bool with_std, NetDims *nd)
{
    if (!dims || n_dims < 2 || n_dims > MAXL + 2) return false;
    nd->n = n_dims - 1;
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

rocblas_handle g_handle = nullptr;

int blas(hipStream_t stream, rocblas_handle *h)
{
    if (!g_handle) {
        if (rocblas_create_handle(&g_handle) != rocblas_status_success) {
            erl_set_error("rocblas_create_handle failed");
            return -2;
        }
        rocblas_set_atomics_mode(g_handle, rocblas_atomics_not_allowed);   // deterministic reductions
        rocblas_set_pointer_mode(g_handle, rocblas_pointer_mode_host);
    }
    if (rocblas_set_stream(g_handle, stream) != rocblas_status_success) {
        erl_set_error("rocblas_set_stream failed");
        return -2;
    }
    *h = g_handle;
    return 0;
}

#define RB(call)                                                           \
    do {                                                                   \
        rocblas_status st_ = (call);                                       \
        if (st_ != rocblas_status_success) {                               \
            erl_set_error("%s -> rocblas status %d", #call, (int)st_);     \
            return -2;                                                     \
        }                                                                  \
    } while (0)

// row-major Z[M][N] = X[M][K] . W[N][K]^T
int gemm_fwd(rocblas_handle h, const float *X, const float *W, float *Z, int M, int N, int K)
{
    const float one = 1.f, zero = 0.f;
    RB(rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, W, K, X, K, &zero, Z, N));
    return 0;
}
// row-major dX[M][K] = dZ[M][N] . W[N][K]
int gemm_dx(rocblas_handle h, const float *dZ, const float *W, float *dX, int M, int N, int K)
{
    const float one = 1.f, zero = 0.f;
    RB(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, K, M, N, &one, W, K, dZ, N, &zero, dX, K));
    return 0;
}
// row-major dW[N][K] = dZ[M][N]^T . X[M][K]
int gemm_dw(rocblas_handle h, const float *dZ, const float *X, float *dW, int M, int N, int K)
{
    const float one = 1.f, zero = 0.f;
    RB(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, K, N, M, &one, X, K, dZ, N, &zero, dW, K));
    return 0;
}
// db[N] = column sums of dZ[M][N]
int colsum(rocblas_handle h, const float *dZ, const float *ones, float *db, int M, int N)
{
    const float one = 1.f, zero = 0.f;
    RB(rocblas_sgemv(h, rocblas_operation_none, N, M, &one, dZ, N, ones, 1, &zero, db, 1));
    return 0;
}

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

// in place: Z <- GELU(Z + b); optionally G <- GELU'(Z + b)
__global__ __launch_bounds__(256) void bias_gelu_kernel(float *__restrict__ Z, float *__restrict__ G, const float *__restrict__ bias,
                                                        int width, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % width);
        float y, gd;
        gelu_and_grad_fast(Z[e] + bias[c], y, gd);
        Z[e] = y;
        if (G) G[e] = gd;
    }
}

__global__ __launch_bounds__(256) void bias_kernel(float *__restrict__ Z, const float *__restrict__ bias, int width, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) Z[e] += bias[e % width];
}

__global__ __launch_bounds__(256) void mul_kernel(float *__restrict__ dH, const float *__restrict__ G, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) dH[e] *= G[e];
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
                                                        float ratio_clip, float lambda_entropy, float inv_batch,
                                                        float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l0 = 0.f, l1 = 0.f;
    if (b < B) {
        const int64_t id = ids[b];
        const int64_t n = id / H, t = id - n * H;
        const int64_t row = t * N + n;
        const float um = unmasks[row] ? 1.f : 0.f;
        if (!ACTOR) {
            const float diff = Y[b] - xa_src[row];
            l0 = diff * diff * um;
            Y[b] = 2.f * diff * um * inv_batch;
        } else {
            float lp = 0.f;
            for (int a = 0; a < A; ++a) {
                const float sdv = expf(std_log[a]), var = sdv * sdv;
                const float diff = actions[row * A + a] - Y[b * A + a];
                lp += -(diff * diff) / (2.f * var) - logf(sdv) - kLogSqrt2PiN;
            }
            const float adv = xb_src[row];
            const float ratio = expf(lp - xa_src[row]);
            const float w = adv > 0.f ? 1.f - ratio_clip : 1.f + ratio_clip;
            const float surr = adv * ratio * w;
            l0 = surr * um;
            l1 = um;
            const float dlp = -(surr * um) * inv_batch;
            const float ent_term = lambda_entropy * um * inv_batch;
            for (int a = 0; a < A; ++a) {
                const float sdv = expf(std_log[a]), var = sdv * sdv;
                const float diff = actions[row * A + a] - Y[b * A + a];
                Y[b * A + a] = dlp * (diff / var);
                DSL[b * A + a] = dlp * (diff * diff / var - 1.f) + ent_term;
            }
        }
    }
    const float t0 = block_sum(l0, red), t1 = block_sum(l1, red);
    if (threadIdx.x == 0) {
        part[(size_t)blockIdx.x * 2 + 0] = t0;
        part[(size_t)blockIdx.x * 2 + 1] = t1;
    }
}

// logs: fold the per-block partials in a fixed order
__global__ void fold_logs_kernel(const float *__restrict__ part, int nparts, const float *__restrict__ std_log, int A, float inv_batch,
                                 int is_actor, float *__restrict__ logs)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < nparts; ++i) {
        s0 += part[2 * i];
        s1 += part[2 * i + 1];
    }
    if (is_actor) {
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

// forward pass into workspace buffers; act[l] = activation after layer l (act[0] = X), keep_g: store GELU' per hidden layer
int forward(rocblas_handle h, hipStream_t s, const NetDims &nd, const float *P, int64_t rows, float **act, float **gd)
{
    for (int l = 0; l < nd.n; ++l) {
        const int K = nd.d[l], Nw = nd.d[l + 1];
        int rc = gemm_fwd(h, act[l], P + nd.oW[l], act[l + 1], (int)rows, Nw, K);
        if (rc) return rc;
        const int64_t total = rows * Nw;
        if (l + 1 < nd.n)
            hipLaunchKernelGGL(bias_gelu_kernel, dim3(grid1d(total)), dim3(256), 0, s, act[l + 1], gd ? gd[l + 1] : nullptr, P + nd.ob[l],
                               Nw, total);
        else
            hipLaunchKernelGGL(bias_kernel, dim3(grid1d(total)), dim3(256), 0, s, act[l + 1], P + nd.ob[l], Nw, total);
    }
    return 0;
}

}  // namespace

extern "C" int64_t erl_mlpn_param_count(const int *dims, int n_dims, int with_std_log)
{
    NetDims nd;
    if (!make_dims(dims, n_dims, with_std_log != 0, &nd)) return -1;
    return nd.count;
}

extern "C" int64_t erl_mlpn_workspace_bytes(const int *dims, int n_dims, int64_t rows, int training)
{
    NetDims nd;
    if (!make_dims(dims, n_dims, false, &nd) || rows < 1) return -1;
    int64_t f = ws_floats_forward(nd, rows);
    if (training) {
        int maxd = 1;
        for (int l = 0; l <= nd.n; ++l) maxd = nd.d[l] > maxd ? nd.d[l] : maxd;
        for (int l = 1; l < nd.n; ++l) f += rows * nd.d[l] + 64;      // GELU'
        f += 2 * (rows * maxd + 64);                                  // dH ping-pong
        f += rows * nd.d[nd.n] + 64;                                  // DSL
        f += rows + 64;                                               // ones
        f += 2 * (erl_cdiv(rows, 256) + 64);                          // loss partials
    }
    return f * 4 + 4096;
}

extern "C" int erl_mlpn_value_forward_f32(const float *params, const float *state_avg, const float *state_std, const int *dims,
                                          int n_dims, const float *states, int64_t rows, float *values, void *workspace,
                                          int64_t workspace_bytes, void *stream)
{
    NetDims nd;
    ERL_REQUIRE(params && state_avg && state_std && states && values && workspace, "erl_mlpn_value_forward_f32: NULL tensor");
    ERL_REQUIRE(make_dims(dims, n_dims, false, &nd) && nd.d[nd.n] == 1, "erl_mlpn_value_forward_f32: bad dims");
    if (rows == 0) return ERL_OK;
    ERL_REQUIRE(rows > 0 && rows < (1LL << 31), "erl_mlpn_value_forward_f32: bad rows");
    hipStream_t s = (hipStream_t)stream;
    rocblas_handle h;
    int rc = blas(s, &h);
    if (rc) return rc;
    Ws ws{(char *)workspace, 0, workspace_bytes};
    float *act[MAXL + 2];
    for (int l = 0; l < nd.n; ++l) act[l] = ws.take(rows * nd.d[l]);
    act[nd.n] = values;   // the (rows, 1) output lands directly in the caller's tensor
    ERL_REQUIRE(act[nd.n - 1] != nullptr, "erl_mlpn_value_forward_f32: workspace too small");
    hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(rows * nd.d[0])), dim3(256), 0, s, states, state_avg, state_std,
                       (const int64_t *)nullptr, (int64_t)1, (int64_t)1, nd.d[0], rows, act[0], (float *)nullptr);
    if ((rc = forward(h, s, nd, params, rows, act, nullptr))) return rc;
    ERL_LAUNCH_CHECK("erl_mlpn_value_forward_f32");
}

extern "C" int erl_mlpn_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std, const int *dims,
                                         int n_dims, const float *state, int64_t N, const float *noise, uint64_t seed,
                                         uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                                         float *out_action_env, void *workspace, int64_t workspace_bytes, void *stream)
{
    NetDims nd;
    ERL_REQUIRE(actor_params && state_avg && state_std && state && workspace, "erl_mlpn_rollout_step_f32: NULL tensor");
    ERL_REQUIRE(make_dims(dims, n_dims, true, &nd), "erl_mlpn_rollout_step_f32: bad dims");
    ERL_REQUIRE(N >= 1 && N < (1LL << 31), "erl_mlpn_rollout_step_f32: bad N");
    hipStream_t s = (hipStream_t)stream;
    rocblas_handle h;
    int rc = blas(s, &h);
    if (rc) return rc;
    Ws ws{(char *)workspace, 0, workspace_bytes};
    float *act[MAXL + 2];
    for (int l = 0; l <= nd.n; ++l) act[l] = ws.take(N * nd.d[l]);
    ERL_REQUIRE(act[nd.n] != nullptr, "erl_mlpn_rollout_step_f32: workspace too small");
    hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(N * nd.d[0])), dim3(256), 0, s, state, state_avg, state_std,
                       (const int64_t *)nullptr, (int64_t)1, (int64_t)1, nd.d[0], N, act[0], out_state_row);
    if ((rc = forward(h, s, nd, actor_params, N, act, nullptr))) return rc;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)erl_cdiv(N, 256)), dim3(256), 0, s, act[nd.n], actor_params + nd.oStd, nd.d[nd.n], N,
                       noise, seed, counter, out_action_row, out_logprob_row, out_action_env);
    ERL_LAUNCH_CHECK("erl_mlpn_rollout_step_f32");
}

// One PPO minibatch for networks of any depth: writes the summed gradient [actor | critic | logs(4)] to flat_grad.
extern "C" int erl_mlpn_ppo_step_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                                     const float *cri_avg, const float *cri_std, const int *actor_dims, int n_dims,
                                     const float *states, const float *actions, const uint8_t *unmasks, const float *logprobs,
                                     const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                                     int64_t B, float ratio_clip, float lambda_entropy, float inv_batch, float *flat_grad,
                                     void *workspace, int64_t workspace_bytes, void *stream)
{
    ERL_REQUIRE(actor_params && critic_params && act_avg && act_std && cri_avg && cri_std && states && actions && unmasks &&
                    logprobs && advantages && reward_sums && ids && flat_grad && workspace,
                "erl_mlpn_ppo_step_f32: NULL tensor");
    NetDims na, nc;
    ERL_REQUIRE(make_dims(actor_dims, n_dims, true, &na), "erl_mlpn_ppo_step_f32: bad dims");
    int cdims[MAXL + 2];
    for (int i = 0; i < n_dims; ++i) cdims[i] = actor_dims[i];
    cdims[n_dims - 1] = 1;
    ERL_REQUIRE(make_dims(cdims, n_dims, false, &nc), "erl_mlpn_ppo_step_f32: bad dims");
    ERL_REQUIRE(H >= 1 && N >= 1 && B >= 1 && B < (1LL << 31), "erl_mlpn_ppo_step_f32: bad shape");
    hipStream_t s = (hipStream_t)stream;
    rocblas_handle h;
    int rc = blas(s, &h);
    if (rc) return rc;
    const int A = na.d[na.n];
    float *logs = flat_grad + na.count + nc.count;

    for (int net = 0; net < 2; ++net) {
        const NetDims &nd = net == 0 ? na : nc;
        const float *P = net == 0 ? actor_params : critic_params;
        float *G = flat_grad + (net == 0 ? 0 : na.count);
        Ws ws{(char *)workspace, 0, workspace_bytes};
        float *act[MAXL + 2], *gd[MAXL + 2];
        int maxd = 1;
        for (int l = 0; l <= nd.n; ++l) {
            act[l] = ws.take(B * nd.d[l]);
            maxd = nd.d[l] > maxd ? nd.d[l] : maxd;
        }
        gd[0] = gd[nd.n] = nullptr;
        for (int l = 1; l < nd.n; ++l) gd[l] = ws.take(B * nd.d[l]);
        float *dA = ws.take(B * maxd), *dB = ws.take(B * maxd);
        float *dsl = ws.take(B * nd.d[nd.n]);
        float *ones = ws.take(B);
        const int nparts = (int)erl_cdiv(B, 256);
        float *part = ws.take(2 * (int64_t)nparts);
        ERL_REQUIRE(part != nullptr, "erl_mlpn_ppo_step_f32: workspace too small (need erl_mlpn_workspace_bytes(dims, rows = B, training = 1))");

        hipLaunchKernelGGL(fill_kernel, dim3(grid1d(B)), dim3(256), 0, s, ones, 1.0f, B);
        hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(B * nd.d[0])), dim3(256), 0, s, states, net == 0 ? act_avg : cri_avg,
                           net == 0 ? act_std : cri_std, ids, H, N, nd.d[0], B, act[0], (float *)nullptr);
        if ((rc = forward(h, s, nd, P, B, act, gd))) return rc;
        float *Y = act[nd.n];
        if (net == 0)
            hipLaunchKernelGGL((objective_kernel<true>), dim3(nparts), dim3(256), 0, s, Y, dsl, ids, H, N, A, B, actions, unmasks, logprobs,
                               advantages, P + nd.oStd, ratio_clip, lambda_entropy, inv_batch, part);
        else
            hipLaunchKernelGGL((objective_kernel<false>), dim3(nparts), dim3(256), 0, s, Y, (float *)nullptr, ids, H, N, 1, B, actions,
                               unmasks, reward_sums, (const float *)nullptr, (const float *)nullptr, ratio_clip, lambda_entropy,
                               inv_batch, part);
        hipLaunchKernelGGL(fold_logs_kernel, dim3(1), dim3(64), 0, s, part, nparts, P + nd.oStd, A, inv_batch, net == 0 ? 1 : 0, logs);
        if (net == 0 && (rc = colsum(h, dsl, ones, G + nd.oStd, (int)B, A))) return rc;     // dL/dstd_log

        // backward: dZ of the output layer is Y (dL/dY); walk the layers down
        const float *dZ = Y;
        for (int l = nd.n - 1; l >= 0; --l) {
            const int K = nd.d[l], Nw = nd.d[l + 1];
            if ((rc = gemm_dw(h, dZ, act[l], G + nd.oW[l], (int)B, Nw, K))) return rc;
            if ((rc = colsum(h, dZ, ones, G + nd.ob[l], (int)B, Nw))) return rc;
            if (l > 0) {
                float *dH = (dZ == dA) ? dB : dA;
                if ((rc = gemm_dx(h, dZ, P + nd.oW[l], dH, (int)B, Nw, K))) return rc;
                hipLaunchKernelGGL(mul_kernel, dim3(grid1d(B * K)), dim3(256), 0, s, dH, gd[l], B * K);
                dZ = dH;
            }
        }
    }
    ERL_LAUNCH_CHECK("erl_mlpn_ppo_step_f32");
}
