// Generic-shape MLP path ("mlpn"): any number of hidden layers, any widths.  gfx950.
//
// The fused register-chained kernels (ppo_step.hip, mlp.hip) cover 2 hidden layers of width <= 128 -- the reference's
// default Config.net_dims = [128, 128] and its Pendulum demos.  Reference demos also use (256, 128), (256, 128, 64),
// (256, 128, 128) (examples/demo_A2C_PPO.py:117,171,224): those shapes take this layered path, so that
// AgentPPO.explore_env / update_net stay drop-in for every build_mlp() the reference can construct
// (elegantrl/agents/AgentBase.py:345-360).  Dense layers are the fp32 MFMA GEMMs of gemm_tiles.h (bias, exact-erf
// GELU and its derivative, the backward gate and the bias gradient are their epilogues; fixed-order reductions =>
// deterministic); around them: gather + normalise, the PPO objective with its analytic dL/dY, sampling / log-prob.
// Activations live in a caller-provided workspace (row-major [rows][width]).
//
// Parameter block (one flat fp32 buffer per network, same convention as the fused kernels):
//   W1[d1][d0] b1[d1] ... WL[dL][dL-1] bL[dL] Wout[out][dL] bout[out] (+ action_std_log[out] for the actor)
#include "mlpn_common.h"
#include "ppo_step_wd.h"

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
        f += colsum_scratch_floats(rows, maxd) + 64;                  // bias-gradient partials
        f += (2 + nd.d[nd.n]) * (erl_cdiv(rows, 256) + 64);          // per-block partials: logged values, dL/dstd_log
        int64_t nk = 1;
        for (int l = 0; l < nd.n; ++l) nk = (int64_t)nd.d[l] * nd.d[l + 1] > nk ? (int64_t)nd.d[l] * nd.d[l + 1] : nk;
        f += dw_scratch_floats(rows, nk) + 64;                        // per-chunk dW partials
        f = 2 * f + 128;                                              // actor and critic each own a region (they run on two streams)
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
    int rc;
    // net_dims = (256, h2) / (256, h2, h3): ONE launch (rollout_wide.hip value_wide_kernel; ERL_WIDE_FUSED=0 keeps the layered launches below)
    if (erl_value_wide_supported(dims, n_dims) && (reinterpret_cast<uintptr_t>(params) & 15) == 0)
        return erl_value_wide_forward(params, state_avg, state_std, dims, n_dims, states, rows, values, s);
    Ws ws{(char *)workspace, 0, workspace_bytes};
    float *act[MAXL + 2];
    for (int l = 0; l < nd.n; ++l) act[l] = ws.take(rows * nd.d[l]);
    act[nd.n] = values;   // the (rows, 1) output lands directly in the caller's tensor
    ERL_REQUIRE(act[nd.n - 1] != nullptr, "erl_mlpn_value_forward_f32: workspace too small");
    hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(rows * nd.d[0])), dim3(256), 0, s, states, state_avg, state_std,
                       (const int64_t *)nullptr, (int64_t)1, (int64_t)1, nd.d[0], rows, act[0], (float *)nullptr);
    if ((rc = forward(s, nd, params, rows, act, nullptr))) return rc;
    ERL_LAUNCH_CHECK("erl_mlpn_value_forward_f32");
}

namespace {

// shared by the Gaussian and the categorical rollout step: normalise (+ store the raw state row), forward, sample
int rollout_impl(const char *what, bool discrete, const float *actor_params, const float *state_avg, const float *state_std,
                 const int *dims, int n_dims, const float *state, int64_t N, const float *noise, uint64_t seed, uint64_t counter,
                 float *out_state_row, void *out_action_row, float *out_logprob_row, void *out_action_env, void *workspace,
                 int64_t workspace_bytes, void *stream)
{
    NetDims nd;
    ERL_REQUIRE(actor_params && state_avg && state_std && state && workspace, "%s: NULL tensor", what);
    ERL_REQUIRE(make_dims(dims, n_dims, !discrete, &nd), "%s: bad dims", what);
    ERL_REQUIRE(!discrete || nd.d[nd.n] <= kMaxDiscrete, "%s: action_dim > %d", what, kMaxDiscrete);
    ERL_REQUIRE(N >= 1 && N < (1LL << 31), "%s: bad N", what);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // net_dims = (256, h2) / (256, h2, h3), a few thousand envs: ONE launch (rollout_wide.hip; ERL_WIDE_FUSED=0 keeps the layered launches below)
    if (!discrete && erl_rollout_wide_supported(dims, n_dims, N) && (reinterpret_cast<uintptr_t>(actor_params) & 15) == 0)
        return erl_rollout_wide_step(actor_params, state_avg, state_std, dims, n_dims, state, N, noise, seed, counter, out_state_row,
                                     (float *)out_action_row, out_logprob_row, (float *)out_action_env, s);
    Ws ws{(char *)workspace, 0, workspace_bytes};
    float *act[MAXL + 2];
    for (int l = 0; l <= nd.n; ++l) act[l] = ws.take(N * nd.d[l]);
    ERL_REQUIRE(act[nd.n] != nullptr, "%s: workspace too small", what);
    hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(N * nd.d[0])), dim3(256), 0, s, state, state_avg, state_std,
                       (const int64_t *)nullptr, (int64_t)1, (int64_t)1, nd.d[0], N, act[0], out_state_row);
    if ((rc = forward(s, nd, actor_params, N, act, nullptr))) return rc;
    if (discrete)
        hipLaunchKernelGGL(sample_categorical_kernel, dim3((unsigned)erl_cdiv(N, 256)), dim3(256), 0, s, act[nd.n], nd.d[nd.n], N, noise,
                           seed, counter, (int32_t *)out_action_row, out_logprob_row, (int64_t *)out_action_env);
    else
        hipLaunchKernelGGL(sample_kernel, dim3((unsigned)erl_cdiv(N, 256)), dim3(256), 0, s, act[nd.n], actor_params + nd.oStd,
                           nd.d[nd.n], N, noise, seed, counter, (float *)out_action_row, out_logprob_row, (float *)out_action_env);
    return erl_hip_status(hipGetLastError(), what);
}

}  // namespace

extern "C" int erl_mlpn_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std, const int *dims,
                                         int n_dims, const float *state, int64_t N, const float *noise, uint64_t seed,
                                         uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                                         float *out_action_env, void *workspace, int64_t workspace_bytes, void *stream)
{
    return rollout_impl("erl_mlpn_rollout_step_f32", false, actor_params, state_avg, state_std, dims, n_dims, state, N, noise, seed,
                        counter, out_state_row, out_action_row, out_logprob_row, out_action_env, workspace, workspace_bytes, stream);
}

extern "C" int erl_mlpn_rollout_step_discrete_f32(const float *actor_params, const float *state_avg, const float *state_std,
                                                  const int *dims, int n_dims, const float *state, int64_t N, const float *uniform,
                                                  uint64_t seed, uint64_t counter, float *out_state_row, int32_t *out_action_row,
                                                  float *out_logprob_row, int64_t *out_action_env, void *workspace,
                                                  int64_t workspace_bytes, void *stream)
{
    return rollout_impl("erl_mlpn_rollout_step_discrete_f32", true, actor_params, state_avg, state_std, dims, n_dims, state, N, uniform,
                        seed, counter, out_state_row, out_action_row, out_logprob_row, out_action_env, workspace, workspace_bytes,
                        stream);
}

// One PPO minibatch for networks of any depth: writes the summed gradient [actor | critic | logs(4)] to flat_grad.
namespace {

// a library-owned second stream per (device, caller stream) (non-blocking) with the two events that fork it from / join it into
// the caller's stream; NULL when ERL_MLPN_STREAMS=1 or the runtime refuses (then everything stays on the caller's stream)
struct SideStream {
    int device = -1;
    hipStream_t owner = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream g_side[16];

SideStream *side_stream(hipStream_t owner)
{
    static const bool off = [] { const char *e = getenv("ERL_MLPN_STREAMS"); return e && atoi(e) == 1; }();
    int dev = -1;
    if (off || hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    for (auto &q : g_side)
        if (q.stream && q.device == dev && q.owner == owner) return &q;
    for (auto &q : g_side) {
        if (q.stream) continue;
        if (hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            q.stream = nullptr;
            return nullptr;
        }
        q.device = dev;
        q.owner = owner;
        return &q;
    }
    return nullptr;
}

// every exit path after the fork joins the side stream back into the caller's
struct SideJoin {
    SideStream *side;
    hipStream_t s;
    bool armed = false;
    ~SideJoin()
    {
        if (armed && side) {
            (void)hipEventRecord(side->join, side->stream);
            (void)hipStreamWaitEvent(s, side->join, 0);
        }
    }
};

int ppo_step_impl(const char *what, bool discrete, const float *actor_params, const float *critic_params, const float *act_avg,
                  const float *act_std, const float *cri_avg, const float *cri_std, const int *actor_dims, int n_dims,
                  const float *states, const void *actions_any, const uint8_t *unmasks, const float *logprobs,
                  const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B,
                  float ratio_clip, float lambda_entropy, float inv_batch, int objective, float *flat_grad, void *workspace,
                  int64_t workspace_bytes, void *stream)
{
    const float *actions = discrete ? nullptr : (const float *)actions_any;
    const int32_t *actions_i = discrete ? (const int32_t *)actions_any : nullptr;
    ERL_REQUIRE(actor_params && critic_params && act_avg && act_std && cri_avg && cri_std && states && actions_any && unmasks &&
                    logprobs && advantages && reward_sums && ids && flat_grad && workspace,
                "%s: NULL tensor", what);
    NetDims na, nc;
    ERL_REQUIRE(make_dims(actor_dims, n_dims, !discrete, &na), "%s: bad dims", what);
    ERL_REQUIRE(!discrete || na.d[na.n] <= kMaxDiscrete, "%s: action_dim > %d", what, kMaxDiscrete);
    int cdims[MAXL + 2];
    for (int i = 0; i < n_dims; ++i) cdims[i] = actor_dims[i];
    cdims[n_dims - 1] = 1;
    ERL_REQUIRE(make_dims(cdims, n_dims, false, &nc), "%s: bad dims", what);
    ERL_REQUIRE(H >= 1 && N >= 1 && B >= 1 && B < (1LL << 31), "%s: bad shape", what);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const int A = na.d[na.n];
    float *logs = flat_grad + na.count + nc.count;

    // The two networks are independent until the optimiser, and each one's ~18 launches are a chain of DEPENDENT launches at
    // ~7 us apiece whatever they compute (profiles/r02_gemm_small_scaling.txt): the critic's chain runs on a library-owned
    // side stream next to the actor's (forked from / joined back into the caller's stream by events), each network in its own
    // half of the workspace.  The host enqueues the two chains in two alternating stages so that neither waits long for the
    // interpreter-free but still finite (~4 us per launch) enqueue of the other.  ERL_MLPN_STREAMS=1 keeps everything on the
    // caller's stream.
    hipStream_t s1 = s;
    SideStream *side = side_stream(s);
    SideJoin joiner{side, s};
    if (side) {
        if ((rc = erl_hip_status(hipEventRecord(side->fork, s), "hipEventRecord(fork)"))) return rc;
        if ((rc = erl_hip_status(hipStreamWaitEvent(side->stream, side->fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
        s1 = side->stream;
        joiner.armed = true;
    }
    const int64_t half = (workspace_bytes / 2) & ~(int64_t)255;
    struct NetState {
        float *act[MAXL + 2], *gd[MAXL + 2];
        float *dA, *dB, *dsl, *cs_scr, *part, *dw_scr;
    } st[2];
    const int nparts = (int)erl_cdiv(B, 256);
    for (int net = 0; net < 2; ++net) {
        const NetDims &nd = net == 0 ? na : nc;
        NetState &q = st[net];
        Ws ws{(char *)workspace + net * half, 0, half};
        int maxd = 1;
        for (int l = 0; l <= nd.n; ++l) {
            q.act[l] = ws.take(B * nd.d[l]);
            maxd = nd.d[l] > maxd ? nd.d[l] : maxd;
        }
        q.gd[0] = q.gd[nd.n] = nullptr;
        for (int l = 1; l < nd.n; ++l) q.gd[l] = ws.take(B * nd.d[l]);
        q.dA = ws.take(B * maxd);
        q.dB = ws.take(B * maxd);
        q.dsl = ws.take(B * nd.d[nd.n]);
        q.cs_scr = ws.take(colsum_scratch_floats(B, maxd));
        q.part = ws.take((2 + (int64_t)nd.d[nd.n]) * nparts);
        int64_t nk = 1;
        for (int l = 0; l < nd.n; ++l) nk = (int64_t)nd.d[l] * nd.d[l + 1] > nk ? (int64_t)nd.d[l] * nd.d[l + 1] : nk;
        q.dw_scr = dw_scratch_floats(B, nk) ? ws.take(dw_scratch_floats(B, nk)) : nullptr;
        ERL_REQUIRE(q.part != nullptr && (q.dw_scr != nullptr || !dw_scratch_floats(B, nk)),
                    "%s: workspace too small (need erl_mlpn_workspace_bytes(dims, rows = B, training = 1))", what);
    }
    for (int stage = 0; stage < 2; ++stage) {
        for (int net = 0; net < 2; ++net) {
            const NetDims &nd = net == 0 ? na : nc;
            const float *P = net == 0 ? actor_params : critic_params;
            float *G = flat_grad + (net == 0 ? 0 : na.count);
            NetState &q = st[net];
            hipStream_t sn = net == 0 ? s : s1;
            float *Y = q.act[nd.n];
            if (stage == 0) {          // gather + normalise, forward, objective and dL/dY
                hipLaunchKernelGGL(gather_norm_kernel, dim3(grid1d(B * nd.d[0])), dim3(256), 0, sn, states, net == 0 ? act_avg : cri_avg,
                                   net == 0 ? act_std : cri_std, ids, H, N, nd.d[0], B, q.act[0], (float *)nullptr);
                if ((rc = forward(sn, nd, P, B, q.act, q.gd))) return rc;
                if (net == 0 && discrete)
                    hipLaunchKernelGGL(objective_discrete_kernel, dim3(nparts), dim3(256), 0, sn, Y, ids, H, N, A, B, actions_i, unmasks,
                                       logprobs, advantages, ratio_clip, lambda_entropy, inv_batch, q.part);
                else if (net == 0)
                    hipLaunchKernelGGL((objective_kernel<true>), dim3(nparts), dim3(256), 0, sn, Y, (float *)nullptr, ids, H, N, A, B, actions, unmasks,
                                       logprobs, advantages, P + nd.oStd, ratio_clip, lambda_entropy, inv_batch, objective, q.part);
                else
                    hipLaunchKernelGGL((objective_kernel<false>), dim3(nparts), dim3(256), 0, sn, Y, (float *)nullptr, ids, H, N, 1, B, actions,
                                       unmasks, reward_sums, (const float *)nullptr, (const float *)nullptr, ratio_clip, lambda_entropy,
                                       inv_batch, objective, q.part);
                const bool gauss = net == 0 && !discrete;
                hipLaunchKernelGGL(fold_logs_kernel, dim3(1), dim3(256), 0, sn, q.part, nparts, gauss ? 2 + A : 2, P + nd.oStd, A, inv_batch,
                                   net == 0 ? (discrete ? 2 : 1) : 0, logs, gauss ? G + nd.oStd : (float *)nullptr);   // + dL/dstd_log
            } else {                   // backward: dZ of the output layer is Y (dL/dY); walk the layers down
                if ((rc = backward(sn, nd, P, B, q.act, q.gd, Y, G, q.cs_scr, nullptr, false, q.dA, q.dB, q.dw_scr))) return rc;
            }
        }
    }
    if (side) {
        if ((rc = erl_hip_status(hipEventRecord(side->join, s1), "hipEventRecord(join)"))) return rc;
        if ((rc = erl_hip_status(hipStreamWaitEvent(s, side->join, 0), "hipStreamWaitEvent(join)"))) return rc;
        joiner.armed = false;
    }
    return erl_hip_status(hipGetLastError(), what);
}

}  // namespace

extern "C" int erl_mlpn_ppo_step_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                                     const float *cri_avg, const float *cri_std, const int *actor_dims, int n_dims,
                                     const float *states, const float *actions, const uint8_t *unmasks, const float *logprobs,
                                     const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                                     int64_t B, float ratio_clip, float lambda_entropy, float inv_batch, int objective,
                                     float *flat_grad, void *workspace, int64_t workspace_bytes, void *stream)
{
    ERL_REQUIRE(objective >= ERL_PPO_OBJ_REFERENCE && objective <= ERL_PPO_OBJ_A2C, "erl_mlpn_ppo_step_f32: unknown objective %d",
                objective);
    // net_dims (256, 128, 64 | 128), the reference's BipedalWalker / Humanoid demo networks: one fused kernel (ppo_step_wd_impl.h)
    if (actor_dims && erl_ppo_wd3_supported(actor_dims, n_dims) && actor_params && critic_params && act_avg && act_std && cri_avg && cri_std &&
        states && actions && unmasks && logprobs && advantages && reward_sums && ids && flat_grad && H >= 1 && N >= 1 && B >= 1)
        return erl_ppo_wd3_step(actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, actor_dims, states, actions, unmasks, logprobs,
                                advantages, reward_sums, H, N, ids, B, ratio_clip, lambda_entropy, inv_batch, objective, flat_grad, stream);
    return ppo_step_impl("erl_mlpn_ppo_step_f32", false, actor_params, critic_params, act_avg, act_std, cri_avg, cri_std, actor_dims,
                         n_dims, states, actions, unmasks, logprobs, advantages, reward_sums, H, N, ids, B, ratio_clip, lambda_entropy,
                         inv_batch, objective, flat_grad, workspace, workspace_bytes, stream);
}

extern "C" int erl_mlpn_ppo_step_discrete_f32(const float *actor_params, const float *critic_params, const float *act_avg,
                                              const float *act_std, const float *cri_avg, const float *cri_std, const int *actor_dims,
                                              int n_dims, const float *states, const int32_t *actions, const uint8_t *unmasks,
                                              const float *logprobs, const float *advantages, const float *reward_sums, int64_t H,
                                              int64_t N, const int64_t *ids, int64_t B, float ratio_clip, float lambda_entropy,
                                              float inv_batch, float *flat_grad, void *workspace, int64_t workspace_bytes, void *stream)
{
    return ppo_step_impl("erl_mlpn_ppo_step_discrete_f32", true, actor_params, critic_params, act_avg, act_std, cri_avg, cri_std,
                         actor_dims, n_dims, states, actions, unmasks, logprobs, advantages, reward_sums, H, N, ids, B, ratio_clip,
                         lambda_entropy, inv_batch, ERL_PPO_OBJ_REFERENCE, flat_grad, workspace, workspace_bytes, stream);
}
