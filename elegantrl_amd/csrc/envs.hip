// GPU-resident vectorised environments used for measurement and demos.
// They implement the env protocol the rollout expects (elegantrl/train/config.py:281-302 as exemplar):
// step(action) -> (state, reward, terminal, truncate), done sub-envs auto-reset and return the
// post-reset state.  One wave per env; the linear maps are staged in LDS.
#include "erl_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, ERL_WAVE));
    return v;
}

// s' = s Ws + a Wa;  reward = -mean(s'^2) - 0.01 mean(a^2);  terminal = max|s'| > 10;
// truncate = (step_count >= max_step) & !terminal;  done rows reset to N(0,1) keyed by (seed, env, episode).
__global__ __launch_bounds__(256) void synenv_step_kernel(float *__restrict__ state, const float *__restrict__ action,
                                                          const float *__restrict__ Ws, const float *__restrict__ Wa,
                                                          int32_t *__restrict__ step_count, int32_t *__restrict__ episode,
                                                          float *__restrict__ reward, uint8_t *__restrict__ terminal,
                                                          uint8_t *__restrict__ truncate, int64_t N, int S, int A, int max_step,
                                                          uint64_t seed)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *lWs = smem, *lWa = lWs + S * S, *rows = lWa + A * S;  // rows: 4 waves x (S + A)
    for (int e = threadIdx.x; e < S * S; e += 256) lWs[e] = Ws[e];
    for (int e = threadIdx.x; e < A * S; e += 256) lWa[e] = Wa[e];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *srow = rows + wave * (S + A), *arow = srow + S;
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        for (int j = lane; j < S; j += 64) srow[j] = state[n * S + j];
        float a2 = 0.f;
        for (int k = lane; k < A; k += 64) {
            const float a = action[n * A + k];
            arow[k] = a;
            a2 += a * a;
        }
        a2 = wave_sum(a2);
        // LDS writes above are read by other lanes of the same wave only
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        float out[2] = {0.f, 0.f};
        float sq = 0.f, mx = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            if (j < S) {
                float acc = 0.f;
                for (int i = 0; i < S; ++i) acc = fmaf(srow[i], lWs[i * S + j], acc);
                for (int k = 0; k < A; ++k) acc = fmaf(arow[k], lWa[k * S + j], acc);
                out[h] = acc;
                sq += acc * acc;
                mx = fmaxf(mx, fabsf(acc));
            }
        }
        sq = wave_sum(sq);
        mx = wave_max(mx);
        const int sc = step_count[n] + 1;
        const bool term = mx > 10.f;
        const bool trunc = (sc >= max_step) && !term;
        const bool done = term || trunc;
        const int ep = episode[n];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            if (j < S) state[n * S + j] = done ? philox_normal(seed, (uint64_t)(ep + 1), (uint32_t)n, (uint32_t)j) : out[h];
        }
        if (lane == 0) {
            reward[n] = -(sq / (float)S) - 0.01f * (a2 / (float)A);
            terminal[n] = term;
            truncate[n] = trunc;
            step_count[n] = done ? 0 : sc;
            if (done) episode[n] = ep + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Pendulum-v1 (g = 10, m = l = 1, dt = 0.05, |u| <= 2, |theta_dot| <= 8) behind the reference's wrapper
// scaling (elegantrl/envs/CustomGymEnv.py:42-44): torque = 2 * action, reward = 0.5 * gym reward.
__global__ __launch_bounds__(256) void pendulum_step_kernel(float *__restrict__ phys, float *__restrict__ obs,
                                                            const float *__restrict__ action, int32_t *__restrict__ step_count,
                                                            int32_t *__restrict__ episode, float *__restrict__ reward,
                                                            uint8_t *__restrict__ terminal, uint8_t *__restrict__ truncate,
                                                            int64_t N, int max_step, uint64_t seed)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float PI = 3.14159265358979323846f;
    float th = phys[2 * n], thdot = phys[2 * n + 1];
    float u = 2.f * action[n];
    u = fminf(fmaxf(u, -2.f), 2.f);
    const float two_pi = 2.f * PI;
    float ang = fmodf(th + PI, two_pi);
    if (ang < 0.f) ang += two_pi;
    ang -= PI;  // angle_normalize
    const float cost = ang * ang + 0.1f * thdot * thdot + 0.001f * u * u;
    float nthdot = thdot + (3.f * 10.f / 2.f * sinf(th) + 3.f * u) * 0.05f;
    nthdot = fminf(fmaxf(nthdot, -8.f), 8.f);
    float nth = th + nthdot * 0.05f;
    const int sc = step_count[n] + 1;
    const bool trunc = sc >= max_step;
    if (trunc) {  // reset: theta ~ U(-pi, pi), theta_dot ~ U(-1, 1)
        const int ep = episode[n] + 1;
        const Philox4 p = philox4x32_10((uint32_t)n, 0u, (uint32_t)ep, 0x50454e44u, (uint32_t)seed, (uint32_t)(seed >> 32));
        nth = ((float)(p.x >> 8) * (1.f / 16777216.f) * 2.f - 1.f) * PI;
        nthdot = (float)(p.y >> 8) * (1.f / 16777216.f) * 2.f - 1.f;
        episode[n] = ep;
    }
    phys[2 * n] = nth;
    phys[2 * n + 1] = nthdot;
    obs[3 * n + 0] = cosf(nth);
    obs[3 * n + 1] = sinf(nth);
    obs[3 * n + 2] = nthdot;
    reward[n] = -0.5f * cost;
    terminal[n] = 0;
    truncate[n] = trunc;
    step_count[n] = trunc ? 0 : sc;
}

}  // namespace

extern "C" int erl_synenv_step_f32(float *state, const float *action, const float *Ws, const float *Wa, int32_t *step_count,
                                   int32_t *episode, float *reward, uint8_t *terminal, uint8_t *truncate, int64_t N, int S, int A,
                                   int max_step, uint64_t seed, void *stream)
{
    ERL_REQUIRE(state && action && Ws && Wa && step_count && episode && reward && terminal && truncate,
                "erl_synenv_step_f32: NULL tensor");
    ERL_REQUIRE(N >= 1 && S >= 1 && S <= ERL_MAX_STATE_DIM && A >= 1 && A <= 64, "erl_synenv_step_f32: bad shape");
    const size_t lds = ((size_t)S * S + (size_t)A * S + 4 * (size_t)(S + A)) * sizeof(float);
    if (lds > 64 * 1024) {
        int rc = erl_hip_status(hipFuncSetAttribute((const void *)synenv_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)lds),
                                "hipFuncSetAttribute(synenv)");
        if (rc) return rc;
    }
    int64_t g = erl_cdiv(N, 4);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(synenv_step_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, state, action, Ws, Wa, step_count,
                       episode, reward, terminal, truncate, N, S, A, max_step, seed);
    ERL_LAUNCH_CHECK("erl_synenv_step_f32");
}

extern "C" int erl_pendulum_step_f32(float *phys, float *obs, const float *action, int32_t *step_count, int32_t *episode,
                                     float *reward, uint8_t *terminal, uint8_t *truncate, int64_t N, int max_step, uint64_t seed,
                                     void *stream)
{
    ERL_REQUIRE(phys && obs && action && step_count && episode && reward && terminal && truncate, "erl_pendulum_step_f32: NULL tensor");
    ERL_REQUIRE(N >= 1 && max_step >= 1, "erl_pendulum_step_f32: bad shape");
    hipLaunchKernelGGL(pendulum_step_kernel, dim3((unsigned)erl_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, phys, obs, action,
                       step_count, episode, reward, terminal, truncate, N, max_step, seed);
    ERL_LAUNCH_CHECK("erl_pendulum_step_f32");
}
