// GPU-resident vectorised environments used for measurement and demos.
// They implement the env protocol the rollout expects (elegantrl/train/config.py:281-302 as exemplar):
// step(action) -> (state, reward, terminal, truncate), done sub-envs auto-reset and return the
// post-reset state.  SynVecEnv: 16 envs per workgroup on fp32 MFMA (synenv_tile_kernel), one wave per env with the linear
// maps staged in LDS for action_dim > 16 (synenv_step_kernel).
#include "erl_common.h"
#include "mlp_chain.h"

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

// The same step on the matrix cores, shaped like the latency form of the rollout kernel it alternates with: a workgroup
// owns 16 envs, wave w owns state features 16 w .. 16 w + 15 of s'^T = Ws^T s^T + Wa^T a^T (v_mfma_f32_16x16x4_f32:
// A operand = the wave's 16 columns of Ws / Wa read straight from L2, B operand = the env rows as 16-byte loads).  Every
// global load of the step is issued up front (one round trip); the per-env sum of squares and max meet in LDS (one
// barrier, fixed order); lane (m, q) then owns s'[m][16 w + 4 q .. +3] and stores it -- or the reset draw -- as 16 bytes.
template <bool VS, bool VA>
__global__ __launch_bounds__(512) void synenv_tile_kernel(float *__restrict__ state, const float *__restrict__ action,
                                                          const float *__restrict__ Ws, const float *__restrict__ Wa,
                                                          int32_t *__restrict__ step_count, int32_t *__restrict__ episode,
                                                          float *__restrict__ reward, uint8_t *__restrict__ terminal,
                                                          uint8_t *__restrict__ truncate, int64_t N, int S, int A, int max_step,
                                                          uint64_t seed)
{
    __shared__ float red[8][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, q = lane >> 4;
    const int nt = (S + 15) >> 4;
    const int64_t n = (int64_t)blockIdx.x * 16 + l15;
    const bool valid = n < N;
    const int64_t row = valid ? n : N - 1;
    const int j = 16 * wave + l15, jc = min(j, S - 1);
    const bool jon = j < S;

    float4 xs[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < nt) xs[t] = load4<VS>(state + row * S, 16 * t + 4 * q, S);
    const float4 av = load4<VA>(action + row * A, 4 * q, A);
    float wa[8][4], wb[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * t + 4 * q + r;
                const float w = Ws[(size_t)min(k, S - 1) * S + jc];
                wa[t][r] = (k < S && jon) ? w : 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float w = Wa[(size_t)min(k, A - 1) * S + jc];
        wb[r] = (k < A && jon) ? w : 0.f;
    }
    const int sc = step_count[row] + 1, ep = episode[row];

    float a2 = (av.x * av.x + av.y * av.y) + (av.z * av.z + av.w * av.w);     // load4 zero-fills beyond A
    a2 += __shfl_xor(a2, 16, 64);
    a2 += __shfl_xor(a2, 32, 64);

    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    c0 = mfma16(wb[0], av.x, c0);
    c1 = mfma16(wb[1], av.y, c1);
    c0 = mfma16(wb[2], av.z, c0);
    c1 = mfma16(wb[3], av.w, c1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t < nt) {
            c0 = mfma16(wa[t][0], xs[t].x, c0);
            c1 = mfma16(wa[t][1], xs[t].y, c1);
            c0 = mfma16(wa[t][2], xs[t].z, c0);
            c1 = mfma16(wa[t][3], xs[t].w, c1);
        }
    }
    // lane (m = l15, q): s'[m][16 wave + 4 q + r]
    const int j0 = 16 * wave + 4 * q;
    float out[4], sq = 0.f, mx = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        out[r] = c0[r] + c1[r];
        if (j0 + r < S) {
            sq += out[r] * out[r];
            mx = fmaxf(mx, fabsf(out[r]));
        }
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (q == 0) { red[wave][l15][0] = sq; red[wave][l15][1] = mx; }
    lds_barrier();
    sq = 0.f; mx = 0.f;
    for (int w = 0; w < nt; ++w) { sq += red[w][l15][0]; mx = fmaxf(mx, red[w][l15][1]); }

    const bool term = mx > 10.f;
    const bool trunc = (sc >= max_step) && !term;
    const bool done = term || trunc;
    if (valid && j0 < S) {
        if (done) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = philox_normal(seed, (uint64_t)(ep + 1), (uint32_t)n, (uint32_t)(j0 + r));
        }
        float *dst = state + n * S + j0;
        if (VS) *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (j0 + r < S) dst[r] = out[r];
        }
    }
    if (valid && wave == 0 && q == 0) {
        reward[n] = -(sq / (float)S) - 0.01f * (a2 / (float)A);
        terminal[n] = term;
        truncate[n] = trunc;
        step_count[n] = done ? 0 : sc;
        if (done) episode[n] = ep + 1;
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
    static const bool tile_ok = [] { const char *e = getenv("ERL_SYNENV_TILE"); return !e || atoi(e) != 0; }();
    if (A <= 16 && tile_ok) {   // 16 envs per workgroup on the matrix cores
        auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool vs = (S % 4 == 0) && al(state), va = (A % 4 == 0) && al(action);
        const dim3 grid((unsigned)erl_cdiv(N, 16)), block(64 * (unsigned)((S + 15) / 16));
#define SYN_LAUNCH(VS_, VA_)                                                                                                   \
    hipLaunchKernelGGL((synenv_tile_kernel<VS_, VA_>), grid, block, 0, (hipStream_t)stream, state, action, Ws, Wa, step_count, \
                       episode, reward, terminal, truncate, N, S, A, max_step, seed)
        if (vs && va) SYN_LAUNCH(true, true);
        else if (vs) SYN_LAUNCH(true, false);
        else if (va) SYN_LAUNCH(false, true);
        else SYN_LAUNCH(false, false);
#undef SYN_LAUNCH
        ERL_LAUNCH_CHECK("erl_synenv_step_f32");
    }
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
