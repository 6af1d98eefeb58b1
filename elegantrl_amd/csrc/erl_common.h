// Shared host/device helpers for liberl_hip.so (gfx950 / CDNA4 only: wave = 64, no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/erl_hip.h"

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void erl_set_error(const char *fmt, ...);
// pinned host-mapped counters that device code bumps for faults an asynchronous launch cannot return; one word per source
// (api.cpp), read through erl_async_fault_count.  NULL when pinned memory is unavailable.
enum ErlFaultSource { ERL_FAULT_GAE_LOOKBACK = 0, ERL_FAULT_P2P_EXCHANGE = 1, ERL_FAULT_ADAM_GRID_WAIT = 2, ERL_FAULT_SAC_Q_EXCHANGE = 3, ERL_FAULT_SOURCES = 4 };
uint32_t *erl_fault_word(int source);

// ---- one-shot peer-to-peer exchange (p2p.hip owns the IPC-mapped stages, grad_tail.hip the kernels) ----------------------
#define ERL_P2P_MAX_WORLD 8
struct ErlExchange {
    char *stage[ERL_P2P_MAX_WORLD];       // rank r's stage: [half 0: world rows][half 1: world rows] (own rank: local pointer)
    uint32_t *flags[ERL_P2P_MAX_WORLD];   // rank r's flag table: [sender rank][workgroup] sequence numbers
    int64_t half_bytes, row_bytes;
    int nblk_max;                         // workgroups (256-element slices) a row is sized for
    uint32_t seq, spin_limit;
    int rank, world;
    uint32_t *fault;
    uint32_t *poison;                     // the communicator's sticky word: set when a wait of this update loop timed out
};
int erl_p2p_create(int rank, int world, int64_t max_count, void **out, uint8_t *out_handle);
int erl_p2p_connect(void *p2p, const uint8_t *handles);
// the exchange descriptor of the NEXT launch on this communicator (advances the sequence number)
int erl_p2p_next(void *p2p, ErlExchange *out);
void erl_p2p_set_spin(void *p2p, uint32_t spins);      // 0 restores the default bound
void erl_p2p_destroy(void *p2p);
uint32_t *erl_p2p_poison_word(void *p2p);                  // device word; non-zero: skip the optimiser steps
void erl_p2p_clear_poison_all();                           // host, synchronous: after the fault has been reported
uint32_t *erl_comm_poison_word(void *comm);                // comm.cpp: nullptr unless a peer-to-peer communicator
// grad_tail.hip
int erl_launch_reduce_exchange_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len,
                                   int n_groups, float grad_scale, bool want_partials, const ErlExchange *ex, hipStream_t stream);
int erl_launch_exchange_f64(double *buf, int64_t count, const ErlExchange *ex, hipStream_t stream);

// optim.hip: clip + Adam with the soft target update folded in; sac_fused.hip: the fused SAC step and its shape class
int erl_clip_adam_soft_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                           const int64_t *group_len, int n_groups, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                           float grad_scale, float *soft, float tau, hipStream_t stream);
// rollout_wide.hip: the one-launch rollout step of net_dims = (256, h2) behind erl_mlpn_rollout_step_f32
int erl_rollout_wide_supported(const int *dims, int n_dims, int64_t N);
int erl_rollout_wide_step(const float *actor_params, const float *state_avg, const float *state_std, const int *dims, int n_dims, const float *state, int64_t N,
                          const float *noise, uint64_t seed, uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                          float *out_action_env, hipStream_t stream);
int erl_value_wide_supported(const int *dims, int n_dims);
int erl_value_wide_forward(const float *params, const float *state_avg, const float *state_std, const int *dims, int n_dims, const float *states,
                           int64_t rows, float *values, hipStream_t stream);
int erl_clip_adam_parts_soft_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t len, const double *parts,
                                 int nparts, int32_t step, float lr, float beta1, float beta2, float eps, float max_norm, float *soft, float tau,
                                 hipStream_t stream);
bool erl_sac_fused_supported(int S, int A, const int *hidden, int n_hidden, int E, int64_t B);
int erl_sac_rollout_fused(const float *actor_params, int S, int A, int h0, int h1, const int64_t *aoff, float *env_state, const float *Ws,
                          const float *Wa, float *phys, int32_t *step_count, int32_t *episode, int max_step, uint64_t env_seed, int64_t N, int64_t H,
                          const float *noise, uint64_t seed, uint64_t counter0, float reward_scale, float *out_states, float *out_actions,
                          float *out_rewards, uint8_t *out_undones, uint8_t *out_unmasks, float *out_last_state, hipStream_t stream);
int64_t erl_sac_fused_ws_floats(int S, int A, int h0, int h1, int E, int64_t B, int64_t Pa, int64_t Pc);
int erl_sac_explore_fused(const float *actor_params, int S, int A, int h0, int h1, const int64_t *aoff, const float *state, int64_t N,
                          const float *noise, uint64_t seed, uint64_t counter, float *action_out, float *state_out, float *lp_scratch,
                          hipStream_t sa);
int erl_sac_update_fused(float *actor_params, float *critic_params, float *target_params, float *alpha_log, float *actor_m, float *actor_v,
                         float *critic_m, float *critic_v, float *alpha_m, float *alpha_v, int S, int A, int h0, int h1, int E,
                         const int64_t *aoff, const int64_t *coff, int64_t Pa, int64_t Pc, const float *state, const float *action,
                         const float *reward, const float *undone, const float *unmask, const float *next_state, const float *is_weight,
                         float *td_error_out, int64_t B, const float *eps_next, const float *eps_cur, uint64_t seed, uint64_t counter,
                         float gamma, float target_entropy, float tau, float lr, float beta1, float beta2, float eps_adam, float max_norm,
                         int32_t step, float *objs_out, float *workspace, const ErlRingSample *ring, hipStream_t s);

#define ERL_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            erl_set_error(__VA_ARGS__);        \
            return ERL_EINVAL;                 \
        }                                      \
    } while (0)

static inline int erl_hip_status(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ERL_OK;
    erl_set_error("%s: %s", what, hipGetErrorString(e));
    return -(1000 + (int)e);
}

#define ERL_LAUNCH_CHECK(what) return erl_hip_status(hipGetLastError(), what)

// ---------------------------------------------------------------------------------------------
// measurement hook (api.cpp; include/erl_hip.h erl_kernel_span_*): a kernel's own duration on the device's constant-rate clock, first
// workgroup in to last workgroup out -- what rocprofv3's kernel duration measures, available to bench.py in the loop without a
// profiler and without an event bracket (a bracket perturbs the kernel inside it and adds its own dispatch / completion time).
// Host: `unsigned long long *sp = erl_span_slot(ERL_SPAN_x, workgroups of the launch)` right before the launch (nullptr while the hook is
// off or this launch is not sampled), passed to the
// kernel; device: `const auto t0 = erl_span_in(sp); ... erl_span_out(sp, t0);` (thread 0 of every workgroup; no early return between).
// ---------------------------------------------------------------------------------------------
unsigned long long *erl_span_slot(int tag, int64_t n_workgroups);
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long erl_span_in(const unsigned long long *span) { return span ? wall_clock64() : 0ull; }
// one {entry, exit} record per workgroup, plain stores (no atomics: 800 same-address atomics cost the slab reduction 12 us)
__device__ __forceinline__ void erl_span_out(unsigned long long *span, unsigned long long t0)
{
    if (span && threadIdx.x == 0 && threadIdx.y == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left
        unsigned long long *rec = span + 2 * ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z));
        rec[0] = t0;
        rec[1] = (unsigned long long)wall_clock64();
    }
}
#endif

static inline int64_t erl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// device: wave / block reductions (wave64)
// ---------------------------------------------------------------------------------------------
#define ERL_WAVE 64

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, ERL_WAVE);
    return v;
}

// sum over the whole block; result valid in every thread. `scratch` holds >= blockDim.x/64 T's.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect scratch reuse
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T t = 0;
    for (int w = 0; w < nw; ++w) t += scratch[w];
    return t;
}

// the five raw sums of the advantage normalisation (elegantrl/agents/AgentPPO.py:149) from per-workgroup fp64 partials
// [n][3] = (sum adv, sum over the [::4, ::4] subsample, sum of squares over it), folded in a fixed order by one 256-thread block:
// adv_stats_fold_kernel (gae.hip) and the extra block of the update loop's weight-image kernel (grad_tail.hip)
__device__ __forceinline__ void erl_adv_stats_fold_block(const double *__restrict__ partials, int nparts, int H, int N, double *__restrict__ stats,
                                                         double *scratch)
{
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        s0 += partials[(size_t)i * 3 + 0];
        s1 += partials[(size_t)i * 3 + 1];
        s2 += partials[(size_t)i * 3 + 2];
    }
    s0 = block_sum(s0, scratch);
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        stats[0] = s0;
        stats[1] = (double)H * (double)N;
        stats[2] = s1;
        stats[3] = s2;
        stats[4] = (double)((H + 3) / 4) * (double)((N + 3) / 4);
        stats[5] = stats[6] = stats[7] = 0.0;      // (the 8-double block is all-reduced whole under data parallelism)
    }
}

// ---------------------------------------------------------------------------------------------
// device: ONE Adam update for every optimiser kernel of the library (clip_adam_kernel, the fused tails, the long-group and the
// partial-norm kernels): every product and sum rounded separately and spelled out, so that no kernel's code generation (fma
// contraction differs with the surrounding code) can make two routes through the library disagree -- m1 * beta1 and
// (1 - beta1) * g cancel almost completely when the gradient changes sign and scale, which turns a contraction difference
// into a visible one.  torch.optim.Adam defaults: exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2);
// denom = sqrt(v) / sqrt(bc2) + eps; p -= (lr / bc1) * m / denom.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void erl_adam_update(float gx, float &m1, float &m2, float &p, float beta1, float beta2, float eps, float step_size,
                                                float bc2_sqrt)
{
#pragma clang fp contract(off)      // (HIP's __fmul_rn / __fadd_rn are plain operators: they do NOT stop the contraction)
    const float a = m1 * beta1 + (1.f - beta1) * gx;
    const float b = m2 * beta2 + (1.f - beta2) * (gx * gx);
    const float denom = sqrtf(b) / bc2_sqrt + eps;
    m1 = a;
    m2 = b;
    p = p - step_size * (a / denom);
}

// AgentBase.soft_update (elegantrl/agents/AgentBase.py:270-278): tar = cur * tau + tar * (1 - tau), products rounded separately
__device__ __forceinline__ float erl_soft_update(float cur, float tar, float tau)
{
#pragma clang fp contract(off)
    return cur * tau + tar * (1.0f - tau);
}

// x + y as one rounded add whatever surrounds it
__device__ __forceinline__ float erl_add_rn(float x, float y)
{
#pragma clang fp contract(off)
    return x + y;
}

// x * y rounded on its own (never folded into a following add)
__device__ __forceinline__ float erl_mul_rn(float x, float y)
{
#pragma clang fp contract(off)
    return x * y;
}

// ---------------------------------------------------------------------------------------------
// device: Philox4x32-10 counter RNG + Box-Muller (production noise for K1 and env resets)
// ---------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// two uniforms in (0,1] -> two independent N(0,1).  Hardware transcendentals: v_log_f32 (log2), v_sqrt_f32 and
// v_sin_f32 / v_cos_f32, which take their argument in REVOLUTIONS -- u2 in [0, 1) is exactly one turn, so there is no
// 2 pi multiply and no range reduction (library sincosf / logf cost ~10x the instructions; a tile whose env resets pays
// 64 of these draws per env inside the rollout's dependent chain).  Absolute error of a draw ~1e-6: production noise only
// has to be N(0,1); every parity test injects its noise.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &n0, float &n1)
{
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));   // -2 ln(u1) = -2 ln2 log2(u1)
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}

// one U[0,1) of the stream (seed, counter, env) -- the categorical draw of the discrete policy (24-bit mantissa)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t counter, uint32_t env)
{
    const Philox4 p = philox4x32_10(env, 0x5eedu, (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    return (float)(p.x >> 8) * (1.0f / 16777216.0f);
}

// i-th N(0,1) of the stream (seed, counter, env): dims are consumed 4 per Philox call.
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t counter, uint32_t env, uint32_t dim)
{
    const Philox4 p = philox4x32_10(env, dim >> 2, (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    float n0, n1;
    if ((dim & 2) == 0) box_muller(p.x, p.y, n0, n1);
    else box_muller(p.z, p.w, n0, n1);
    return (dim & 1) ? n1 : n0;
}
