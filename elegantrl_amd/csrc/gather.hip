// K5: PPO minibatch index decomposition + gather; K8/K9: replay ring write / sample.  gfx950.
// All HBM-bound byte/index work: flat element mapping so that reads inside a gathered row and all
// writes are coalesced; indices are int64 and bit-exact (ids % L, ids // L; the reference's th.fmod /
// th.div(rounding_mode='floor') on non-negative ids).
#include "erl_common.h"

namespace {

__global__ __launch_bounds__(256) void split_ids_kernel(const int64_t *__restrict__ ids, int64_t B, int64_t L,
                                                        int64_t *__restrict__ ids0, int64_t *__restrict__ ids1)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int64_t id = ids[i];
    const int64_t q = id / L;
    if (ids0) ids0[i] = id - q * L;
    if (ids1) ids1[i] = q;
}

// One "virtual row" per sample of width W = S + A + 4 columns:
//   [0,S) state | [S,S+A) action | S+A: unmask | +1: logprob | +2: advantage | +3: reward_sum
// source element = x[(t*N + n) * dim + c] with t = id % H, n = id // H   (AgentPPO.py:179-187)
__global__ __launch_bounds__(256) void ppo_gather_kernel(const float *__restrict__ states, const float *__restrict__ actions,
                                                         const uint8_t *__restrict__ unmasks,
                                                         const float *__restrict__ logprobs,
                                                         const float *__restrict__ advantages,
                                                         const float *__restrict__ reward_sums, int64_t H, int64_t N,
                                                         int S, int A, const int64_t *__restrict__ ids, int64_t B,
                                                         float *__restrict__ o_state, float *__restrict__ o_action,
                                                         uint8_t *__restrict__ o_unmask, float *__restrict__ o_logprob,
                                                         float *__restrict__ o_adv, float *__restrict__ o_ret,
                                                         int64_t *__restrict__ o_ids0, int64_t *__restrict__ o_ids1)
{
    const int W = S + A + 4;
    const int64_t total = B * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / W;
        const int c = (int)(e - b * W);
        const int64_t id = ids[b];
        const int64_t n = id / H, t = id - n * H;
        const int64_t row = t * N + n;
        if (c < S) {
            if (o_state) o_state[b * S + c] = states[row * S + c];
        } else if (c < S + A) {
            if (o_action) o_action[b * A + (c - S)] = actions[row * A + (c - S)];
        } else {
            switch (c - S - A) {
            case 0:
                if (o_unmask) o_unmask[b] = unmasks[row];
                if (o_ids0) o_ids0[b] = t;
                if (o_ids1) o_ids1[b] = n;
                break;
            case 1: if (o_logprob) o_logprob[b] = logprobs[row]; break;
            case 2: if (o_adv) o_adv[b] = advantages[row]; break;
            default: if (o_ret) o_ret[b] = reward_sums[row]; break;
            }
        }
    }
}

// Ring write: virtual row of width W = S + A + 3 per (time row i, sequence q);
// destination time row = (p + i) mod max_size  (replay_buffer.py:86-105).
// ACT_U8: discrete actions -- (add, num_seqs) int32 in, (max_size, num_seqs) uint8 stored (replay_buffer.py:53-54; the
// assignment of an int32 row into the uint8 buffer keeps the low byte), one column instead of A.
template <bool FLAG_F32, bool ACT_U8>
__global__ __launch_bounds__(256) void replay_write_kernel(float *__restrict__ b_states, void *__restrict__ b_actions,
                                                           float *__restrict__ b_rewards, float *__restrict__ b_undones,
                                                           float *__restrict__ b_unmasks, const float *__restrict__ states,
                                                           const void *__restrict__ actions,
                                                           const float *__restrict__ rewards, const void *__restrict__ undones,
                                                           const void *__restrict__ unmasks, int64_t max_size,
                                                           int64_t num_seqs, int S, int A, int64_t p, int64_t add)
{
    const int W = S + A + 3;
    const int64_t rows = add * num_seqs, total = rows * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / W;
        const int c = (int)(e - r * W);
        const int64_t i = r / num_seqs, q = r - i * num_seqs;
        int64_t ti = p + i;
        if (ti >= max_size) ti -= max_size;
        const int64_t d = ti * num_seqs + q;
        if (c < S) b_states[d * S + c] = states[r * S + c];
        else if (c < S + A) {
            if (ACT_U8) ((uint8_t *)b_actions)[d] = (uint8_t)((const int32_t *)actions)[r];
            else ((float *)b_actions)[d * A + (c - S)] = ((const float *)actions)[r * A + (c - S)];
        }
        else if (c == S + A) b_rewards[d] = rewards[r];
        else if (c == S + A + 1)
            b_undones[d] = FLAG_F32 ? ((const float *)undones)[r] : (((const uint8_t *)undones)[r] ? 1.f : 0.f);
        else
            b_unmasks[d] = FLAG_F32 ? ((const float *)unmasks)[r] : (((const uint8_t *)unmasks)[r] ? 1.f : 0.f);
    }
}

// Sample: virtual row of width W = 2S + A + 3:
//   [0,S) state | [S,S+A) action | reward | undone | unmask | [S+A+3, 2S+A+3) next_state = states[t+1, n]
// A workgroup owns RS_SAMPLES samples: their ring rows are resolved once (one 64-bit division per sample, results in
// LDS and in ids0/ids1), then all 256 threads stream the RS_SAMPLES x W output elements with 32-bit index math.
constexpr int RS_SAMPLES = 128;

template <bool ACT_U8>
__global__ __launch_bounds__(256) void replay_sample_kernel(const float *__restrict__ b_states,
                                                            const void *__restrict__ b_actions,
                                                            const float *__restrict__ b_rewards,
                                                            const float *__restrict__ b_undones,
                                                            const float *__restrict__ b_unmasks, int64_t num_seqs, int S,
                                                            int A, const int64_t *__restrict__ ids, int64_t B,
                                                            int64_t sample_len, float *__restrict__ o_state,
                                                            void *__restrict__ o_action, float *__restrict__ o_reward,
                                                            float *__restrict__ o_undone, float *__restrict__ o_unmask,
                                                            float *__restrict__ o_next, int64_t *__restrict__ o_ids0,
                                                            int64_t *__restrict__ o_ids1, int spw, unsigned long long *span)
{
    __shared__ int64_t s_row[RS_SAMPLES];
    const unsigned long long t_span = erl_span_in(span);
    const int W = 2 * S + A + 3;
    for (int64_t b0 = (int64_t)blockIdx.x * spw; b0 < B; b0 += (int64_t)gridDim.x * spw) {
        const int nb = (int)min((int64_t)spw, B - b0);
        __syncthreads();   // s_row reuse
        if ((int)threadIdx.x < nb) {
            const int64_t id = ids[b0 + threadIdx.x];
            const int64_t n = id / sample_len, t = id - n * sample_len;   // ids0 = ids % L, ids1 = ids // L  (:124-125)
            s_row[threadIdx.x] = t * num_seqs + n;
            if (o_ids0) o_ids0[b0 + threadIdx.x] = t;
            if (o_ids1) o_ids1[b0 + threadIdx.x] = n;
        }
        __syncthreads();
        const int total = nb * W;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int bl = e / W, c = e - bl * W;
            const int64_t row = s_row[bl], b = b0 + bl;
            if (c < S) {
                o_state[b * S + c] = b_states[row * S + c];
            } else if (c < S + A) {
                if (ACT_U8) ((uint8_t *)o_action)[b] = ((const uint8_t *)b_actions)[row];
                else ((float *)o_action)[b * A + (c - S)] = ((const float *)b_actions)[row * A + (c - S)];
            } else if (c < S + A + 3) {
                const int k = c - S - A;
                if (k == 0) o_reward[b] = b_rewards[row];
                else if (k == 1) o_undone[b] = b_undones[row];
                else o_unmask[b] = b_unmasks[row];
            } else {
                const int cs = c - S - A - 3;
                o_next[b * S + cs] = b_states[(row + num_seqs) * S + cs];
            }
        }
    }
    erl_span_out(span, t_span);
}

// ---- the interleaved ring (round 6): ONE block ring[num_seqs][max_size][RW] fp32, RW = S + A + 3 rounded up to 4 floats, a row =
// [state (S) | action (A) | reward | undone | unmask | pad].  A transition is one 16-byte-aligned row, and its next state
// (states[ids0 + 1, ids1], replay_buffer.py:133) is the head of the NEXT row of the same sequence: a sample reads RW + S consecutive
// floats -- one or two 128-byte lines instead of the planar layout's six to seven (44-byte state row, 44-byte next-state row num_seqs
// rows further on, 12-byte action row, three 4-byte scalars: 5.8x the algorithmic bytes fetched at B = 2^20, profiles/r05_k9_pmc_by_size.json).
// The class keeps the reference's attributes as strided VIEWS of the block (train/replay_buffer.py).
template <bool FLAG_F32>
__global__ __launch_bounds__(256) void replay_write_rows_kernel(float *__restrict__ ring, const float *__restrict__ states,
                                                                const float *__restrict__ actions, const float *__restrict__ rewards,
                                                                const void *__restrict__ undones, const void *__restrict__ unmasks,
                                                                int64_t max_size, int64_t num_seqs, int S, int A, int RW, int64_t p, int64_t add)
{
    const int W = S + A + 3;
    const int64_t rows = add * num_seqs, total = rows * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / W;
        const int c = (int)(e - r * W);
        const int64_t i = r / num_seqs, q = r - i * num_seqs;
        int64_t ti = p + i;
        if (ti >= max_size) ti -= max_size;
        float v;
        if (c < S) v = states[r * S + c];
        else if (c < S + A) v = actions[r * A + (c - S)];
        else if (c == S + A) v = rewards[r];
        else if (c == S + A + 1) v = FLAG_F32 ? ((const float *)undones)[r] : (((const uint8_t *)undones)[r] ? 1.f : 0.f);
        else v = FLAG_F32 ? ((const float *)unmasks)[r] : (((const uint8_t *)unmasks)[r] ? 1.f : 0.f);
        ring[(q * max_size + ti) * RW + c] = v;
    }
}

// a thread moves one 16-byte chunk: chunks [0, RW / 4) of the sample's own row, then ceil(S / 4) chunks of the next row's state
__global__ __launch_bounds__(256) void replay_sample_rows_kernel(const float *__restrict__ ring, int64_t max_size, int S, int A, int RW,
                                                                 const int64_t *__restrict__ ids, int64_t B, int64_t sample_len,
                                                                 float *__restrict__ o_state, float *__restrict__ o_action,
                                                                 float *__restrict__ o_reward, float *__restrict__ o_undone,
                                                                 float *__restrict__ o_unmask, float *__restrict__ o_next,
                                                                 int64_t *__restrict__ o_ids0, int64_t *__restrict__ o_ids1, int spw,
                                                                 unsigned long long *span)
{
    __shared__ int64_t s_row[RS_SAMPLES], s_nxt[RS_SAMPLES];
    const unsigned long long t_span = erl_span_in(span);
    const int C0 = RW >> 2, C1 = (S + 3) >> 2, CW = C0 + C1;
    for (int64_t b0 = (int64_t)blockIdx.x * spw; b0 < B; b0 += (int64_t)gridDim.x * spw) {
        const int nb = (int)min((int64_t)spw, B - b0);
        __syncthreads();   // s_row reuse
        if ((int)threadIdx.x < nb) {
            const int64_t id = ids[b0 + threadIdx.x];
            const int64_t n = id / sample_len, t = id - n * sample_len;   // ids0 = ids % L, ids1 = ids // L  (:124-125)
            s_row[threadIdx.x] = n * max_size + t;
            // the next row of the same sequence; ids0 = max_size - 1 has none (the reference would raise an IndexError; the uniform
            // sampler's sample_len = cur_size - 1 and the prioritised sampler's moves never produce it): clamped, never out of the block
            s_nxt[threadIdx.x] = n * max_size + (t + 1 < max_size ? t + 1 : t);
            if (o_ids0) o_ids0[b0 + threadIdx.x] = t;
            if (o_ids1) o_ids1[b0 + threadIdx.x] = n;
        }
        __syncthreads();
        const int total = nb * CW;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int bl = e / CW, c = e - bl * CW;
            const int64_t row = s_row[bl], b = b0 + bl;
            if (c < C0) {
                const float4 v4 = *reinterpret_cast<const float4 *>(ring + row * RW + 4 * c);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = 4 * c + k;
                    if (col < S) o_state[b * S + col] = v[k];
                    else if (col < S + A) o_action[b * A + (col - S)] = v[k];
                    else if (col == S + A) o_reward[b] = v[k];
                    else if (col == S + A + 1) o_undone[b] = v[k];
                    else if (col == S + A + 2) o_unmask[b] = v[k];
                }
            } else {
                const int cc = c - C0;
                const float4 v4 = *reinterpret_cast<const float4 *>(ring + s_nxt[bl] * RW + 4 * cc);   // states[ids0 + 1, ids1]  (:133)
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * cc + k < S) o_next[b * S + 4 * cc + k] = v[k];
            }
        }
    }
    erl_span_out(span, t_span);
}

inline int grid_for(int64_t total)
{
    int64_t g = erl_cdiv(total, 256);
    if (g > 256 * 8) g = 256 * 8;  // grid-stride beyond 8 blocks per CU
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int erl_split_ids_i64(const int64_t *ids, int64_t B, int64_t sample_len, int64_t *ids0, int64_t *ids1, void *stream)
{
    ERL_REQUIRE(ids && B >= 0 && sample_len >= 1, "erl_split_ids_i64: bad argument");
    if (B == 0) return ERL_OK;
    hipLaunchKernelGGL(split_ids_kernel, dim3((unsigned)erl_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, ids, B,
                       sample_len, ids0, ids1);
    ERL_LAUNCH_CHECK("erl_split_ids_i64");
}

extern "C" int erl_ppo_gather_f32(const float *states, const float *actions, const uint8_t *unmasks, const float *logprobs,
                                  const float *advantages, const float *reward_sums, int64_t H, int64_t N, int S, int A,
                                  const int64_t *ids, int64_t B, float *out_state, float *out_action, uint8_t *out_unmask,
                                  float *out_logprob, float *out_advantage, float *out_reward_sum, int64_t *out_ids0,
                                  int64_t *out_ids1, void *stream)
{
    ERL_REQUIRE(ids && H >= 1 && N >= 1 && S >= 1 && A >= 1 && B >= 0, "erl_ppo_gather_f32: bad shape");
    ERL_REQUIRE((!out_state || states) && (!out_action || actions) && (!out_unmask || unmasks) &&
                    (!out_logprob || logprobs) && (!out_advantage || advantages) && (!out_reward_sum || reward_sums),
                "erl_ppo_gather_f32: output requested without its source tensor");
    if (B == 0) return ERL_OK;
    hipLaunchKernelGGL(ppo_gather_kernel, dim3(grid_for(B * (S + A + 4))), dim3(256), 0, (hipStream_t)stream, states, actions,
                       unmasks, logprobs, advantages, reward_sums, H, N, S, A, ids, B, out_state, out_action, out_unmask,
                       out_logprob, out_advantage, out_reward_sum, out_ids0, out_ids1);
    ERL_LAUNCH_CHECK("erl_ppo_gather_f32");
}

namespace {

int replay_write_impl(const char *what, bool act_u8, float *buf_states, void *buf_actions, float *buf_rewards, float *buf_undones,
                      float *buf_unmasks, const float *states, const void *actions, const float *rewards, const void *undones,
                      const void *unmasks, int flag_is_f32, int64_t max_size, int64_t num_seqs, int S, int A, int64_t p, int64_t add,
                      void *stream)
{
    ERL_REQUIRE(buf_states && buf_actions && buf_rewards && buf_undones && buf_unmasks && states && actions && rewards && undones &&
                    unmasks,
                "%s: NULL tensor", what);
    ERL_REQUIRE(max_size >= 1 && num_seqs >= 1 && S >= 1 && A >= 1, "%s: bad shape", what);
    ERL_REQUIRE(add >= 0 && add <= max_size && p >= 0 && p <= max_size, "%s: add=%lld p=%lld max_size=%lld", what, (long long)add,
                (long long)p, (long long)max_size);
    if (add == 0) return ERL_OK;
    const int g = grid_for(add * num_seqs * (S + A + 3));
    hipStream_t st = (hipStream_t)stream;
#define ERL_RW(F, U)                                                                                                            \
    hipLaunchKernelGGL((replay_write_kernel<F, U>), dim3(g), dim3(256), 0, st, buf_states, buf_actions, buf_rewards, buf_undones, \
                       buf_unmasks, states, actions, rewards, undones, unmasks, max_size, num_seqs, S, A, p, add)
    if (flag_is_f32 && act_u8) ERL_RW(true, true);
    else if (flag_is_f32) ERL_RW(true, false);
    else if (act_u8) ERL_RW(false, true);
    else ERL_RW(false, false);
#undef ERL_RW
    return erl_hip_status(hipGetLastError(), what);
}

int replay_sample_impl(const char *what, bool act_u8, const float *buf_states, const void *buf_actions, const float *buf_rewards,
                       const float *buf_undones, const float *buf_unmasks, int64_t max_size, int64_t num_seqs, int S, int A,
                       const int64_t *ids, int64_t B, int64_t sample_len, float *out_state, void *out_action, float *out_reward,
                       float *out_undone, float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1,
                       void *stream)
{
    ERL_REQUIRE(buf_states && buf_actions && buf_rewards && buf_undones && buf_unmasks && ids, "%s: NULL tensor", what);
    ERL_REQUIRE(out_state && out_action && out_reward && out_undone && out_unmask && out_next_state, "%s: NULL output", what);
    // sample_len = cur_size - 1 for the uniform sampler; the prioritised sampler passes cur_size (<= max_size) and guarantees
    // ids0 <= cur_size - 2 itself, so that row ids0 + 1 exists either way
    ERL_REQUIRE(num_seqs >= 1 && S >= 1 && A >= 1 && B >= 0 && sample_len >= 1 && sample_len <= max_size,
                "%s: bad shape (sample_len=%lld max_size=%lld)", what, (long long)sample_len, (long long)max_size);
    if (B == 0) return ERL_OK;
    // samples per workgroup: 128 for large batches; a small batch (the SAC step's 256 rows) is spread so that every thread
    // moves about one element in ONE round trip -- two workgroups looping 14 dependent trips took 28-35 us for 256 rows
    const int W = 2 * S + A + 3;
    int64_t spw = B / 1024;
    if (spw < 256 / W) spw = 256 / W;
    if (spw < 1) spw = 1;
    if (spw > RS_SAMPLES) spw = RS_SAMPLES;
    const int g = grid_for(erl_cdiv(B, spw) * 256);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *sp = erl_span_slot(ERL_SPAN_REPLAY_SAMPLE, g);
    if (act_u8)
        hipLaunchKernelGGL((replay_sample_kernel<true>), dim3(g), dim3(256), 0, st, buf_states, buf_actions, buf_rewards, buf_undones,
                           buf_unmasks, num_seqs, S, A, ids, B, sample_len, out_state, out_action, out_reward, out_undone, out_unmask,
                           out_next_state, out_ids0, out_ids1, (int)spw, sp);
    else
        hipLaunchKernelGGL((replay_sample_kernel<false>), dim3(g), dim3(256), 0, st, buf_states, buf_actions, buf_rewards, buf_undones,
                           buf_unmasks, num_seqs, S, A, ids, B, sample_len, out_state, out_action, out_reward, out_undone, out_unmask,
                           out_next_state, out_ids0, out_ids1, (int)spw, sp);
    return erl_hip_status(hipGetLastError(), what);
}

}  // namespace

extern "C" int erl_replay_write_f32(float *buf_states, float *buf_actions, float *buf_rewards, float *buf_undones,
                                    float *buf_unmasks, const float *states, const float *actions, const float *rewards,
                                    const void *undones, const void *unmasks, int flag_is_f32, int64_t max_size,
                                    int64_t num_seqs, int S, int A, int64_t p, int64_t add, void *stream)
{
    return replay_write_impl("erl_replay_write_f32", false, buf_states, buf_actions, buf_rewards, buf_undones, buf_unmasks, states,
                             actions, rewards, undones, unmasks, flag_is_f32, max_size, num_seqs, S, A, p, add, stream);
}

extern "C" int erl_replay_write_discrete_f32(float *buf_states, uint8_t *buf_actions, float *buf_rewards, float *buf_undones,
                                             float *buf_unmasks, const float *states, const int32_t *actions, const float *rewards,
                                             const void *undones, const void *unmasks, int flag_is_f32, int64_t max_size,
                                             int64_t num_seqs, int S, int64_t p, int64_t add, void *stream)
{
    return replay_write_impl("erl_replay_write_discrete_f32", true, buf_states, buf_actions, buf_rewards, buf_undones, buf_unmasks,
                             states, actions, rewards, undones, unmasks, flag_is_f32, max_size, num_seqs, S, 1, p, add, stream);
}

extern "C" int erl_replay_sample_f32(const float *buf_states, const float *buf_actions, const float *buf_rewards,
                                     const float *buf_undones, const float *buf_unmasks, int64_t max_size, int64_t num_seqs,
                                     int S, int A, const int64_t *ids, int64_t B, int64_t sample_len, float *out_state,
                                     float *out_action, float *out_reward, float *out_undone, float *out_unmask,
                                     float *out_next_state, int64_t *out_ids0, int64_t *out_ids1, void *stream)
{
    return replay_sample_impl("erl_replay_sample_f32", false, buf_states, buf_actions, buf_rewards, buf_undones, buf_unmasks,
                              max_size, num_seqs, S, A, ids, B, sample_len, out_state, out_action, out_reward, out_undone,
                              out_unmask, out_next_state, out_ids0, out_ids1, stream);
}

extern "C" int erl_replay_sample_discrete_f32(const float *buf_states, const uint8_t *buf_actions, const float *buf_rewards,
                                              const float *buf_undones, const float *buf_unmasks, int64_t max_size,
                                              int64_t num_seqs, int S, const int64_t *ids, int64_t B, int64_t sample_len,
                                              float *out_state, uint8_t *out_action, float *out_reward, float *out_undone,
                                              float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1,
                                              void *stream)
{
    return replay_sample_impl("erl_replay_sample_discrete_f32", true, buf_states, buf_actions, buf_rewards, buf_undones,
                              buf_unmasks, max_size, num_seqs, S, 1, ids, B, sample_len, out_state, out_action, out_reward,
                              out_undone, out_unmask, out_next_state, out_ids0, out_ids1, stream);
}

extern "C" int64_t erl_replay_row_floats(int S, int A) { return S >= 1 && A >= 1 ? ((int64_t)S + A + 3 + 3) / 4 * 4 : -1; }

extern "C" int erl_replay_write_rows_f32(float *ring, int64_t max_size, int64_t num_seqs, int S, int A, const float *states,
                                         const float *actions, const float *rewards, const void *undones, const void *unmasks,
                                         int flag_is_f32, int64_t p, int64_t add, void *stream)
{
    ERL_REQUIRE(ring && states && actions && rewards && undones && unmasks, "erl_replay_write_rows_f32: NULL tensor");
    ERL_REQUIRE(max_size >= 1 && num_seqs >= 1 && S >= 1 && A >= 1, "erl_replay_write_rows_f32: bad shape");
    ERL_REQUIRE((reinterpret_cast<uintptr_t>(ring) & 15) == 0, "erl_replay_write_rows_f32: the ring must be 16-byte aligned");
    ERL_REQUIRE(add >= 0 && add <= max_size && p >= 0 && p <= max_size, "erl_replay_write_rows_f32: add=%lld p=%lld max_size=%lld",
                (long long)add, (long long)p, (long long)max_size);
    if (add == 0) return ERL_OK;
    const int RW = (int)erl_replay_row_floats(S, A);
    const int g = grid_for(add * num_seqs * (S + A + 3));
    hipStream_t st = (hipStream_t)stream;
    if (flag_is_f32)
        hipLaunchKernelGGL((replay_write_rows_kernel<true>), dim3(g), dim3(256), 0, st, ring, states, actions, rewards, undones, unmasks,
                           max_size, num_seqs, S, A, RW, p, add);
    else
        hipLaunchKernelGGL((replay_write_rows_kernel<false>), dim3(g), dim3(256), 0, st, ring, states, actions, rewards, undones, unmasks,
                           max_size, num_seqs, S, A, RW, p, add);
    return erl_hip_status(hipGetLastError(), "erl_replay_write_rows_f32");
}

extern "C" int erl_replay_sample_rows_f32(const float *ring, int64_t max_size, int64_t num_seqs, int S, int A, const int64_t *ids, int64_t B,
                                          int64_t sample_len, float *out_state, float *out_action, float *out_reward, float *out_undone,
                                          float *out_unmask, float *out_next_state, int64_t *out_ids0, int64_t *out_ids1, void *stream)
{
    ERL_REQUIRE(ring && ids, "erl_replay_sample_rows_f32: NULL tensor");
    ERL_REQUIRE(out_state && out_action && out_reward && out_undone && out_unmask && out_next_state, "erl_replay_sample_rows_f32: NULL output");
    ERL_REQUIRE((reinterpret_cast<uintptr_t>(ring) & 15) == 0, "erl_replay_sample_rows_f32: the ring must be 16-byte aligned");
    ERL_REQUIRE(num_seqs >= 1 && S >= 1 && A >= 1 && B >= 0 && sample_len >= 1 && sample_len <= max_size,
                "erl_replay_sample_rows_f32: bad shape (sample_len=%lld max_size=%lld)", (long long)sample_len, (long long)max_size);
    if (B == 0) return ERL_OK;
    const int RW = (int)erl_replay_row_floats(S, A);
    const int CW = RW / 4 + (S + 3) / 4;
    int64_t spw = B / 1024;                       // (as replay_sample_impl: a small batch is spread, one chunk per thread and one round trip)
    if (spw < 256 / CW) spw = 256 / CW;
    if (spw < 1) spw = 1;
    if (spw > RS_SAMPLES) spw = RS_SAMPLES;
    const int g = grid_for(erl_cdiv(B, spw) * 256);
    unsigned long long *sp = erl_span_slot(ERL_SPAN_REPLAY_SAMPLE, g);
    hipLaunchKernelGGL(replay_sample_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, ring, max_size, S, A, RW, ids, B, sample_len,
                       out_state, out_action, out_reward, out_undone, out_unmask, out_next_state, out_ids0, out_ids1, (int)spw, sp);
    return erl_hip_status(hipGetLastError(), "erl_replay_sample_rows_f32");
}
