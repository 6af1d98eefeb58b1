// The weight images of ppo_step_s3 (K6 on the bf16 matrix pipe) as the optimiser sees them: global-memory copies of the kernel's LDS
// images [row][3 parts][K bf16] (16-byte chunks XOR-swizzled by the row, see ppo_step_s3_impl.h) -- W2 ([h2][3][h1]) and, since
// round 4, W1 ([h1][3][K1], K1 = the state dimension padded to 32 or 64 columns, pad = 0) -- built once per update loop from the
// fp32 parameters and refreshed by clip + Adam element by element, so that the minibatch kernel copies them straight into LDS
// (LDS-DMA: W1 while the sample ids are in flight, W2 under the first layer's MFMAs) instead of every workgroup splitting the same
// 24k weights (round 4's prologue profile: 2.4k of the kernel's first 11.7k cycles were that split).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct S3Image {
    unsigned char *img;     // W2 image; nullptr: none
    int64_t w2_off;         // offset of W2 inside the network's flat parameter block
    int h1, h2;
    unsigned char *img1;    // W1 image [h1][3][K1 bf16] (W1 sits at offset 0 of the block, row-major [h1][S]); nullptr: none
    int S, K1;              // real / padded columns of W1 (K1 = 32 for S <= 32, else 64)
};
struct S3Images {
    S3Image net[2];         // actor, critic (= parameter groups 0, 1 of the update loop)
    // round 6: the per-sample values of a minibatch row as ONE 64-byte record per buffer row, [a_0 .. a_7 | logprob, advantage (raw),
    // reward_sum, unmask (0 / 1) | pad], built with the images once per update loop: a minibatch kernel's lane fetches its sample's
    // scalars with one or two 16-byte loads from ONE line instead of four gathers from four arrays (the address processing of those
    // gathers was a third of the actor workgroups' prologue); nullptr: none
    const float *aux = nullptr;
};
// what the records are built from (the rollout buffers of the update loop; rows = H * N)
struct S3AuxSrc {
    const float *actions, *logprobs, *advantages, *reward_sums;
    const uint8_t *unmasks;
    int A;
    int64_t rows;
};
constexpr int kS3AuxFloats = 16;

// chunk swizzle of an image with CP = K / 8 chunks per part (the formulas of swz<> in ppo_step_s3_impl.h)
__host__ __device__ inline int s3_swz(int CP, int r)
{
    const int r0 = r & 1, r1 = (r >> 1) & 1, r2 = (r >> 2) & 1, r3 = (r >> 3) & 1;
    if (CP == 16) return ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2;
    if (CP == 8) return (r1 << 2) | (r2 << 1) | (r0 ^ r3);
    return (r2 << 1) | (r1 ^ r3);
}

__device__ inline unsigned short s3_bf16_rne(float x)
{
    typedef __bf16 bf16x2_s3 __attribute__((ext_vector_type(2)));
    typedef float f32x2_s3 __attribute__((ext_vector_type(2)));
    return (unsigned short)(__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_s3{x, 0.f}, bf16x2_s3)) & 0xffffu);
}

// element (row, col) of a K-column matrix into its image: x = h + m + l exactly, each a bf16 (round to nearest even)
__device__ inline void s3_image_put(unsigned char *img, int K, int row, int col, float x)
{
    const int CP = K >> 3;
    const unsigned short h = s3_bf16_rne(x);
    const float r = x - __uint_as_float((uint32_t)h << 16);
    const unsigned short m = s3_bf16_rne(r);
    const float q = r - __uint_as_float((uint32_t)m << 16);
    const unsigned short l = s3_bf16_rne(q);
    unsigned char *p = img + (size_t)row * (48 * CP) + 16 * ((col >> 3) ^ s3_swz(CP, row)) + 2 * (col & 7);
    *reinterpret_cast<unsigned short *>(p) = h;
    *reinterpret_cast<unsigned short *>(p + 16 * CP) = m;
    *reinterpret_cast<unsigned short *>(p + 32 * CP) = l;
}

inline size_t s3_image_bytes(int h1, int h2) { return (size_t)h2 * 6 * h1; }
inline int s3_image_k1(int S) { return S <= 32 ? 32 : 64; }
inline size_t s3_image1_bytes(int h1, int S) { return (size_t)h1 * 6 * s3_image_k1(S); }

// ppo_step.hip / grad_tail.hip: the entry points of the update loop (comm.cpp) that carry the images along
// the arithmetic a call gets: its mode word's arith bits (ERL_PPO_MODE), else the process-wide default (ppo_step.hip)
int erl_ppo_arith_for_call(int S, int h1, int h2, int A, int arith_call);
// the next minibatch-kernel launch on the current device is the first of an update loop (ppo_step.hip: code touch on slow-fetch devices)
void erl_k6_touch_next_launch();
int erl_ppo_step_images_f32(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std,
                            const float *cri_avg, const float *cri_std, int S, int h1, int h2, int A, const float *states,
                            const float *actions, const uint8_t *unmasks, const float *logprobs, const float *advantages,
                            const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B, float ratio_clip,
                            float lambda_entropy, float inv_batch, int objective, float *slabs, int n_slabs, const S3Images *images,
                            const double *adv_stats, const int64_t *next_ids, void *stream, int only_net = -1);
// the two-launch tail's work in ONE launch (grad_tail.hip, tail_fused_kernel; single process) and whether a row of `stride` floats can take it
extern "C" int erl_tail_fused_ok(int64_t stride);
int erl_tail_fused_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len, int n_groups,
                       float grad_scale, float *params, float *exp_avg, float *exp_avg_sq, int32_t step, float lr, float beta1, float beta2,
                       float eps, float max_norm, const S3Images *images, hipStream_t stream);
int erl_clip_adam_partials_images_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride,
                                      const int64_t *group_off, const int64_t *group_len, int n_groups, int32_t step, float lr, float beta1,
                                      float beta2, float eps, float max_norm, float grad_scale, const S3Images *images, const uint32_t *poison,
                                      void *stream);
// one parameter group's share of the two-launch tail (grad_tail.hip; the two-chain update loop of comm.cpp)
int erl_launch_reduce_group_f32(const float *slabs, int n_slabs, int64_t stride, float *out, const int64_t *off, const int64_t *len, int n_groups,
                                int group, float grad_scale, hipStream_t stream);
int erl_clip_adam_partials_group_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t stride, const int64_t *group_off,
                                     const int64_t *group_len, int n_groups, int group, int32_t step, float lr, float beta1, float beta2, float eps,
                                     float max_norm, float grad_scale, const S3Images *images, void *stream);
// the workgroup map the current device keeps for a minibatch-kernel family (ppo_step.hip): -1 not measured yet
int erl_k6_wg_map_choice(int family);
// grad_tail.hip: library-owned image buffers of (device, stream), built from the flat parameters [actor | critic]
// (adv_partials != nullptr: one more block of the same launch folds the n_partials x 3 fp64 partial sums of the rollout's advantage
// epilogue into adv_stats -- erl_adv_stats_fold_f32 without a launch of its own)
int erl_s3_images_build(const float *flat_params, int S, int h1, int h2, int A, S3Images *out, const double *adv_partials, int n_partials,
                        int64_t H, int64_t N, double *adv_stats, hipStream_t stream, const S3AuxSrc *aux_src = nullptr);
