// K6 for net_dims = (256, h2) (ppo_step_wd.hip): what ppo_step.hip / comm.cpp / grad_tail.hip need to know about it.
#pragma once
#include <stdint.h>
#include "s3_image.h"

bool erl_ppo_wd_supported(int S, int h1, int h2, int A);      // h1 == 256, h2 in {64, 128}, S <= 64, A <= 8

// the W2 image of this shape class: four column-quarter images [h2][3][64 bf16], contiguous (quarter q at q * h2 * 384 bytes), each
// in s3_image.h's layout with K = 64 -- the minibatch kernel streams them through LDS one at a time
__device__ inline void s3_image_put_w2(unsigned char *img, int h1, int h2, int row, int col, float x)
{
    if (h1 == 256) s3_image_put(img + (size_t)(col >> 6) * ((size_t)h2 * 384), 64, row, col & 63, x);
    else s3_image_put(img, h1, row, col, x);
}

// erl_ppo_step_images_f32 (s3_image.h) for this shape class; images == nullptr: built inside (actor and critic parameters must then be
// the two halves of one flat block)
int erl_ppo_wd_step(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std, const float *cri_avg,
                    const float *cri_std, int S, int h1, int h2, int A, const float *states, const float *actions, const uint8_t *unmasks,
                    const float *logprobs, const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                    int64_t B, float ratio_clip, float lambda_entropy, float inv_batch, int objective, float *slabs, int n_slabs,
                    int64_t stride, const S3Images *images, const double *adv_stats, void *stream);

// ERL_PROFILE builds: where the kernel's cycle stamps go ([net][8 waves][32] int64), which workgroup stamps
void erl_ppo_wd_set_prof(long long *dev_buf, int block);

// net_dims = (256, 128, 64 | 128) (dims = [S, 256, 128, h3, A], S <= 64, A <= 8): the fused minibatch step behind erl_mlpn_ppo_step_f32
// (ERL_WIDE_FUSED=0 keeps the layered step); flat_grad receives [actor | critic | 3 objectives, 0]
bool erl_ppo_wd3_supported(const int *dims, int n_dims);
int erl_ppo_wd3_step(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std, const float *cri_avg,
                     const float *cri_std, const int *dims, const float *states, const float *actions, const uint8_t *unmasks,
                     const float *logprobs, const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B,
                     float ratio_clip, float lambda_entropy, float inv_batch, int objective, float *flat_grad, void *stream);
