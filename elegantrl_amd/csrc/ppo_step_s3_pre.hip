// K6 on the bf16 matrix pipe (ppo_step_s3_impl.h), W2 images supplied by the update loop: the [128,128] instantiations.
#include "ppo_step_s3_impl.h"

int erl_ppo_s3_launch_pre_128_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);      // ppo_step_s3_42.hip
int erl_ppo_s3_launch_pre_64_128(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);      // ppo_step_s3_24.hip
int erl_ppo_s3_launch_pre_64_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);       // ppo_step_s3_22.hip

int erl_ppo_s3_launch_pre(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream)
{
    if (g.h1 == 128 && g.h2 == 128) return launch_s3_shape<4, 4, true>(g, n_slabs, vec, stream);
    if (g.h1 == 128 && g.h2 == 64) return erl_ppo_s3_launch_pre_128_64(g, n_slabs, vec, stream);
    if (g.h1 == 64 && g.h2 == 128) return erl_ppo_s3_launch_pre_64_128(g, n_slabs, vec, stream);
    if (g.h1 == 64 && g.h2 == 64) return erl_ppo_s3_launch_pre_64_64(g, n_slabs, vec, stream);
    erl_set_error("erl_ppo_s3_launch_pre: unsupported net [%d,%d]", g.h1, g.h2);
    return ERL_EINVAL;
}
