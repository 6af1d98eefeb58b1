// K6 on the bf16 matrix pipe (ppo_step_s3_impl.h): the [128,64] instantiations (stand-alone and with W2 images).
#include "ppo_step_s3_impl.h"

int erl_ppo_s3_launch_128_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream) { return launch_s3_shape<4, 2, false>(g, n_slabs, vec, stream); }
int erl_ppo_s3_launch_pre_128_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream) { return launch_s3_shape<4, 2, true>(g, n_slabs, vec, stream); }
