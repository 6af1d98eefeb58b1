// K6 on the bf16 matrix pipe (ppo_step_s3_impl.h): the [64,128] instantiations (stand-alone and with W2 images).
#include "ppo_step_s3_impl.h"

int erl_ppo_s3_launch_64_128(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream) { return launch_s3_shape<2, 4, false>(g, n_slabs, vec, stream); }
int erl_ppo_s3_launch_pre_64_128(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream) { return launch_s3_shape<2, 4, true>(g, n_slabs, vec, stream); }
