// K6, one-wave-per-SIMD form, net [128,64] (templates: ppo_step_w4_impl.h)
#include "ppo_step_w4_impl.h"

int erl_ppo_w4_launch_128_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream) { return launch_w4_shape<4, 2>(g, n_slabs, vec, stream); }
