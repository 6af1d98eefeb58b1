// K6 on the bf16 matrix pipe, fp32-equivalent through three-way operand splitting (templates: ppo_step_s3_impl.h): the [128,128]
// instantiations and the shape dispatch.
#include "ppo_step_s3_impl.h"

bool erl_ppo_s3_supported(int S, int h1, int h2, int A)
{
    return S >= 1 && S <= 64 && (h1 == 64 || h1 == 128) && (h2 == 64 || h2 == 128) && A >= 1 && A <= 8;
}

int erl_ppo_s3_launch_128_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);      // ppo_step_s3_42.hip
int erl_ppo_s3_launch_64_128(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);      // ppo_step_s3_24.hip
int erl_ppo_s3_launch_64_64(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream);       // ppo_step_s3_22.hip

int erl_ppo_s3_launch(const Ppo2Args &g, int n_slabs, bool vec, hipStream_t stream)
{
    if (g.h1 == 128 && g.h2 == 128) return launch_s3_shape<4, 4, false>(g, n_slabs, vec, stream);
    if (g.h1 == 128 && g.h2 == 64) return erl_ppo_s3_launch_128_64(g, n_slabs, vec, stream);
    if (g.h1 == 64 && g.h2 == 128) return erl_ppo_s3_launch_64_128(g, n_slabs, vec, stream);
    if (g.h1 == 64 && g.h2 == 64) return erl_ppo_s3_launch_64_64(g, n_slabs, vec, stream);
    erl_set_error("erl_ppo_s3_launch: unsupported net [%d,%d]", g.h1, g.h2);
    return ERL_EINVAL;
}
