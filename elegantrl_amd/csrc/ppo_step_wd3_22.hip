// K6 for net_dims = (256, 128, 64), 32 < S <= 64 (templates: ppo_step_wd_impl.h)
#include "ppo_step_wd_impl.h"

int erl_ppo_wd3_launch_22(const PpoWdArgs &a, int n_slabs, bool vec, hipStream_t stream)
{
    return vec ? launch_wd<2, 4, 2, true>(a, n_slabs, stream) : launch_wd<2, 4, 2, false>(a, n_slabs, stream);
}
