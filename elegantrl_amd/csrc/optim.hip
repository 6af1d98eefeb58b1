// K7: global-norm gradient clip + Adam for up to 4 parameter groups in one launch.  gfx950.
// Replaces clip_grad_norm_ + torch.optim.Adam.step in AgentBase.optimizer_backward
// (elegantrl/agents/AgentBase.py:246-248; Adam built with defaults at AgentPPO.py:24-25).
//
// grid = (blocks_per_group, n_groups).  Each block first computes the group's full squared gradient
// norm on its own (a group is ~25k floats = 100 KB, L2 resident, so the redundant reads are cheaper
// than a second launch or a cross-block handshake, and the summation order is fixed => every block
// derives a bit-identical clip coefficient), then updates its slice.
#include "erl_common.h"

namespace {

struct AdamGroups {
    int64_t off[4], len[4];
};

__global__ __launch_bounds__(1024) void clip_adam_kernel(float *__restrict__ params, const float *__restrict__ grads,
                                                         float *__restrict__ m1, float *__restrict__ m2, AdamGroups gr,
                                                         const int32_t *__restrict__ step_base, int32_t step_offset, float lr,
                                                         float beta1, float beta2, float eps, float max_norm, float grad_scale,
                                                         float host_step_size, float host_bc2_sqrt)
{
    __shared__ double scratch[16];
    const int gi = blockIdx.y;
    const int64_t off = gr.off[gi], len = gr.len[gi];
    const float *g = grads + off;
    // this thread's element of the Adam phase: its four loads ride the same round trip as the norm's
    const int64_t per = (len + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = (lo + per < len) ? lo + per : len;
    const int64_t ie = lo + threadIdx.x;
    const bool own = ie < hi && per <= 1024;          // per > 1024 (very long groups): the loop at the end re-reads
    float e_g = 0.f, e_m1 = 0.f, e_m2 = 0.f, e_p = 0.f;
    if (own) { e_g = g[ie]; e_m1 = m1[off + ie]; e_m2 = m2[off + ie]; e_p = params[off + ie]; }
    double ss = 0.0;
    {   // every load of the thread is issued before the first use: ONE L2 round trip for groups up to 32 Ki elements
        constexpr int U = 32;
        for (int64_t i0 = threadIdx.x; i0 < len; i0 += (int64_t)U * 1024) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                x[u] = g[i < len ? i : len - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + (int64_t)u * 1024;
                const float xs = x[u] * grad_scale;
                if (i < len) ss += (double)xs * xs;
            }
        }
    }
    ss = block_sum(ss, scratch);
    const float total_norm = (float)sqrt(ss);
    float coef = max_norm / (total_norm + 1e-6f);   // clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    coef = coef > 1.f ? 1.f : coef;
    const float gmul = grad_scale * coef;

    float step_size = host_step_size, bc2_sqrt = host_bc2_sqrt;   // computed on the host when the step is a host value
    if (step_base) {
        const int step = *step_base + step_offset;
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }

    auto adam = [&](int64_t i, float graw, float m1v, float m2v, float pv) {
        const float gx = graw * gmul;
        const float a = m1v * beta1 + (1.f - beta1) * gx;          // exp_avg.lerp_(grad, 1 - beta1)
        const float b = m2v * beta2 + (1.f - beta2) * (gx * gx);   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        m1[off + i] = a;
        m2[off + i] = b;
        const float denom = sqrtf(b) / bc2_sqrt + eps;
        params[off + i] = pv - step_size * (a / denom);
    };
    if (per <= 1024) {
        if (own) adam(ie, e_g, e_m1, e_m2, e_p);
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) adam(i, g[i], m1[off + i], m2[off + i], params[off + i]);
    }
}

}  // namespace

extern "C" int erl_clip_adam_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const int64_t *group_off,
                                 const int64_t *group_len, int n_groups, const int32_t *step_base, int32_t step_offset, float lr,
                                 float beta1, float beta2, float eps, float max_norm, float grad_scale, void *stream)
{
    ERL_REQUIRE(params && grads && exp_avg && exp_avg_sq && group_off && group_len, "erl_clip_adam_f32: NULL argument");
    ERL_REQUIRE(n_groups >= 1 && n_groups <= 4, "erl_clip_adam_f32: n_groups must be 1..4");
    ERL_REQUIRE(step_base || step_offset >= 1, "erl_clip_adam_f32: Adam step must be >= 1");
    AdamGroups gr;
    int64_t longest = 0;
    for (int i = 0; i < 4; ++i) {
        gr.off[i] = i < n_groups ? group_off[i] : 0;
        gr.len[i] = i < n_groups ? group_len[i] : 0;
        ERL_REQUIRE(gr.off[i] >= 0 && gr.len[i] >= 0, "erl_clip_adam_f32: negative group bounds");
        if (gr.len[i] > longest) longest = gr.len[i];
    }
    int bx = (int)erl_cdiv(longest, 1024);   // one element per thread in the Adam phase (latency bound: no serial loop)
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (!step_base) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step_offset), bc2 = 1.0 - pow((double)beta2, (double)step_offset);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    hipLaunchKernelGGL(clip_adam_kernel, dim3(bx, n_groups), dim3(1024), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       gr, step_base, step_offset, lr, beta1, beta2, eps, max_norm, grad_scale, step_size, bc2_sqrt);
    ERL_LAUNCH_CHECK("erl_clip_adam_f32");
}

