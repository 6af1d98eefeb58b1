// K6 for net_dims = (256, h2) (templates: ppo_step_wd_impl.h): instantiations, the scratch blocks and the entry point behind
// erl_ppo_step_f32 / erl_ppo_update_dp_f32 for this shape class.
#include "ppo_step_wd_args.h"
#include "ppo_step_wd.h"
#include "erl_common.h"

unsigned long long *erl_k6_timing_begin(hipStream_t stream);   // api.cpp (measurement hook, no-op unless enabled)
void erl_k6_timing_end(hipStream_t stream);

bool erl_ppo_wd_supported(int S, int h1, int h2, int A)
{
    return S >= 1 && S <= 64 && h1 == 256 && (h2 == 64 || h2 == 128) && A >= 1 && A <= 8;
}

namespace {
long long *g_wd_prof = nullptr;
int g_wd_prof_block = 0;
}  // namespace
void erl_ppo_wd_set_prof(long long *dev_buf, int block) { g_wd_prof = dev_buf; g_wd_prof_block = block; }

namespace {

// GELU'(z1) and H2 of every workgroup between the forward and the backward pass: library-owned, one block per (device, stream)
struct WdScratch {
    int device = -1;
    hipStream_t stream = nullptr;
    float *buf = nullptr;
    size_t floats = 0;
};
WdScratch g_wd_scratch[16];

int wd_scratch(size_t floats, hipStream_t stream, float **out)
{
    int dev = 0;
    int rc = erl_hip_status(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    WdScratch *slot = nullptr;
    for (auto &s : g_wd_scratch)
        if (s.buf && s.device == dev && s.stream == stream) slot = &s;
    if (!slot)
        for (auto &s : g_wd_scratch)
            if (!s.buf) { slot = &s; break; }
    ERL_REQUIRE(slot, "erl_ppo_step_f32 (wide): more than 16 (device, stream) pairs");
    if (slot->floats < floats) {
        if (slot->buf) {
            if ((rc = erl_hip_status(hipStreamSynchronize(slot->stream), "hipStreamSynchronize"))) return rc;
            if ((rc = erl_hip_status(hipFree(slot->buf), "hipFree"))) return rc;
            slot->buf = nullptr;
            slot->floats = 0;
        }
        if ((rc = erl_hip_status(hipMalloc((void **)&slot->buf, floats * sizeof(float)), "hipMalloc(wide-net scratch)"))) return rc;
        slot->floats = floats;
    }
    slot->device = dev;
    slot->stream = stream;
    *out = slot->buf;
    return ERL_OK;
}

}  // namespace

int erl_ppo_wd_step(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std, const float *cri_avg,
                    const float *cri_std, int S, int h1, int h2, int A, const float *states, const float *actions, const uint8_t *unmasks,
                    const float *logprobs, const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids,
                    int64_t B, float ratio_clip, float lambda_entropy, float inv_batch, int objective, float *slabs, int n_slabs,
                    int64_t stride, const S3Images *images, const double *adv_stats, void *stream)
{
    ERL_REQUIRE(erl_ppo_wd_supported(S, h1, h2, A), "erl_ppo_step_f32 (wide): unsupported dims S=%d net=[%d,%d] A=%d", S, h1, h2, A);
    hipStream_t st = (hipStream_t)stream;
    S3Images own{};
    if (!images || !images->net[0].img || !images->net[1].img || !images->net[0].img1 || !images->net[1].img1) {
        // stand-alone call: the images are built here (the update loop builds them once and lets clip + Adam keep them current);
        // the networks must be the two halves of ONE flat block [actor | critic] for that
        const Dims da{S, h1, h2, A};
        ERL_REQUIRE(critic_params == actor_params + da.count(true),
                    "erl_ppo_step_f32 (wide): actor and critic parameters must be contiguous ([actor | critic]) for net [%d,%d]", h1, h2);
        int rc = erl_s3_images_build(actor_params, S, h1, h2, A, &own, nullptr, 0, H, N, nullptr, st);
        if (rc) return rc;
        images = &own;
    }
    PpoWdArgs a;
    Ppo2Args &g = a.g;
    g.P[0] = actor_params; g.P[1] = critic_params;
    g.avg[0] = act_avg; g.avg[1] = cri_avg;
    g.sd[0] = act_std; g.sd[1] = cri_std;
    g.states = states; g.actions = actions; g.logprobs = logprobs; g.advantages = advantages; g.reward_sums = reward_sums;
    g.unmasks = unmasks; g.ids = ids;
    g.H = H; g.N = N; g.B = B;
    g.S = S; g.h1 = h1; g.h2 = h2; g.A = A;
    g.ratio_clip = ratio_clip; g.lambda_entropy = lambda_entropy; g.inv_batch = inv_batch;
    g.objective = objective;
    g.slabs = slabs;
    g.Pa = Dims{S, h1, h2, A}.count(true);
    g.Pc = Dims{S, h1, h2, 1}.count(false);
    g.stride = stride;
    g.adv_stats = adv_stats;
    g.w2img[0] = images->net[0].img; g.w2img[1] = images->net[1].img;
    g.w1img[0] = images->net[0].img1; g.w1img[1] = images->net[1].img1;
    g.prof = g_wd_prof;
    g.prof_block = g_wd_prof_block;
    const int N2 = h2 / 32;
    int rc = wd_scratch((size_t)n_slabs * 2 * wd_scratch_floats(N2), st, &a.scratch);
    if (rc) return rc;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (S % 4 == 0) && al(actor_params) && al(critic_params) && al(states);
    g.span = erl_k6_timing_begin(st);
    if (h2 == 128) rc = S > 32 ? erl_ppo_wd_launch_24(a, n_slabs, vec, st) : erl_ppo_wd_launch_14(a, n_slabs, vec, st);
    else rc = S > 32 ? erl_ppo_wd_launch_22(a, n_slabs, vec, st) : erl_ppo_wd_launch_12(a, n_slabs, vec, st);
    erl_k6_timing_end(st);
    return rc;
}
