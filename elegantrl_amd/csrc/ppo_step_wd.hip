// K6 for net_dims = (256, h2) (templates: ppo_step_wd_impl.h): instantiations, the scratch blocks and the entry point behind
// erl_ppo_step_f32 / erl_ppo_update_dp_f32 for this shape class.
#include "ppo_step_wd_args.h"
#include "ppo_step_wd.h"
#include "erl_common.h"
#include "../../include/erl_hip.h"
#include <cstdlib>

unsigned long long *erl_k6_timing_begin(hipStream_t stream, int n_slabs);   // api.cpp (measurement hook, no-op unless enabled; grid = (n_slabs, 2))
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
    g.next_ids = nullptr;
    g.exp_net = -1; g.wg_map = 0;
    g.prof_block = g_wd_prof_block;
    a.w3img[0] = a.w3img[1] = nullptr;
    a.h3 = 0;
    const int N2 = h2 / 32;
    int rc = wd_scratch((size_t)n_slabs * 2 * wd_scratch_floats(N2, 0), st, &a.scratch);
    if (rc) return rc;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (S % 4 == 0) && al(actor_params) && al(critic_params) && al(states);
    auto go = [&](const PpoWdArgs &x) {
        if (h2 == 128) return S > 32 ? erl_ppo_wd_launch_24(x, n_slabs, vec, st) : erl_ppo_wd_launch_14(x, n_slabs, vec, st);
        return S > 32 ? erl_ppo_wd_launch_22(x, n_slabs, vec, st) : erl_ppo_wd_launch_12(x, n_slabs, vec, st);
    };
    g.span = nullptr;           // (one network per XCD or both on every XCD: ppo_step.h k6_wg_map; these kernels are 125 KB per network)
    g.wg_map = erl_k6_wg_map_for_launch(1, n_slabs, st, [&](int m) { PpoWdArgs t = a; t.g.wg_map = m; return go(t); });
    g.span = erl_k6_timing_begin(st, n_slabs);
    rc = go(a);
    erl_k6_timing_end(st);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// net_dims = (256, 128, 64 | 128): the reference's other two demo networks (examples/demo_A2C_PPO.py:171, :224).  Reached through
// erl_mlpn_ppo_step_f32 (mlpn.hip), whose contract is the summed gradient row [actor | critic | 3 objectives, 0]: the weight images, the
// scratch blocks and the slabs are library-owned per (device, stream); the slab reduction runs here.
// ---------------------------------------------------------------------------------------------------------------------------------
bool erl_ppo_wd3_supported(const int *dims, int n_dims)
{
    static const bool on = [] { const char *e = getenv("ERL_WIDE_FUSED"); return !(e && e[0] == '0'); }();
    return on && n_dims == 5 && dims[0] >= 1 && dims[0] <= 64 && dims[1] == 256 && dims[2] == 128 && (dims[3] == 64 || dims[3] == 128) &&
           dims[4] >= 1 && dims[4] <= 8;
}

namespace {

struct Wd3Imgs {
    unsigned char *w1[2], *w2[2], *w3[2];
};

// W1 image [256][3][K1] (pad columns 0), W2 as four column-quarter images, W3 image [h3][3][128], for both networks: one thread per element
__global__ __launch_bounds__(256) void wd3_images_kernel(const float *__restrict__ Pa_, const float *__restrict__ Pc_, int S, int K1, int h3, Wd3Imgs im)
{
    const int net = blockIdx.y;
    const float *P = net ? Pc_ : Pa_;
    const int64_t n1 = 256 * (int64_t)K1, n2 = 128 * 256, n3 = (int64_t)h3 * 128;
    const int64_t oW2 = 256 * (int64_t)S + 256, oW3 = oW2 + n2 + 128;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n1) {
        const int row = (int)(e / K1), col = (int)(e % K1);
        s3_image_put(im.w1[net], K1, row, col, col < S ? P[(int64_t)row * S + col] : 0.f);
    } else if ((e -= n1) < n2) {
        s3_image_put_w2(im.w2[net], 256, 128, (int)(e / 256), (int)(e % 256), P[oW2 + e]);
    } else if ((e -= n2) < n3) {
        s3_image_put(im.w3[net], 128, (int)(e / 128), (int)(e % 128), P[oW3 + e]);
    }
}

struct Wd3Bufs {
    int device = -1;
    hipStream_t stream = nullptr;
    unsigned char *buf = nullptr;
    size_t bytes = 0;
};
Wd3Bufs g_wd3[16];

}  // namespace

int erl_ppo_wd3_step(const float *actor_params, const float *critic_params, const float *act_avg, const float *act_std, const float *cri_avg,
                     const float *cri_std, const int *dims, const float *states, const float *actions, const uint8_t *unmasks,
                     const float *logprobs, const float *advantages, const float *reward_sums, int64_t H, int64_t N, const int64_t *ids, int64_t B,
                     float ratio_clip, float lambda_entropy, float inv_batch, int objective, float *flat_grad, void *stream)
{
    const int S = dims[0], h3 = dims[3], A = dims[4], N3 = h3 / 32;
    hipStream_t st = (hipStream_t)stream;
    auto count = [&](int out, bool sd) { return 256 * (int64_t)S + 256 + 128 * 256 + 128 + (int64_t)h3 * 128 + h3 + (int64_t)out * h3 + out + (sd ? out : 0); };
    const int64_t Pa = count(A, true), Pc = count(1, false), stride = Pa + Pc + 4;
    const int n_slabs = (int)erl_cdiv(B, PB);
    const int K1 = s3_image_k1(S);
    auto up = [](size_t x) { return (x + 1023) / 1024 * 1024; };
    const size_t b1 = up((size_t)256 * 6 * K1), b2 = up((size_t)128 * 6 * 256), b3 = up((size_t)h3 * 6 * 128);
    const size_t img_bytes = 2 * (b1 + b2 + b3), slab_bytes = up((size_t)n_slabs * stride * 4);
    const size_t scr_bytes = (size_t)n_slabs * 2 * wd_scratch_floats(N3, N3) * 4, need = img_bytes + slab_bytes + scr_bytes;
    int dev = 0;
    int rc = erl_hip_status(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    Wd3Bufs *slot = nullptr;
    for (auto &s : g_wd3)
        if (s.buf && s.device == dev && s.stream == st) slot = &s;
    if (!slot)
        for (auto &s : g_wd3)
            if (!s.buf) { slot = &s; break; }
    ERL_REQUIRE(slot, "erl_mlpn_ppo_step_f32 (256,128,h3): more than 16 (device, stream) pairs");
    if (slot->bytes < need) {
        if (slot->buf) {
            if ((rc = erl_hip_status(hipStreamSynchronize(slot->stream), "hipStreamSynchronize"))) return rc;
            if ((rc = erl_hip_status(hipFree(slot->buf), "hipFree"))) return rc;
            slot->buf = nullptr;
            slot->bytes = 0;
        }
        if ((rc = erl_hip_status(hipMalloc((void **)&slot->buf, need), "hipMalloc(wide-net buffers)"))) return rc;
        slot->bytes = need;
    }
    slot->device = dev;
    slot->stream = st;
    Wd3Imgs im;
    unsigned char *p = slot->buf;
    for (int n = 0; n < 2; ++n) { im.w1[n] = p; p += b1; im.w2[n] = p; p += b2; im.w3[n] = p; p += b3; }
    float *slabs = reinterpret_cast<float *>(p);
    float *scratch = reinterpret_cast<float *>(p + slab_bytes);
    const int64_t elems = 256 * (int64_t)K1 + 128 * 256 + (int64_t)h3 * 128;
    hipLaunchKernelGGL(wd3_images_kernel, dim3((unsigned)erl_cdiv(elems, 256), 2), dim3(256), 0, st, actor_params, critic_params, S, K1, h3, im);
    if ((rc = erl_hip_status(hipGetLastError(), "wd3_images_kernel"))) return rc;
    PpoWdArgs a;
    Ppo2Args &g = a.g;
    g.P[0] = actor_params; g.P[1] = critic_params;
    g.avg[0] = act_avg; g.avg[1] = cri_avg;
    g.sd[0] = act_std; g.sd[1] = cri_std;
    g.states = states; g.actions = actions; g.logprobs = logprobs; g.advantages = advantages; g.reward_sums = reward_sums;
    g.unmasks = unmasks; g.ids = ids;
    g.H = H; g.N = N; g.B = B;
    g.S = S; g.h1 = 256; g.h2 = 128; g.A = A;
    g.ratio_clip = ratio_clip; g.lambda_entropy = lambda_entropy; g.inv_batch = inv_batch;
    g.objective = objective;
    g.slabs = slabs;
    g.Pa = Pa; g.Pc = Pc; g.stride = stride;
    g.adv_stats = nullptr;
    g.w2img[0] = im.w2[0]; g.w2img[1] = im.w2[1];
    g.w1img[0] = im.w1[0]; g.w1img[1] = im.w1[1];
    g.prof = g_wd_prof;
    g.next_ids = nullptr;
    g.exp_net = -1; g.wg_map = 0;
    g.prof_block = g_wd_prof_block;
    a.w3img[0] = im.w3[0]; a.w3img[1] = im.w3[1];
    a.h3 = h3;
    a.scratch = scratch;
    auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = (S % 4 == 0) && al(actor_params) && al(critic_params) && al(states);
    auto go = [&](const PpoWdArgs &x) {
        if (h3 == 128) return S > 32 ? erl_ppo_wd3_launch_24(x, n_slabs, vec, st) : erl_ppo_wd3_launch_14(x, n_slabs, vec, st);
        return S > 32 ? erl_ppo_wd3_launch_22(x, n_slabs, vec, st) : erl_ppo_wd3_launch_12(x, n_slabs, vec, st);
    };
    g.span = nullptr;
    g.wg_map = erl_k6_wg_map_for_launch(1, n_slabs, st, [&](int m) { PpoWdArgs t = a; t.g.wg_map = m; return go(t); });
    g.span = erl_k6_timing_begin(st, n_slabs);
    rc = go(a);
    erl_k6_timing_end(st);
    if (rc) return rc;
    return erl_grad_reduce_f32(slabs, n_slabs, stride, flat_grad, stream);
}
