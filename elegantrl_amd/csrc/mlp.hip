// K2 value pre-pass, K1 rollout step, slab reduction of K6 (the PPO minibatch kernel itself: ppo_step.hip).  gfx950 / fp32 MFMA.
// Building blocks and the LDS/MFMA formulation are documented in mlp_tiles.h.
#include "mlp_tiles.h"

namespace {

struct MlpDims {
    int S, h1, h2, out;
    __host__ __device__ int Sc() const { return (S + 31) & ~31; }
    __host__ __device__ int64_t oW1() const { return 0; }
    __host__ __device__ int64_t ob1() const { return (int64_t)h1 * S; }
    __host__ __device__ int64_t oW2() const { return ob1() + h1; }
    __host__ __device__ int64_t ob2() const { return oW2() + (int64_t)h2 * h1; }
    __host__ __device__ int64_t oW3() const { return ob2() + h2; }
    __host__ __device__ int64_t ob3() const { return oW3() + (int64_t)out * h2; }
    __host__ __device__ int64_t oStd() const { return ob3() + out; }
    __host__ __device__ int64_t count(bool with_std) const { return oStd() + (with_std ? out : 0); }
};

bool dims_ok(int S, int h1, int h2, int out)
{
    return S >= 1 && S <= ERL_MAX_STATE_DIM && h1 >= 32 && h1 <= ERL_MAX_HIDDEN && (h1 % 32) == 0 && h2 >= 32 &&
           h2 <= ERL_MAX_HIDDEN && (h2 % 32) == 0 && out >= 1 && out <= ERL_MAX_ACTION_DIM;
}

constexpr float kLogSqrt2Pi = 0.91893853320467274178f;  // log(sqrt(2 pi))

// ---------------------------------------------------------------------------------------------
// forward-only kernels (K2: critic values, K1: actor rollout step)
// LDS: XT [Sc][LD] | H1T [h1][LD] | H2T [h2][LD] | YT [16][LD] | ACT [16][LD]
// ---------------------------------------------------------------------------------------------
template <int M>
__host__ __device__ inline size_t fwd_lds_floats(const MlpDims &d)
{
    return (size_t)(d.Sc() + d.h1 + d.h2 + 32) * (M + 1);
}

template <int M, int NW>
__global__ __launch_bounds__(NW * 64) void value_forward_kernel(const float *__restrict__ P, const float *__restrict__ avg,
                                                                const float *__restrict__ sd, MlpDims d,
                                                                const float *__restrict__ states, int64_t rows,
                                                                float *__restrict__ values)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = M + 1;
    float *XT = smem, *H1T = XT + d.Sc() * LD, *H2T = H1T + d.h1 * LD, *YT = H2T + d.h2 * LD;
    const int64_t ntiles = (rows + M - 1) / M;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * M;
        const int valid = (int)min((int64_t)M, rows - row0);
        gather_states<M>(XT, d.Sc(), d.S, states, avg, sd, valid, [&](int m) { return row0 + m; }, nullptr, 0);
        __syncthreads();
        layer_forward<M, NW, true, false>(P + d.oW1(), P + d.ob1(), d.h1, d.S, XT, H1T, nullptr);
        __syncthreads();
        layer_forward<M, NW, true, false>(P + d.oW2(), P + d.ob2(), d.h2, d.h1, H1T, H2T, nullptr);
        __syncthreads();
        output_layer<M>(P + d.oW3(), P + d.ob3(), 1, d.h2, H2T, YT);
        __syncthreads();
        if ((int)threadIdx.x < valid) values[row0 + threadIdx.x] = YT[threadIdx.x];
        __syncthreads();
    }
}

template <int M, int NW>
__global__ __launch_bounds__(NW * 64) void rollout_step_kernel(const float *__restrict__ P, const float *__restrict__ avg,
                                                               const float *__restrict__ sd, MlpDims d,
                                                               const float *__restrict__ state, int64_t N,
                                                               const float *__restrict__ noise, uint64_t seed,
                                                               uint64_t counter, float *__restrict__ o_state,
                                                               float *__restrict__ o_action, float *__restrict__ o_logprob,
                                                               float *__restrict__ o_env)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = M + 1;
    float *XT = smem, *H1T = XT + d.Sc() * LD, *H2T = H1T + d.h1 * LD, *YT = H2T + d.h2 * LD, *ACT = YT + 16 * LD;
    const int A = d.out;
    const float *std_log = P + d.oStd();
    const int64_t ntiles = (N + M - 1) / M;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * M;
        const int valid = (int)min((int64_t)M, N - row0);
        gather_states<M>(XT, d.Sc(), d.S, state, avg, sd, valid, [&](int m) { return row0 + m; }, o_state, row0);
        __syncthreads();
        layer_forward<M, NW, true, false>(P + d.oW1(), P + d.ob1(), d.h1, d.S, XT, H1T, nullptr);
        __syncthreads();
        layer_forward<M, NW, true, false>(P + d.oW2(), P + d.ob2(), d.h2, d.h1, H1T, H2T, nullptr);
        __syncthreads();
        output_layer<M>(P + d.oW3(), P + d.ob3(), A, d.h2, H2T, YT);
        __syncthreads();
        // sample: a = mean + std * eps  (torch.normal(mean, std));  element order (row, a) so that the
        // global reads of `noise` and the writes of the action rows are coalesced.
        for (int e = threadIdx.x; e < valid * A; e += blockDim.x) {
            const int m = e / A, a = e - m * A;
            const int64_t row = row0 + m;
            const float eps = noise ? noise[row * A + a] : philox_normal(seed, counter, (uint32_t)row, (uint32_t)a);
            const float sdv = expf(std_log[a]);
            const float act = YT[a * LD + m] + sdv * eps;
            ACT[a * LD + m] = act;
            if (o_action) o_action[row * A + a] = act;
            if (o_env) o_env[row * A + a] = tanhf(act);
        }
        __syncthreads();
        if ((int)threadIdx.x < valid && o_logprob) {
            const int m = threadIdx.x;
            float lp = 0.f;
            for (int a = 0; a < A; ++a) {  // Normal.log_prob: -(x-mu)^2/(2 var) - log(std) - log(sqrt(2 pi))
                const float sl = std_log[a], sdv = expf(sl), var = sdv * sdv;
                const float diff = ACT[a * LD + m] - YT[a * LD + m];
                lp += -(diff * diff) / (2.f * var) - logf(sdv) - kLogSqrt2Pi;
            }
            o_logprob[row0 + m] = lp;
        }
        __syncthreads();
    }
}

// sum the per-workgroup slabs into the flat gradient.  Deterministic: element e is summed by 4 threads (slab
// quarter p = threadIdx.x / 64 takes slabs k = p (mod 4)... in ascending order, 8 loads in flight), combined in a
// fixed order through LDS.  Latency bound (26 MB, 128 strided rows): the split buys 4x the loads in flight.
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float *__restrict__ slabs, int n_slabs, int64_t stride,
                                                          float *__restrict__ flat)
{
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + el;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < stride) {
        const float *src = slabs + i;
        int k = p;
        for (; k + 28 < n_slabs; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += src[(size_t)(k + 4 * u) * stride];
        }
        for (; k < n_slabs; k += 4) s[0] += src[(size_t)k * stride];
    }
    part[p][el] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (p == 0 && i < stride) flat[i] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

// ---------------------------------------------------------------------------------------------
// MFMA helper self-test: C (32x32) = A (32xK) . B (Kx32) through the same lane mapping as layer_forward
// ---------------------------------------------------------------------------------------------
__global__ void selftest_kernel(const float *A, const float *B, int K, float *C)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 8) {
        const int kb = k0 + 4 * hi;
        for (int j = 0; j < 4; ++j) acc = mfma32(A[l31 * K + kb + j], B[(kb + j) * 32 + l31], acc);
    }
    for (int r = 0; r < 16; ++r) C[crow(r, hi) * 32 + l31] = acc[r];
}

template <typename Kern>
int set_lds(Kern kern, size_t bytes)
{
    if (bytes <= 64 * 1024) return ERL_OK;
    return erl_hip_status(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                          "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace

extern "C" int64_t erl_mlp_param_count(int S, int h1, int h2, int out, int with_std_log)
{
    if (!dims_ok(S, h1, h2, out)) return -1;
    return MlpDims{S, h1, h2, out}.count(with_std_log != 0);
}

extern "C" int erl_value_forward_f32(const float *critic_params, const float *state_avg, const float *state_std, int S, int h1,
                                     int h2, const float *states, int64_t rows, float *values, void *stream)
{
    ERL_REQUIRE(critic_params && state_avg && state_std && states && values, "erl_value_forward_f32: NULL tensor");
    ERL_REQUIRE(dims_ok(S, h1, h2, 1), "erl_value_forward_f32: unsupported dims S=%d net=[%d,%d]", S, h1, h2);
    ERL_REQUIRE(rows >= 0, "erl_value_forward_f32: rows < 0");
    if (rows == 0) return ERL_OK;
    MlpDims d{S, h1, h2, 1};
    constexpr int M = 64, NW = 8;
    const size_t lds = fwd_lds_floats<M>(d) * sizeof(float);
    int rc = set_lds(value_forward_kernel<M, NW>, lds);
    if (rc) return rc;
    int64_t tiles = erl_cdiv(rows, M);
    const int grid = (int)(tiles < num_cus() ? tiles : num_cus());
    hipLaunchKernelGGL((value_forward_kernel<M, NW>), dim3(grid), dim3(NW * 64), lds, (hipStream_t)stream, critic_params,
                       state_avg, state_std, d, states, rows, values);
    ERL_LAUNCH_CHECK("erl_value_forward_f32");
}

extern "C" int erl_rollout_step_f32(const float *actor_params, const float *state_avg, const float *state_std, int S, int h1,
                                    int h2, int A, const float *state, int64_t N, const float *noise, uint64_t seed,
                                    uint64_t counter, float *out_state_row, float *out_action_row, float *out_logprob_row,
                                    float *out_action_env, void *stream)
{
    ERL_REQUIRE(actor_params && state_avg && state_std && state, "erl_rollout_step_f32: NULL tensor");
    ERL_REQUIRE(dims_ok(S, h1, h2, A), "erl_rollout_step_f32: unsupported dims S=%d net=[%d,%d] A=%d", S, h1, h2, A);
    ERL_REQUIRE(N >= 1, "erl_rollout_step_f32: N < 1");
    MlpDims d{S, h1, h2, A};
    constexpr int M = 32, NW = 4;
    const size_t lds = fwd_lds_floats<M>(d) * sizeof(float);
    int rc = set_lds(rollout_step_kernel<M, NW>, lds);
    if (rc) return rc;
    int64_t tiles = erl_cdiv(N, M);
    const int grid = (int)(tiles < 2 * num_cus() ? tiles : 2 * num_cus());
    hipLaunchKernelGGL((rollout_step_kernel<M, NW>), dim3(grid), dim3(NW * 64), lds, (hipStream_t)stream, actor_params, state_avg,
                       state_std, d, state, N, noise, seed, counter, out_state_row, out_action_row, out_logprob_row,
                       out_action_env);
    ERL_LAUNCH_CHECK("erl_rollout_step_f32");
}

extern "C" int erl_grad_reduce_f32(const float *slabs, int n_slabs, int64_t stride, float *flat_grad, void *stream)
{
    ERL_REQUIRE(slabs && flat_grad && n_slabs >= 1 && stride >= 1, "erl_grad_reduce_f32: bad argument");
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)erl_cdiv(stride, 64)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                       stride, flat_grad);
    ERL_LAUNCH_CHECK("erl_grad_reduce_f32");
}

extern "C" int erl_selftest_mfma(float *max_err)
{
    ERL_REQUIRE(max_err, "erl_selftest_mfma: NULL");
    const int K = 24;
    float hA[32 * K], hB[K * 32], hC[32 * 32];
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (float &x : hA) x = rnd();
    for (float &x : hB) x = rnd();
    float *dA, *dB, *dC;
    int rc;
    if ((rc = erl_hip_status(hipMalloc(&dA, sizeof(hA)), "hipMalloc"))) return rc;
    if ((rc = erl_hip_status(hipMalloc(&dB, sizeof(hB)), "hipMalloc"))) return rc;
    if ((rc = erl_hip_status(hipMalloc(&dC, sizeof(hC)), "hipMalloc"))) return rc;
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    rc = erl_hip_status(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost), "selftest memcpy");
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    if (rc) return rc;
    float worst = 0.f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[i * K + k] * hB[k * 32 + j];
            const float e = fabsf((float)ref - hC[i * 32 + j]);
            worst = e > worst ? e : worst;
        }
    *max_err = worst;
    return ERL_OK;
}
