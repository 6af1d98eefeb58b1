/* Plain-C restatement of the reference's GAE / lambda-return backward scan.
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Used by tests/ as a second,
 * independently written checker and by bench.py's cpu_baseline leg ("port") for GAE GB/s.
 *
 * Follows elegantrl/agents/AgentPPO.py:207-232 (get_advantages) + :146 (reward_sums):
 *   trunc = !unmask;  r[trunc] += V(s_t) (= values[trunc]);  undone[trunc] = 0      (:211-214)
 *   m = undone * gamma                                                              (:216)
 *   v-trace branch (:223-227):  nv = r + m*nv;  A = (nv - v) + (m*lam)*A;  nv = v
 *   alt branch     (:228-231):  adv = r - v + m*A';  A' = v + lam*adv   (A' starts at 0)
 * Every product and sum is rounded separately (build with -ffp-contract=off) like the chain of
 * ATen ops in the reference.  Layout: time-major (H, N) row-major, flags are bytes (torch.bool).
 * Pinned by tests/test_oracle_golden.py against tests/golden/ppo_*.npz.
 */
#include <stddef.h>
#include <stdint.h>


/* returns 0; adv/ret are outputs; if mutate != 0 rewards/undones are updated in place like the reference */
int erl_oracle_gae_f32(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                       const float *next_value, float *adv, float *ret, int64_t H, int64_t N, float gamma,
                       float lam, int use_v_trace, int mutate)
{
    for (int64_t n = 0; n < N; ++n) {
        float nv = next_value[n];
        float a = 0.0f;
        for (int64_t t = H - 1; t >= 0; --t) {
            const size_t i = (size_t)t * (size_t)N + (size_t)n;
            float r = rewards[i];
            const float v = values[i];
            uint8_t ud = undones[i];
            if (!unmasks[i]) {
                r = r + v;
                ud = 0;
                if (mutate) { rewards[i] = r; undones[i] = 0; }
            }
            const float m = ud ? gamma : 0.0f;
            float out;
            if (use_v_trace) {
                const float mn = m * nv;
                nv = r + mn;
                const float d = nv - v;
                const float ml = m * lam;
                const float mla = ml * a;
                a = d + mla;
                out = a;
                nv = v;
            } else {
                const float d = r - v;
                const float ma = m * a;
                out = d + ma;
                const float la = lam * out;
                a = v + la;
            }
            adv[i] = out;
            ret[i] = out + v;
        }
    }
    return 0;
}

/* column-blocked variant used only for the CPU-baseline timing (same arithmetic, cache friendlier):
 * processes envs [n0, n1) so callers can split columns across threads. */
int erl_oracle_gae_f32_cols(float *rewards, uint8_t *undones, const uint8_t *unmasks, const float *values,
                            const float *next_value, float *adv, float *ret, int64_t H, int64_t N,
                            int64_t n0, int64_t n1, float gamma, float lam)
{
    enum { W = 64 };
    float nv[W], a[W];
    for (int64_t b = n0; b < n1; b += W) {
        const int64_t w = (n1 - b) < W ? (n1 - b) : W;
        for (int64_t j = 0; j < w; ++j) { nv[j] = next_value[b + j]; a[j] = 0.0f; }
        for (int64_t t = H - 1; t >= 0; --t) {
            const size_t base = (size_t)t * (size_t)N + (size_t)b;
            for (int64_t j = 0; j < w; ++j) {
                float r = rewards[base + j];
                const float v = values[base + j];
                uint8_t ud = undones[base + j];
                if (!unmasks[base + j]) { r = r + v; ud = 0; rewards[base + j] = r; undones[base + j] = 0; }
                const float m = ud ? gamma : 0.0f;
                const float mn = m * nv[j];
                const float q = r + mn;
                const float d = q - v;
                const float ml = m * lam;
                const float mla = ml * a[j];
                a[j] = d + mla;
                adv[base + j] = a[j];
                ret[base + j] = a[j] + v;
                nv[j] = v;
            }
        }
    }
    return 0;
}
