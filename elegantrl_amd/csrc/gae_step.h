// One step of the exact GAE recurrence (elegantrl/agents/AgentPPO.py:207-232), shared by the stand-alone scan (gae.hip, EXACT) and
// the persistent rollout's epilogue (rollout_fused.hip): the reference's op order with every product and sum rounded on its own
// (no fma contraction), so both are bit-identical to oracle/gae_scan.c.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// state carried from step t + 1 to step t: `nv` (the value of the next state; cri(last_state) at t = H - 1) and `a` (the carried
// advantage, 0 at t = H - 1).  (r, v, undone, unmask) are the step's reward, value and flags; a truncated step (unmask = 0)
// bootstraps with its OWN value and cuts the chain (:211-214) -- the caller decides whether that is written back.
// Returns the advantage of step t; `r_eff` / `ud_eff` are what get_advantages leaves in the caller's rewards / undones.
template <bool VTRACE>
__device__ __forceinline__ float erl_gae_step(float r, float v, uint8_t ud, uint8_t um, float gamma, float lam, float &nv, float &a, float &r_eff,
                                              uint8_t &ud_eff)
{
#pragma clang fp contract(off)
    if (!um) {
        r = r + v;
        ud = 0;
    }
    r_eff = r;
    ud_eff = ud;
    const float m = ud ? gamma : 0.f;
    float out;
    if (VTRACE) {      // :225-227
        nv = r + m * nv;
        a = (nv - v) + (m * lam) * a;
        out = a;
        nv = v;
    } else {           // :229-231
        out = (r - v) + m * a;
        a = v + lam * out;
    }
    return out;
}
