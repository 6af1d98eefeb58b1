// The actor objective of one sample, shared by every PPO minibatch kernel (ppo_step.hip, ppo_step_w4.hip, mlpn_common.h).
// `objective` selects the form (include/erl_hip.h, ERL_PPO_OBJ_*):
//   REFERENCE  elegantrl/agents/AgentPPO.py:199     surr = adv ratio (adv > 0 ? 1 - clip : 1 + clip)   (a sign-dependent
//              scale, not a clamp);   loss = -(mean(surr unmask) - lambda_entropy mean(entropy unmask))
//   CANONICAL  helloworld/helloworld_PPO_single_file.py:337-339   surr = min(adv ratio, adv clamp(ratio, 1 - clip, 1 + clip)),
//              same loss around it
//   A2C        elegantrl/agents/AgentPPO.py:296-303 (AgentA2C.update_objectives)   obj = mean over (batch, action dims) of
//              adv logp_a = mean_B(adv logp / A): no ratio, no clip, no unmask, no entropy term;   loss = -obj
#pragma once
#include "../../include/erl_hip.h"

struct PpoActorTerms {
    float logged;     // this sample's contribution to the logged actor objective (before the 1/B)
    float dlp;        // d loss / d logp_new of this sample (before the 1/B)
    float ent_w;      // weight of the entropy term's gradient d loss / d entropy-sum (before the 1/B): lambda_entropy unmask
    float ent_mask;   // this sample's weight in the logged entropy mean
};

__device__ __forceinline__ PpoActorTerms ppo_actor_terms(int objective, float adv, float logp_new, float logp_old, float clip,
                                                          float lambda_entropy, float um, int A, bool fast_exp)
{
    PpoActorTerms o;
    if (objective == ERL_PPO_OBJ_A2C) {
        const float w = adv / (float)A;
        o.logged = w * logp_new;
        o.dlp = -w;
        o.ent_w = 0.f;
        o.ent_mask = 0.f;
        return o;
    }
    const float ratio = fast_exp ? __expf(logp_new - logp_old) : expf(logp_new - logp_old);
    float surr, dsurr;
    if (objective == ERL_PPO_OBJ_CANONICAL) {
        const float s1 = adv * ratio;
        const float s2 = adv * fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        const bool first = s1 <= s2;                  // torch.min: a tie splits the gradient over two equal branches
        surr = first ? s1 : s2;
        const bool inside = ratio >= 1.f - clip && ratio <= 1.f + clip;   // clamp passes the gradient inside the interval
        dsurr = (first || inside) ? s1 : 0.f;
    } else {
        surr = adv * ratio * (adv > 0.f ? 1.f - clip : 1.f + clip);
        dsurr = surr;
    }
    o.logged = surr * um;
    o.dlp = -(dsurr * um);
    o.ent_w = lambda_entropy * um;
    o.ent_mask = um;
    return o;
}
