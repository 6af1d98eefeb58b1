"""numpy restatement of the reference's PPO / replay arithmetic (oracle; test infrastructure only).

Every function cites the reference lines it restates (paths relative to
/root/reference).  All functions are dtype-generic: pass float32 arrays for the
"what the reference computes" answer (numpy rounds every op separately, exactly
like the reference's chain of ATen ops) or float64 arrays for a high-precision
answer to bound both implementations against.

Nothing here is used by the product path (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf as _erf

_SQRT2 = math.sqrt(2.0)
_LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))  # torch.distributions.Normal.log_prob constant


# --------------------------------------------------------------------------------------
# MLP pieces: elegantrl/agents/AgentBase.py:345-360 (build_mlp: Linear, GELU(exact erf), ..., Linear)
# --------------------------------------------------------------------------------------
def gelu(x: np.ndarray) -> np.ndarray:
    """nn.GELU() default (approximate='none'): x * Phi(x).  AgentBase.py:354."""
    dt = x.dtype
    return (x * (dt.type(0.5) * (dt.type(1.0) + _erf(x / dt.type(_SQRT2))))).astype(dt)


def gelu_grad(x: np.ndarray) -> np.ndarray:
    """d/dx [x Phi(x)] = Phi(x) + x phi(x)."""
    dt = x.dtype
    cdf = dt.type(0.5) * (dt.type(1.0) + _erf(x / dt.type(_SQRT2)))
    pdf = np.exp(dt.type(-0.5) * x * x) * dt.type(1.0 / math.sqrt(2.0 * math.pi))
    return (cdf + x * pdf).astype(dt)


@dataclass
class Mlp:
    """weights[i]: (out_i, in_i) like nn.Linear.weight; biases[i]: (out_i,)."""
    weights: List[np.ndarray]
    biases: List[np.ndarray]
    state_avg: np.ndarray
    state_std: np.ndarray
    action_std_log: Optional[np.ndarray] = None  # (A,) for the actor, None for the critic

    def astype(self, dt) -> "Mlp":
        return Mlp([w.astype(dt) for w in self.weights], [b.astype(dt) for b in self.biases],
                   self.state_avg.astype(dt), self.state_std.astype(dt),
                   None if self.action_std_log is None else self.action_std_log.astype(dt))

    def trainable(self) -> List[np.ndarray]:
        """flat order used by the HIP path: W1,b1,W2,b2,W3,b3[,action_std_log]."""
        out = []
        for w, b in zip(self.weights, self.biases):
            out += [w, b]
        if self.action_std_log is not None:
            out.append(self.action_std_log)
        return out


def state_norm(state: np.ndarray, net: Mlp) -> np.ndarray:
    """(s - avg) / (std + 1e-4).  AgentPPO.py:360-361 (actor), :440-441 (critic)."""
    dt = state.dtype
    return (state - net.state_avg) / (net.state_std + dt.type(1e-4))


def mlp_forward(x: np.ndarray, net: Mlp, keep: bool = False):
    """net(x) for the Sequential built by build_mlp.  Returns y (and the per-layer cache)."""
    acts = [x]
    pre = []
    h = x
    n = len(net.weights)
    for i, (w, b) in enumerate(zip(net.weights, net.biases)):
        z = h @ w.T + b
        pre.append(z)
        h = gelu(z) if i < n - 1 else z
        acts.append(h)
    return (h, (acts, pre)) if keep else h


def mlp_backward(dy: np.ndarray, net: Mlp, cache) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Gradients of sum(dy * y) w.r.t. weights and biases (what autograd's backward yields)."""
    acts, pre = cache
    n = len(net.weights)
    gw: List[np.ndarray] = [None] * n  # type: ignore
    gb: List[np.ndarray] = [None] * n  # type: ignore
    dz = dy
    for i in range(n - 1, -1, -1):
        gw[i] = dz.T @ acts[i]
        gb[i] = dz.sum(axis=0)
        if i > 0:
            dh = dz @ net.weights[i]
            dz = dh * gelu_grad(pre[i - 1])
    return gw, gb


# --------------------------------------------------------------------------------------
# Actor head: AgentPPO.py:368-386 (ActorPPO.get_action / get_logprob_entropy)
# --------------------------------------------------------------------------------------
def actor_mean(state: np.ndarray, actor: Mlp, keep: bool = False):
    return mlp_forward(state_norm(state, actor), actor, keep=keep)


def gaussian_logprob(action: np.ndarray, mean: np.ndarray, std_log: np.ndarray) -> np.ndarray:
    """Normal(mean, exp(std_log)).log_prob(action).sum(1); torch/distributions/normal.py log_prob:
    -((x - loc)**2) / (2*var) - log(scale) - log(sqrt(2 pi)).  AgentPPO.py:373-375, :383-384."""
    dt = action.dtype
    std = np.exp(std_log)
    var = std * std
    lp = -((action - mean) ** 2) / (dt.type(2.0) * var) - np.log(std) - dt.type(_LOG_SQRT_2PI)
    return lp.sum(axis=1)


def gaussian_entropy(std_log: np.ndarray, batch: int) -> np.ndarray:
    """Normal.entropy() = 0.5 + 0.5*log(2 pi) + log(scale), summed over actions.  AgentPPO.py:385."""
    dt = std_log.dtype
    per_dim = dt.type(0.5 + 0.5 * math.log(2.0 * math.pi)) + np.log(np.exp(std_log))
    return np.full((batch,), per_dim.sum(), dtype=dt)


def actor_sample(state: np.ndarray, actor: Mlp, eps: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """get_action with the N(0,1) draw made explicit: a = mean + std*eps (torch.normal(mean, std)),
    logprob = log_prob(a).sum(1).  The *pre-tanh* action is what gets stored (AgentPPO.py:115-119)."""
    mean = actor_mean(state, actor)
    std = np.exp(actor.action_std_log)
    action = mean + std * eps
    return action, gaussian_logprob(action, mean, actor.action_std_log)


def critic_value(state: np.ndarray, critic: Mlp, keep: bool = False):
    """CriticPPO.forward(state).squeeze(-1).  AgentPPO.py:435-438."""
    out = mlp_forward(state_norm(state, critic), critic, keep=keep)
    if keep:
        return out[0][..., 0], out[1]
    return out[..., 0]


# --------------------------------------------------------------------------------------
# GAE / lambda-return backward scan: AgentPPO.py:207-232 (get_advantages)
# --------------------------------------------------------------------------------------
def gae_scan(rewards: np.ndarray, undones: np.ndarray, unmasks: np.ndarray, values: np.ndarray,
             next_value: np.ndarray, gamma: float, lam: float, use_v_trace: bool = True,
             trunc_values: Optional[np.ndarray] = None):
    """Returns (advantages, rewards_after, undones_after).

    rewards/values: (H, N) float; undones/unmasks: (H, N) bool; next_value: (N,) = cri(last_state).
    The reference mutates ``rewards``/``undones`` in place (:211-214); here the mutated copies are
    returned.  ``trunc_values`` is V(s_t) used by the truncation fix-up; the reference re-runs the
    critic on the truncated rows, which equals ``values`` at those rows (default).
    """
    dt = rewards.dtype
    rewards = rewards.copy()
    undones = undones.copy()
    trunc = ~unmasks
    if trunc.any():                                    # :212
        tv = values if trunc_values is None else trunc_values
        rewards[trunc] = rewards[trunc] + tv[trunc]   # :213
        undones[trunc] = False                         # :214
    masks = undones.astype(dt) * dt.type(gamma)        # :216 (bool * python float -> float32)
    H = rewards.shape[0]
    adv = np.empty_like(values)
    nv = next_value.astype(dt).copy()                  # :219-220
    a = np.zeros_like(nv)                              # :222
    lam_t = dt.type(lam)
    if use_v_trace:                                    # :223-227
        for t in range(H - 1, -1, -1):
            nv = rewards[t] + masks[t] * nv
            a = nv - values[t] + masks[t] * lam_t * a  # ((nv - v) + ((m*lam)*a)) in this order
            adv[t] = a
            nv = values[t]
    else:                                              # :228-231 (bootstrap value is ignored)
        for t in range(H - 1, -1, -1):
            adv[t] = rewards[t] - values[t] + masks[t] * a
            a = values[t] + lam_t * adv[t]
    return adv, rewards, undones


def adv_normalize(adv: np.ndarray) -> np.ndarray:
    """(adv - adv.mean()) / (adv[::4, ::4].std() + 1e-5), unbiased std.  AgentPPO.py:149."""
    dt = adv.dtype
    sub = adv[::4, ::4]
    mean = adv.astype(np.float64).mean()
    std = sub.astype(np.float64).std(ddof=1)
    return ((adv - dt.type(mean)) / (dt.type(std) + dt.type(1e-5))).astype(dt)


def reward_sums(adv: np.ndarray, values: np.ndarray) -> np.ndarray:
    """AgentPPO.py:146."""
    return adv + values


# --------------------------------------------------------------------------------------
# Minibatch index decomposition: AgentPPO.py:178-180 and replay_buffer.py:123-125
# --------------------------------------------------------------------------------------
def split_ids(ids: np.ndarray, sample_len: int) -> Tuple[np.ndarray, np.ndarray]:
    """ids0 = ids % sample_len (time row), ids1 = ids // sample_len (env column) -- NOT row-major."""
    ids = ids.astype(np.int64)
    return ids % sample_len, ids // sample_len


# --------------------------------------------------------------------------------------
# PPO objectives: AgentPPO.py:173-205 (update_objectives), with manual backward
# --------------------------------------------------------------------------------------
def critic_objective(state, reward_sum, unmask, critic: Mlp):
    """obj_critic = (MSE_none(cri(state), reward_sum) * unmask).mean() and its parameter gradients."""
    dt = state.dtype
    v, cache = critic_value(state, critic, keep=True)
    um = unmask.astype(dt)
    diff = v - reward_sum
    obj = (diff * diff * um).mean()
    B = dt.type(state.shape[0])
    dv = (dt.type(2.0) * diff * um / B)[:, None]
    gw, gb = mlp_backward(dv, critic, cache)
    return obj, gw, gb


def actor_objective(state, action, logprob_old, advantage, unmask, actor: Mlp,
                    ratio_clip: float, lambda_entropy: float, objective: str = "reference"):
    """The actor objective of one minibatch and its gradients.  `objective`:
      "reference"  AgentPPO.py:193-204:  ratio = exp(new_logprob - old_logprob);
                   surrogate = adv * ratio * where(adv > 0, 1 - clip, 1 + clip);
                   minimise -( mean(surrogate*unmask) - lambda_entropy * mean(entropy*unmask) )
      "canonical"  the same loss around the textbook surrogate of helloworld/helloworld_PPO_single_file.py:337-339,
                   min(adv * ratio, adv * clamp(ratio, 1 - clip, 1 + clip))   (torch.min / clamp sub-gradients)
      "a2c"        AgentA2C.update_objectives (AgentPPO.py:296-303) for one env: new_logprob is per ACTION DIMENSION (the
                   reference's .sum(1) runs over the env axis of size 1), obj = mean over (batch, action dims) of
                   adv * logp_a, minimise -obj; no unmask, no entropy term (obj_entropy reported as 0)
    Returns (obj_actor, obj_entropy, grads_w, grads_b, grad_std_log) of the *minimised* loss.
    """
    dt = state.dtype
    B = state.shape[0]
    um = unmask.astype(dt)
    mean, cache = actor_mean(state, actor, keep=True)
    std_log = actor.action_std_log
    std = np.exp(std_log)
    var = std * std
    new_lp = gaussian_logprob(action, mean, std_log)
    ent = gaussian_entropy(std_log, B)
    diff = action - mean
    if objective == "a2c":
        A = action.shape[1]
        obj_s = (advantage * new_lp).mean() / dt.type(A)
        dlp = -advantage / dt.type(B * A)
        dmean = dlp[:, None] * (diff / var)
        dstd_log = (dlp[:, None] * (diff * diff / var - dt.type(1.0))).sum(axis=0)
        gw, gb = mlp_backward(dmean, actor, cache)
        return obj_s, dt.type(0.0), gw, gb, dstd_log.astype(dt)
    ratio = np.exp(new_lp - logprob_old)
    if objective == "canonical":
        lo, hi = dt.type(1.0 - ratio_clip), dt.type(1.0 + ratio_clip)
        s1 = advantage * ratio
        s2 = advantage * np.clip(ratio, lo, hi)
        surrogate = np.minimum(s1, s2)
        inside = (ratio >= lo) & (ratio <= hi)
        dsurr = np.where((s1 <= s2) | inside, s1, dt.type(0.0))     # d surrogate / d new_logprob
    else:
        w = np.where(advantage > 0, dt.type(1.0 - ratio_clip), dt.type(1.0 + ratio_clip)).astype(dt)
        surrogate = advantage * ratio * w
        dsurr = surrogate
    obj_s = (surrogate * um).mean()
    obj_e = (ent * um).mean()
    # loss = -(obj_s - lambda*obj_e)
    dlp = -(dsurr * um) / dt.type(B)                     # dloss/dnew_logprob
    dmean = dlp[:, None] * (diff / var)
    dstd_log = (dlp[:, None] * (diff * diff / var - dt.type(1.0))).sum(axis=0)
    dstd_log = dstd_log + dt.type(lambda_entropy) * um.mean()
    gw, gb = mlp_backward(dmean, actor, cache)
    return obj_s, obj_e, gw, gb, dstd_log.astype(dt)


# --------------------------------------------------------------------------------------
# Discrete policy: ActorDiscretePPO (AgentPPO.py:393-422) through torch.distributions.Categorical(probs)
# --------------------------------------------------------------------------------------
_CAT_EPS = float(np.finfo(np.float32).eps)      # Categorical clamps probs to [eps, 1 - eps] before the log (fp32 policy)


def softmax(z: np.ndarray) -> np.ndarray:
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def categorical_logits(p: np.ndarray) -> np.ndarray:
    """torch/distributions/utils.py probs_to_logits: log(clamp(p, eps, 1 - eps))."""
    dt = p.dtype
    return np.log(np.clip(p, dt.type(_CAT_EPS), dt.type(1.0 - _CAT_EPS)))


def categorical_sample(state: np.ndarray, actor: Mlp, u: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """get_action (AgentPPO.py:405-411) with the draw made explicit: inverse CDF of softmax(net(state)) at u in [0, 1)
    (first index whose running sum exceeds u, last index otherwise); log-prob of the drawn action."""
    p = softmax(actor_mean(state, actor))
    c = np.cumsum(p, axis=1)
    act = np.minimum((c <= u[:, None]).sum(axis=1), p.shape[1] - 1).astype(np.int32)
    return act, categorical_logits(p)[np.arange(p.shape[0]), act]


def actor_objective_discrete(state, action, logprob_old, advantage, unmask, actor: Mlp, ratio_clip: float, lambda_entropy: float):
    """AgentPPO.update_objectives (:193-204) with ActorDiscretePPO.get_logprob_entropy (:413-418):
    log_prob = logits[a], entropy = -sum p * logits (state dependent: its gradient reaches the network).
    Returns (obj_surrogate, obj_entropy, grads_w, grads_b) of the minimised loss."""
    dt = state.dtype
    B, um = state.shape[0], unmask.astype(dt)
    z, cache = actor_mean(state, actor, keep=True)
    p = softmax(z)
    L = categorical_logits(p)
    inside = ((p > dt.type(_CAT_EPS)) & (p < dt.type(1.0 - _CAT_EPS))).astype(dt)     # clamp passes no gradient outside
    rows = np.arange(B)
    a = action.astype(np.int64)
    new_lp = L[rows, a]
    ent = -(p * L).sum(axis=1)
    ratio = np.exp(new_lp - logprob_old)
    w = np.where(advantage > 0, dt.type(1.0 - ratio_clip), dt.type(1.0 + ratio_clip)).astype(dt)
    surrogate = advantage * ratio * w
    obj_s, obj_e = (surrogate * um).mean(), (ent * um).mean()
    dlp = -(surrogate * um) / dt.type(B) * inside[rows, a]
    dent = dt.type(lambda_entropy) * um / dt.type(B)
    onehot = np.zeros_like(p)
    onehot[rows, a] = 1
    h = -(L + inside)                                                                  # dH/dp_k
    dz = dlp[:, None] * (onehot - p) + dent[:, None] * p * (h - (p * h).sum(axis=1, keepdims=True))
    gw, gb = mlp_backward(dz.astype(dt), actor, cache)
    return obj_s, obj_e, gw, gb


def ppo_minibatch_step_discrete(buf, ids, actor: Mlp, critic: Mlp, st_a: "AdamState", st_c: "AdamState", *, lr: float,
                                max_norm: float, ratio_clip: float, lambda_entropy: float):
    """ppo_minibatch_step for the discrete agent: actions (H, N) int32."""
    states, actions, unmasks, logprobs, advantages, rsums = buf
    i0, i1 = split_ids(ids, states.shape[0])
    s, a = states[i0, i1], actions[i0, i1]
    um, lp, adv, rs = unmasks[i0, i1], logprobs[i0, i1], advantages[i0, i1], rsums[i0, i1]
    obj_c, gw, gb = critic_objective(s, rs, um, critic)
    optimizer_backward(critic.trainable(), [x for pair in zip(gw, gb) for x in pair], st_c, lr, max_norm)
    obj_s, obj_e, gw, gb = actor_objective_discrete(s, a, lp, adv, um, actor, ratio_clip, lambda_entropy)
    optimizer_backward(actor.trainable(), [x for pair in zip(gw, gb) for x in pair], st_a, lr, max_norm)
    return obj_c, obj_s, obj_e


# --------------------------------------------------------------------------------------
# optimizer_backward: AgentBase.py:239-248 (clip_grad_norm_(max_norm) then Adam.step)
# --------------------------------------------------------------------------------------
def clip_coef(grads: Sequence[np.ndarray], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g||_2 + 1e-6))."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return min(1.0, max_norm / (total + 1e-6))


@dataclass
class AdamState:
    m: List[np.ndarray] = field(default_factory=list)
    v: List[np.ndarray] = field(default_factory=list)
    step: int = 0


def adam_step(params: List[np.ndarray], grads: List[np.ndarray], st: AdamState, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (AgentPPO.py:24-25), single-tensor formulation, in place."""
    if not st.m:
        st.m = [np.zeros_like(p) for p in params]
        st.v = [np.zeros_like(p) for p in params]
    st.step += 1
    bc1 = 1.0 - beta1 ** st.step
    bc2 = 1.0 - beta2 ** st.step
    for p, g, m, v in zip(params, grads, st.m, st.v):
        dt = p.dtype
        m *= dt.type(beta1)
        m += dt.type(1.0 - beta1) * g
        v *= dt.type(beta2)
        v += dt.type(1.0 - beta2) * g * g
        denom = np.sqrt(v) / dt.type(math.sqrt(bc2)) + dt.type(eps)
        p -= dt.type(lr / bc1) * (m / denom)


def optimizer_backward(params, grads, st: AdamState, lr: float, max_norm: float) -> float:
    c = clip_coef(grads, max_norm)
    if c < 1.0:
        grads = [g * g.dtype.type(c) for g in grads]
    adam_step(params, grads, st, lr)
    return c


def ppo_minibatch_step(buf, ids, actor: Mlp, critic: Mlp, st_a: AdamState, st_c: AdamState, *,
                       lr: float, max_norm: float, ratio_clip: float, lambda_entropy: float, objective: str = "reference"):
    """One update_objectives call (AgentPPO.py:173-205; AgentA2C's :286-303 with objective="a2c", whose ids are time rows
    of a one-env buffer) on explicit ``ids``.
    buf = (states(H,N,S), actions(H,N,A), unmasks(H,N) bool, logprobs, advantages, reward_sums)."""
    states, actions, unmasks, logprobs, advantages, rsums = buf
    H = states.shape[0]
    i0, i1 = split_ids(ids, H)
    s, a = states[i0, i1], actions[i0, i1]
    um, lp, adv, rs = unmasks[i0, i1], logprobs[i0, i1], advantages[i0, i1], rsums[i0, i1]
    obj_c, gw, gb = critic_objective(s, rs, um, critic)
    g = [x for pair in zip(gw, gb) for x in pair]
    optimizer_backward(critic.trainable(), g, st_c, lr, max_norm)
    obj_s, obj_e, gw, gb, gsl = actor_objective(s, a, lp, adv, um, actor, ratio_clip, lambda_entropy, objective)
    g = [x for pair in zip(gw, gb) for x in pair] + [gsl]
    optimizer_backward(actor.trainable(), g, st_a, lr, max_norm)
    return obj_c, obj_s, obj_e


# --------------------------------------------------------------------------------------
# n-step discounted return of the off-policy agents: elegantrl/agents/AgentBase.py:226-237
# --------------------------------------------------------------------------------------
def cum_rewards(rewards: np.ndarray, undones: np.ndarray, next_value: np.ndarray, gamma: float) -> np.ndarray:
    """masks = undones * gamma (:229); for t = H-1 .. 0: cum[t] = next_value = rewards[t] + masks[t] * next_value (:235-236).
    dtype-generic; with float32 inputs every product / sum rounds separately like the reference's ATen ops."""
    dt = rewards.dtype
    masks = undones.astype(dt) * dt.type(gamma)
    out = np.empty_like(rewards)
    nv = next_value.astype(dt).reshape(-1)
    for t in range(rewards.shape[0] - 1, -1, -1):
        out[t] = nv = rewards[t] + masks[t] * nv
    return out


def cum_rewards_slice(p: int, add_size: int, max_size: int) -> Tuple[int, int]:
    """rows ReplayBuffer.update_cum_rewards hands to get_cumulative_rewards (elegantrl/train/replay_buffer.py:213-223)."""
    p1 = p if p >= add_size else max_size
    return p1 - add_size, p1


# --------------------------------------------------------------------------------------
# Off-policy ring buffer: elegantrl/train/replay_buffer.py:11-134
# --------------------------------------------------------------------------------------
class Ring:
    """ReplayBuffer.__init__/update/sample restated on numpy (non-PER path)."""

    def __init__(self, max_size: int, state_dim: int, action_dim: int, num_seqs: int = 1, if_discrete: bool = False):
        self.p = 0
        self.if_full = False
        self.cur_size = 0
        self.add_size = 0
        self.max_size = max_size
        self.num_seqs = num_seqs
        f = np.float32
        self.states = np.zeros((max_size, num_seqs, state_dim), f)
        # discrete agents: one uint8 per transition (replay_buffer.py:53-54); int32 actions are narrowed on assignment
        self.actions = np.zeros((max_size, num_seqs), np.uint8) if if_discrete else np.zeros((max_size, num_seqs, action_dim), f)
        self.rewards = np.zeros((max_size, num_seqs), f)
        self.undones = np.zeros((max_size, num_seqs), f)   # floats, unlike the on-policy bools (:57-58)
        self.unmasks = np.zeros((max_size, num_seqs), f)

    def update(self, items) -> None:                       # replay_buffer.py:78-118
        states, actions, rewards, undones, unmasks = items
        self.add_size = rewards.shape[0]
        p = self.p + self.add_size
        dst = (self.states, self.actions, self.rewards, self.undones, self.unmasks)
        if p > self.max_size:                              # :87 wrap: tail then head
            self.if_full = True
            p0, p1, p2 = self.p, self.max_size, self.max_size - self.p
            p = p - self.max_size
            for d, s in zip(dst, items):
                d[p0:p1] = s[:p2]
                d[0:p] = s[-p:]
        else:                                              # :100 contiguous (p == max_size lands here)
            for d, s in zip(dst, items):
                d[self.p:p] = s
        self.p = p
        self.cur_size = self.max_size if self.if_full else self.p

    def sample(self, ids: np.ndarray):                     # replay_buffer.py:120-134 with ids explicit
        sample_len = self.cur_size - 1
        i0, i1 = split_ids(ids, sample_len)
        return (self.states[i0, i1], self.actions[i0, i1], self.rewards[i0, i1], self.undones[i0, i1],
                self.unmasks[i0, i1], self.states[i0 + 1, i1]), (i0, i1)
