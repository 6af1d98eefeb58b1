"""Row f3: the files and the recorder the evaluator leaves behind, against what the reference's Evaluator
(elegantrl/train/evaluator.py:12-155) leaves for the same env, actor and call schedule (tests/golden/evaluator_format.npz,
oracle/make_golden.py:make_evaluator).  Pure torch on the CPU: the evaluator is off the hot path."""
import os

import numpy as np
import pytest
import torch as th

from tests.helpers import EVAL_SCHEDULE, ToyActor, ToySingleEnv, ToyVecEnv, load


@pytest.mark.parametrize("tag", ["single", "vec", "vec_overwrite"])
def test_evaluator_files_and_recorder_match_the_reference(tag, tmp_path):
    from elegantrl_amd.train import Config
    from elegantrl_amd.train.evaluator import Evaluator
    g = load("evaluator_format.npz")
    env = ToySingleEnv() if tag == "single" else ToyVecEnv(6)
    args = Config()
    args.gpu_id, args.eval_times, args.eval_per_step, args.eval_record_step = 0, 4 if tag == "single" else 12, 150, 0
    args.save_gap, args.if_keep_save, args.if_over_write = 2, True, tag == "vec_overwrite"
    actor = ToyActor.build(env.state_dim, env.action_dim)
    cwd = str(tmp_path)
    with th.no_grad():
        ev = Evaluator(cwd=cwd, env=env, args=args)
        for steps, exp_r, log in EVAL_SCHEDULE:
            ev.evaluate_and_save(actor, steps, exp_r, log)
        ev.save_or_load_recoder(if_save=True)
    assert sorted(os.listdir(cwd)) == [str(x) for x in g[f"{tag}_files"]]          # actor__{step:012}_{maxR:09.3f}.pt, actor.pt, ...
    rec = np.load(f"{cwd}/recorder.npy")
    assert rec.shape == g[f"{tag}_recorder"].shape                                  # (evaluations, 4 + logged values)
    np.testing.assert_allclose(rec, g[f"{tag}_recorder"], rtol=1e-6, atol=1e-6)      # step, avgR, stdR, expR, objC, objA, ...
    loaded = th.load(os.path.join(cwd, [f for f in os.listdir(cwd) if f.endswith(".pt")][0]), weights_only=False)
    assert th.equal(loaded(th.ones(1, env.state_dim)), actor(th.ones(1, env.state_dim)))   # whole pickled module, callable
    ev2 = Evaluator(cwd=cwd, env=env, args=args)
    ev2.save_or_load_recoder(if_save=False)
    assert ev2.total_step == rec[-1, 0] and len(ev2.recorder) == len(rec)
