"""Known-answer / property tests of the numpy restatement of SB3's VecNormalize (oracle/vecnorm.py).  SB3 itself is absent
here, so the restatement is pinned by the mathematics it implements: the parallel-variance merge must reproduce the moments of
the concatenated data (with RunningMeanStd's prior pseudo-batch: mean 0, var 1, count 1e-4), and the wrapper's bookkeeping
(discounted returns, reset at dones, clipping, frozen statistics in evaluation) must hold."""
import numpy as np

from oracle.vecnorm import RunningMeanStd, VecNormalizeRef


def test_running_mean_std_equals_moments_of_concatenation():
    rng = np.random.default_rng(0)
    a, b, c = rng.normal(1.0, 2.0, size=(100, 5)), rng.normal(-3.0, 0.5, size=(37, 5)), rng.normal(0.0, 4.0, size=(1, 5))
    r = RunningMeanStd(shape=(5,))
    for x in (a, b, c):
        r.update(x)
    eps = 1e-4   # the prior is a pseudo-batch of eps samples with mean 0 and variance 1
    n = eps + 138
    x = np.concatenate([a, b, c])
    mean = x.sum(axis=0) / n
    m2 = ((x - mean) ** 2).sum(axis=0) + eps * (1.0 + mean ** 2)
    np.testing.assert_allclose(r.mean, mean, rtol=1e-12)
    np.testing.assert_allclose(r.var, m2 / n, rtol=1e-12)
    assert abs(r.count - n) < 1e-12


def test_vecnormalize_bookkeeping():
    rng = np.random.default_rng(1)
    v = VecNormalizeRef(4, 3, gamma=0.9, clip_obs=1.5, clip_reward=0.7)
    o0 = rng.normal(size=(4, 3)).astype(np.float32)
    assert np.abs(v.reset(o0)).max() <= 1.5 and v.obs_rms.count == 1e-4 + 4
    ret = np.zeros(4)
    for t in range(5):
        rew = rng.normal(size=4).astype(np.float32)
        done = np.array([t == 2, False, False, t == 3])
        ret = ret * 0.9 + rew
        expected_var = None
        obs, r, term = v.step(rng.normal(size=(4, 3)).astype(np.float32), rew, done, rng.normal(size=(4, 3)).astype(np.float32))
        assert obs.dtype == np.float32 and np.abs(obs).max() <= 1.5 and np.abs(r).max() <= 0.7 and np.abs(term).max() <= 1.5
        np.testing.assert_allclose(r, np.clip(rew / np.sqrt(v.ret_rms.var + 1e-8), -0.7, 0.7))
        ret[done] = 0
        np.testing.assert_allclose(v.returns, ret)
    # evaluation: statistics frozen, rewards untouched (load_model.py:114-116)
    e = VecNormalizeRef(4, 3, training=False, norm_reward=False)
    e.obs_rms.mean[:] = 1.0; e.obs_rms.var[:] = 4.0
    obs, r, _ = e.step(np.full((4, 3), 3.0, np.float32), np.ones(4, np.float32), np.zeros(4, bool))
    np.testing.assert_allclose(obs, 1.0, atol=1e-6)
    assert np.all(r == 1.0) and e.obs_rms.count == 1e-4 and np.all(e.returns == 0)
