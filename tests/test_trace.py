"""Synthetic trace generator: determinism, independence of the batch size, format round trip."""
import numpy as np

from awesomeslam_amd import trace as tg


def test_deterministic_and_batch_independent():
    a = tg.make_traces(8, 200, B=3, seed=5)
    b = tg.make_traces(8, 200, B=1, seed=5, first_traj=2, max_obs=a.max_obs)
    for f in ("odom", "dt", "obs_new", "n_obs", "obs", "landmarks", "truth"):
        assert np.array_equal(getattr(a, f)[2], getattr(b, f)[0]), f
    c = tg.make_traces(8, 200, B=3, seed=5)
    assert np.array_equal(a.obs, c.obs) and np.array_equal(a.odom, c.odom)
    d = tg.make_traces(8, 200, B=1, seed=6)
    assert not np.array_equal(a.odom[0], d.odom[0])


def test_shapes_types_and_scenario():
    L, T = 13, 300
    tr = tg.make_traces(L, T, B=2, seed=1)
    assert tr.odom.shape == (2, T, 8) and tr.odom.dtype == np.float64
    assert tr.dt.dtype == np.float32 and tr.obs.dtype == np.float32 and tr.n_obs.dtype == np.int32
    assert tr.obs.shape[2] % 4 == 0 and tr.obs.shape[2] >= L
    assert tr.warmup == 3 * tg.STOP_STEPS
    assert tr.n_obs[:, tr.warmup:].min() == L          # every landmark stays in view (module docstring)
    d = np.hypot(*(tr.landmarks[0][:, None] - tr.landmarks[0][None]).transpose(2, 0, 1))
    d[np.eye(L, dtype=bool)] = 9
    assert d.min() > 2 * 0.5                            # > 2 x MIN_DIST_THRESH: association is unambiguous
    tw = tr.odom[0, tr.warmup:, 6:8]
    assert (tw[:, 1] == 0).any() and ((tw[:, 0] == 0) & (tw[:, 1] == 0)).any()      # exact straights and stops
    assert ((np.abs(tw[:, 1]) <= 1e-3) & (tw[:, 1] != 0)).any() or T < 2000          # |wz| <= 0.001 branch
    assert tg.full_dim(L) == 29 and tg.dim_cap(L) == 30


def test_rewalk_messages_are_repeated():
    tr = tg.make_traces(5, 60, B=1, seed=2, sensor_every=3)
    assert tr.obs_new[0, :7].tolist() == [1, 0, 0, 1, 0, 0, 1]
    assert np.array_equal(tr.obs[0, 1], tr.obs[0, 0]) and np.array_equal(tr.obs[0, 2], tr.obs[0, 0])


def test_save_load_roundtrip(tmp_path):
    tr = tg.make_traces(5, 50, B=2, seed=3)
    p = str(tmp_path / "t.npz")
    tr.save(p)
    back = tg.Trace.load(p)
    for f in ("odom", "dt", "obs_new", "n_obs", "obs", "landmarks", "truth"):
        assert np.array_equal(getattr(tr, f), getattr(back, f))
    assert back.warmup == tr.warmup
