"""CPU: host-side frame preprocessing against the oracle's restatement on hand-made
Atari-shaped frames, and both against the REFERENCE's own outputs (tests/golden/g9_preprocessing.npz, recorded by
make_golden.py g9 from /root/reference/a2c/preprocessing.py with skimage.color stubbed)."""
import numpy as np

from a2c_amd import preprocessing as P
from oracle import a2c_oracle as O


def _frame(seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, size=(210, 160, 3), dtype=np.uint8)
    f[40:60, 20:40, 0] = 144
    f[100:120, 60:90, 0] = 109
    return f


def test_pong_prep_shape_values_and_oracle():
    for s in range(3):
        f = _frame(s)
        out = P.pong_prep(f.copy())
        assert out.shape == (1, 80, 80) and out.dtype == np.uint8
        assert set(np.unique(out)) <= {0, 1}
        assert np.array_equal(out, O.pong_prep(f.copy()))
        assert out[0, (40 - 35) // 2 + 1, 20 // 2 + 1] == 0          # background colour erased


def test_null_and_breakout_prep():
    f = _frame(5)
    assert P.null_prep(f).shape == (1, 210, 160, 3)
    b = P.breakout_prep(f)
    # skimage <= 0.18 rgb2grey returns a 2-D input unchanged (see a2c_amd/preprocessing.py): uint8 0..255, not floats
    assert b.shape == (1, 80, 72) and b.dtype == np.uint8 and b.max() > 1
    assert np.array_equal(b[0], f[35:195, 8:-8][::2, ::2, 0])
    n = P.normalize_prep(np.array([[0.0, 255.0]]))
    assert np.allclose(n, [[[-3.0, 3.0]]])


def test_preprocessors_match_the_reference_recorded_outputs(golden):
    from cases import PREP_SEEDS, atari_frame
    g = golden["g9_preprocessing"]
    for s in PREP_SEEDS:
        for mod in (P, O):
            out = mod.pong_prep(atari_frame(s))
            assert out.dtype == g[f"pong{s}"].dtype and np.array_equal(out, g[f"pong{s}"]), (mod.__name__, s)
            out = mod.breakout_prep(atari_frame(s))
            assert out.dtype == g[f"breakout{s}"].dtype and np.array_equal(out, g[f"breakout{s}"]), (mod.__name__, s)
            assert tuple(mod.null_prep(atari_frame(s)).shape) == tuple(g[f"null{s}_shape"])
        assert int(g[f"pong{s}"].sum()) > 0 and int((g[f"pong{s}"] == 0).sum()) > 1000       # both branches present
