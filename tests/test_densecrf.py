"""The DenseCRF oracle (oracle/densecrf.py — parity with pydensecrf itself is unpinned, the library is installed nowhere this
runs): properties that hold for the published algorithm and that a wrong restatement would break."""
import numpy as np
import pytest

from oracle import densecrf as D


def test_barycentric_weights_form_a_partition_of_unity():
    rng = np.random.default_rng(0)
    for d in (2, 5):
        lat = D.Permutohedral((rng.random((d, 500)) * 40).astype(np.float32))
        b = lat.barycentric
        assert b.shape == (500, d + 1) and np.abs(b.sum(1) - 1).max() < 1e-5 and b.min() > -1e-5
        assert lat.offset.min() >= 0 and lat.offset.max() < lat.M
        # every point's d+1 vertices are distinct lattice points
        assert all(len(set(row)) == d + 1 for row in lat.offset)


def test_lattice_filter_tracks_a_gaussian_filter():
    """On smooth positions (a 2-D pixel grid scaled by sigma = 3) the splat-blur-slice filter is a Gaussian filter up to its
    known constant gain; on a 5-D bilateral feature set it must at least be strongly correlated with it."""
    rng = np.random.default_rng(1)
    feat = D.features_gaussian(24, 32, 3.0)
    x = rng.random((24 * 32, 2)).astype(np.float32)
    ratio = D.Permutohedral(feat).compute(x) / D.gaussian_filter_bruteforce(feat, x)
    assert 0.8 < ratio.mean() < 0.95 and ratio.std() < 0.04
    img = rng.integers(0, 256, (16, 20, 3)).astype(np.uint8)
    img[:, :10] = img[:, :10] // 8 + 100
    f5 = D.features_bilateral(img, 5, 20)
    got, ref = D.Permutohedral(f5).compute(x[:320]), D.gaussian_filter_bruteforce(f5, x[:320])
    assert np.corrcoef(got.reshape(-1), ref.reshape(-1))[0, 1] > 0.9


def test_filter_is_linear_and_nonnegative():
    rng = np.random.default_rng(2)
    lat = D.Permutohedral(D.features_gaussian(10, 12, 1.0))
    a, b = rng.random((120, 2)).astype(np.float32), rng.random((120, 2)).astype(np.float32)
    assert np.abs(lat.compute(a + b) - (lat.compute(a) + lat.compute(b))).max() < 1e-4
    assert lat.compute(a).min() >= 0


def test_unary_and_features_follow_the_call_site():
    u = D.unary_from_mask(np.array([[0, 255, 128]], np.uint8))
    assert u.shape == (2, 3) and u.dtype == np.float32
    assert np.allclose(u[:, 0], [0.0, -np.log(np.float32(1e-5))]) and np.allclose(u[:, 1], [-np.log(np.float32(1e-5)), 0.0])
    assert np.allclose(u[:, 2], -np.log(np.array([127, 128], np.float32) / 255))
    f = D.features_bilateral(np.arange(24, dtype=np.uint8).reshape(2, 4, 3), 23, 7)
    assert f.shape == (5, 8) and np.allclose(f[:, 5], [1 / 23, 1 / 23, 15 / 7, 16 / 7, 17 / 7])


def test_refine_mask_recovers_dark_glyphs_from_a_loose_noisy_mask():
    """The behaviour the reference relies on: a padded, noisy text mask over high-contrast glyphs snaps to the glyphs."""
    rng = np.random.default_rng(0)
    H, W = 60, 200
    img = np.full((H, W, 3), 230, np.uint8) + rng.integers(0, 10, (H, W, 3)).astype(np.uint8)
    mask = np.zeros((H, W), np.uint8)
    for k in range(8):
        x0 = 10 + k * 22
        img[15:45, x0:x0 + 12] = 30
        mask[13:47, x0 - 2:x0 + 14] = 255
    mask[rng.random((H, W)) < 0.03] ^= 255
    out, q = D.refine_mask(img, mask, return_q=True)
    assert out.dtype == np.uint8 and set(np.unique(out)) <= {0, 255}
    assert np.array_equal(out > 0, img[..., 0] < 100)
    assert np.abs(q.sum(0) - 1).max() < 1e-5
    # an all-background mask stays empty, a 1-pixel-tall crop works
    assert not D.refine_mask(img, np.zeros((H, W), np.uint8)).any()
    assert D.refine_mask(img[:1], mask[:1]).shape == (1, W)
