"""Producer side of the path on the CPU: the stratified sampler of dsac_amd/e2e.py against the REAL
stochasticSubSample (bit-exact: same mt19937 stream), and the patch layout against the reference's getCoordImg."""
import numpy as np
import pytest

from oracle import reference as ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libdsac_ref.so not built and /root/reference absent")


def numpy_patches(img, xy, patch=42):
    half = patch // 2
    return np.stack([img[y - half:y + half, x - half:x + half].transpose(2, 0, 1).astype(np.float32) for x, y in xy])


@pytest.mark.parametrize("seed", [1305, 1, 20260925])
def test_sampler_and_patch_layout_match_the_reference(seed):
    from dsac_amd.e2e import stochastic_sub_sample
    ref.lib()
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    xy_ref, patches_ref = ref.subsample_and_patches(seed, img)
    xy = stochastic_sub_sample(640, 480, seed=seed)
    assert np.array_equal(xy, xy_ref)
    # one pixel per cell of the 40 x 40 partition of the interior, never closer than half a patch to the border
    assert xy[:, 0].min() >= 21 and xy[:, 0].max() <= 619 and xy[:, 1].min() >= 21 and xy[:, 1].max() <= 459
    g = xy.reshape(40, 40, 2)
    assert (np.diff(g[:, :, 0], axis=1) >= 0).all() and (np.diff(g[:, :, 1], axis=0) >= 0).all()  # cell borders are not integers
    assert patches_ref.shape == (1600, 3, 42, 42)  # no border patch was skipped
    assert np.array_equal(numpy_patches(img, xy), patches_ref)
