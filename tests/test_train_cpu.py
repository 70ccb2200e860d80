"""CPU checks of row f2's host side: the numpy restatement of the dropout hash is pinned to the library's own
host function, and the site numbering is collision free."""
import numpy as np

from hawkeye_amd import lib as L
from oracle import dropout_masks as DM


def test_numpy_hash_equals_library_hash():
    lib = L.load()
    rng = np.random.default_rng(0)
    for seed, site in [(0, 0), (1, 2), (0xDEADBEEF, DM.EMBED_SITE), (12345, DM.site(11, DM.TFFN_PATH))]:
        idx = np.concatenate([np.arange(64, dtype=np.uint64), rng.integers(0, 2 ** 32, 64, dtype=np.uint64)])
        want = np.array([lib.hk_dropout_keep_u32(seed, site, int(i)) for i in idx], dtype=np.uint64)
        assert np.array_equal(DM.hash_u32(seed, site, idx), want)


def test_keep_rate_and_threshold():
    for p in (0.1, 0.2, 0.5):
        m = DM.keep_mask(7, 3, (400, 500), p)
        assert abs(m.mean() - (1 - p)) < 5e-3
    assert DM.keep_mask(7, 3, (4, 5), 0.0).all()
    assert DM.threshold(0.1) == int(float(np.float32(0.1)) * 2 ** 32)


def test_sites_are_distinct():
    sites = {DM.site(l, k) for l in range(12) for k in range(10)} | {DM.EMBED_SITE}
    assert len(sites) == 12 * 10 + 1
