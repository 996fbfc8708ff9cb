"""Host-side logic vs vectors captured from the imported reference (tests/golden/make_golden.py).
Bit-exact for indexing / integer / float32 sequencing semantics."""
import os

import numpy as np
import pytest

from idiaptts_amd.misc.utils import compute_deltas, interpolate_lin


@pytest.fixture(scope="module")
def host(golden_dir):
    return np.load(os.path.join(golden_dir, "host_logic.npz"))


def test_interpolate_lin_bit_exact(host):
    for i in range(int(host["il_count"])):
        ip, vuv = interpolate_lin(host["il_in_%d" % i])
        assert ip.dtype == host["il_ip_%d" % i].dtype
        assert np.array_equal(ip, host["il_ip_%d" % i]), i
        assert np.array_equal(vuv, host["il_vuv_%d" % i]), i


def test_compute_deltas_bit_exact(host):
    for i in range(int(host["cd_count"])):
        out = compute_deltas(host["cd_in_%d" % i])
        assert out.dtype == np.float32 and np.array_equal(out, host["cd_out_%d" % i])
    # SURVEY.md Appendix C known answer
    x = np.array([[1, 2], [2, 5], [4, 4], [8, 0]], dtype=np.float32)
    assert np.array_equal(compute_deltas(x), np.array([[1, 3], [1.5, 1], [3, -2.5], [4, -4]],
                                                       dtype=np.float32))
