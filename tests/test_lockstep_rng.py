# -*- coding: utf-8 -*-
"""lockstep.random_partition: the uint8 fast path draws the SAME values from the SAME numbers of numpy's global generator as
np.random.randint(0, k, n) (kmeans.py:171) -- the reference's goldens depend on that stream."""
import numpy as np


def test_random_partition_is_np_random_randint_draw_for_draw():
    from sr.recognition import lockstep
    assert lockstep._fast_partition_ok()          # (a numpy whose legacy generator changed would turn the fast path off)
    for k in (2, 4, 8, 16, 3, 5):
        np.random.seed(7 + k)
        want = [np.random.randint(0, k, n) for n in (1, 17, 1000, 33333)]
        after_want = np.random.random(3)
        np.random.seed(7 + k)
        got = [lockstep.random_partition(k, n) for n in (1, 17, 1000, 33333)]
        after_got = np.random.random(3)
        for a, b in zip(want, got):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(after_want, after_got)
        assert got[2].dtype == (np.uint8 if k & (k - 1) == 0 else want[2].dtype)
