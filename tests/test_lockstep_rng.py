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


def test_partition_stream_draws_ahead_what_the_sequential_refit_draws():
    """PartitionStream: the partitions of a whole refit drawn ahead on a worker thread from a copy of the global
    generator -- the values np.random.randint(0, 2^(i+1), n_s) gives state after state, split after split, and the global
    generator left exactly where those draws leave it (whatever upper bound the worker was given)."""
    from sr.recognition import lockstep
    for lengths, n_splits, slack in (([5, 0, 70000, 131072 - 70005, 3], 2, 12345), ([1000, 2000], 3, 0), ([], 2, 777),
                                      ([65536], 1, 0), ([65536, 65536], 2, 65536)):
        np.random.seed(11)
        want = [[np.random.randint(0, 2 ** (i + 1), n) for i in range(n_splits)] for n in lengths]
        after_want = np.random.random(3)
        np.random.seed(11)
        st = lockstep.PartitionStream(n_splits * sum(lengths) + slack)
        got = st.take(lengths, n_splits)
        after_got = np.random.random(3)
        assert len(got) == len(want)
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
                assert y.dtype == np.uint8
        np.testing.assert_array_equal(after_want, after_got)
    np.random.seed(5)
    want = np.random.random(2)
    np.random.seed(5)
    lockstep.PartitionStream(1000).cancel()
    np.testing.assert_array_equal(np.random.random(2), want)
