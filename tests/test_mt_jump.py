"""Host side of the sampler: GF(2) jump-ahead of MT19937 (raytracer_amd/csrc/rt_mt19937.cpp) against
sequential generation by the oracle's generator (which is pinned to the reference's random_double()).
CPU only."""
import numpy as np
import pytest

import orc
import raytracer_amd as ra

UPPER = np.uint32(0x80000000)


def oracle_words_from_state(state, n):
    """Continue an MT19937 from a raw 624-word window (reference algorithm, numpy, vectorised blocks)."""
    x = state.copy()
    out = np.empty(n, np.uint32)
    got = 0
    while got < n:
        for k in range(624):
            y = (x[k] & UPPER) | (x[(k + 1) % 624] & np.uint32(0x7FFFFFFF))
            x[k] = x[(k + 397) % 624] ^ (y >> np.uint32(1)) ^ (np.uint32(0x9908B0DF) if (y & np.uint32(1)) else np.uint32(0))
        y = x.copy()
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
        y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
        y ^= y >> np.uint32(18)
        m = min(624, n - got)
        out[got:got + m] = y[:m]
        got += m
    return out


@pytest.mark.parametrize("pos", [0, 1, 2, 623, 624, 625, 1000, 19937, 360000, 2**20 + 12345])
def test_state_at_matches_sequential(pos):
    st = ra.mt19937_state_at(pos)
    r = orc.Rng(5489)
    r.discard(pos)
    want = r.u32(700)
    got = oracle_words_from_state(st, 700)
    assert np.array_equal(got, want)


def test_far_jump_is_consistent():
    # 2^40 + 7 reached in one jump == reached in two jumps through the stream-state helper
    a = ra.mt19937_state_at(2**40 + 7)
    b = ra.mt19937_stream_states(7, 2**39, 3)[2]
    assert np.array_equal(oracle_words_from_state(a, 64), oracle_words_from_state(b, 64))


def test_stream_states_tree_doubling():
    base, stride, n = 4000, 6 * 400 * 4, 7  # 7 streams: not a power of two
    states = ra.mt19937_stream_states(base, stride, n)
    for r in range(n):
        g = orc.Rng(5489)
        g.discard(base + r * stride)
        assert np.array_equal(oracle_words_from_state(states[r], 64), g.u32(64)), r


def test_other_seed():
    st = ra.mt19937_state_at(5000, seed=12345)
    r = orc.Rng(12345)
    r.discard(5000)
    assert np.array_equal(oracle_words_from_state(st, 100), r.u32(100))


def test_stream_states_stay_correct_once_the_polynomial_cache_is_full():
    """The host cache of jump polynomials stops growing at 4096 entries.  mt_stream_states holds one polynomial
    per doubling level at the same time; they must stay distinct when none of them can be cached any more
    (a long-lived process seeking to many distinct positions fills the cache)."""
    for p in range(1, 4300):  # 4299 distinct distances: more than the cache holds
        ra.mt19937_state_at(p)
    base, stride, n = 1234, 5 * 624, 9  # 4 doubling levels, none of their strides cached
    states = ra.mt19937_stream_states(base, stride, n)
    for r in range(n):
        g = orc.Rng(5489)
        g.discard(base + r * stride)
        assert np.array_equal(oracle_words_from_state(states[r], 64), g.u32(64)), r
    # and a single state after the fill
    st = ra.mt19937_state_at(987654)
    g = orc.Rng(5489)
    g.discard(987654)
    assert np.array_equal(oracle_words_from_state(st, 64), g.u32(64))


def test_compiled_in_characteristic_polynomial_equals_a_fresh_berlekamp_massey_run():
    """phi (degree 19937, 135 terms) is a table in the library (rt_mt19937_phi.inc); recomputing it from the
    generator's own output must give the same 312 words."""
    from raytracer_amd import _lib as L
    table, fresh = np.zeros(312, np.uint64), np.zeros(312, np.uint64)
    L.check(L.lib.rt_mt_phi(table.ctypes.data, 0), "rt_mt_phi")
    L.check(L.lib.rt_mt_phi(fresh.ctypes.data, 1), "rt_mt_phi")
    assert np.array_equal(table, fresh)
    assert (int(table[311]) >> (19937 % 64)) & 1 and int(table[311]) >> (19937 % 64 + 1) == 0  # monic, degree 19937
    assert sum(bin(int(x)).count("1") for x in table) == 135
