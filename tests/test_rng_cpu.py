"""CPU: the NumPy restatement of the device generator (oracle/rng.py) against the known-answer vectors published with
Philox4x32-10 (Random123 kat_vectors, Salmon et al. SC'11), and the statistical sanity of the derived draws."""

import numpy as np

from oracle import rng


def _kat(counter, key):
    return [int(x) for x in rng.philox4x32_10(*[np.uint32(c) for c in counter], key[0], key[1])]


def test_philox_known_answer_vectors():
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _kat((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _kat((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_draws_are_a_function_of_seed_chain_transition_only():
    a = rng.normal(11, 0, 8, 5, 3)
    b = rng.normal(11, 4, 4, 5, 3)
    assert np.array_equal(a[4:], b)  # chain_offset + i is the chain's identity: sharding does not matter
    assert not np.array_equal(rng.normal(12, 0, 8, 5, 3), a)
    assert not np.array_equal(rng.normal(11, 0, 8, 5, 4), a)
    assert np.array_equal(rng.uniform(11, 4, 4, 9), rng.uniform(11, 0, 8, 9)[4:])
    s = rng.steps(5, 0, 4000, 1, 2, 7)
    assert s.min() == 2 and s.max() == 6 and abs(s.mean() - 4.0) < 0.1


def test_moments():
    z = rng.normal(7, 0, 4000, 33, 1)  # odd dimension: the last pair is half used
    assert z.shape == (4000, 33)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.1
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.05
    u = rng.uniform(7, 0, 100000, 2)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.005 and abs(u.var() - 1 / 12) < 0.002
