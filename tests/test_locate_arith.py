"""The probe's two exact-arithmetic shortcuts against plain integer arithmetic
(MIBloomFilter.hpp:468 `hashes[i] % m_bv.size()`):
  x % m     reciprocal multiply + one conditional subtract (grp_mod_m)
  pos / W   magic-number multiply, W in [13, 64]           (grp_div_w)
Adversarial operands: m near 2^32, 2^33, the BASELINE sizes, the implementation limit
2^50; x = q*m, q*m - 1, q*m + 1, 2^64 - 1, values whose estimated quotient is one short.
The host instantiation runs on the CPU (no GPU needed); the device instantiation
(__umul64hi) is checked by the same vectors under `-m gpu`."""
import numpy as np
import pytest

M_VALUES = [64, 65, 127, 1000003, (1 << 32) - 64, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 32) + 64, (1 << 33) + 64,
            2847366528,        # C1 (SURVEY 8 size table)
            61146729472,       # C2 / C3
            101911215744,      # C4
            (1 << 40) - 87, (1 << 49) + 12345, (1 << 50) - 64, (1 << 50) - 1]
W_VALUES = [13, 14, 21, 31, 32, 33, 47, 59, 60, 61, 63, 64]
U64 = (1 << 64) - 1


def vectors(m, rng):
    xs = [0, 1, m - 1, m, m + 1, 2 * m - 1, 2 * m, U64, U64 - 1, U64 - m, U64 - m + 1, 1 << 63, (1 << 63) - 1, (1 << 63) + 1]
    qmax = U64 // m
    qs = [1, 2, 3, qmax, qmax - 1, qmax // 2, qmax // 3, (1 << 31) % (qmax + 1), (1 << 32) % (qmax + 1)] + [int(q) for q in rng.integers(0, qmax + 1, size=400, dtype=np.uint64)]
    for q in qs:
        for d in (-1, 0, 1, m - 1, m // 2):
            x = q * m + d
            if 0 <= x <= U64:
                xs.append(x)
    xs += [int(v) for v in rng.integers(0, 1 << 64, size=4000, dtype=np.uint64)]
    return np.array(xs, dtype=np.uint64)


def check(native, on_device, engine=None):
    rng = np.random.default_rng(2024)
    for m in M_VALUES:
        x = vectors(m, rng)
        exp_mod = np.array([int(v) % m for v in x], dtype=np.uint64)
        for W in W_VALUES:
            mod, div = native.debug_locate(x, m, W, on_device, engine)
            assert np.array_equal(mod, exp_mod), (m, W, "mod")
            assert np.array_equal(div, exp_mod // np.uint64(W)), (m, W, "div")
    # positions up to the implementation limit of the bucket division: every multiple of W
    # around 2^50 and the last position of the bucket before it
    for W in W_VALUES:
        m = (1 << 50) - 1
        base = ((1 << 50) // W - 300) * W
        pos = np.array([base + i * W + d for i in range(290) for d in (0, 1, W - 1)], dtype=np.uint64)
        pos = pos[pos < m]
        mod, div = native.debug_locate(pos, m, W, on_device, engine)
        assert np.array_equal(mod, pos) and np.array_equal(div, pos // np.uint64(W)), W


def test_mod_and_bucket_division_are_exact_host(native):
    check(native, False)


@pytest.mark.gpu
def test_mod_and_bucket_division_are_exact_device(native):
    eng = native.Engine(22, 3, 1000, 1 << 20, ["1" * 22, "1" * 10 + "0" + "1" * 12, "1" * 10 + "00" + "1" * 12])
    check(native, True, eng)
    eng.close()
