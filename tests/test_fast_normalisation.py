"""The reciprocal form of GetResponse's normalisation (Mapper.cpp:852: response = sum / (nBeams * 100)).

csrc/scan_matcher.hip evaluates it as q = sum * inv, q' = fma(fma(-q, d, sum), inv, q) with inv = RN(1 / d) -- three
instructions instead of an IEEE division -- but only when lslam_matcher_create has found the two forms bit-equal for EVERY
numerator a response sum can take (0 .. nBeams * 100).  This is the same exhaustive check on the host, for the beam counts the
GPU tests use: it documents that the fast path is the one that runs, and that it is exact."""
import ctypes
import ctypes.util

import numpy as np
import pytest

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double]


@pytest.mark.parametrize("n_beams", [360, 721, 1080, 1081, 2500])
def test_reciprocal_form_equals_the_division_for_every_numerator(n_beams):
    d = float(n_beams * 100)
    inv = 1.0 / d
    sums = np.arange(0, n_beams * 100 + 1, dtype=np.float64)
    exact = sums / d  # IEEE division, elementwise
    q = sums * inv
    # numpy has no fma: the two fused steps through libm's (correctly rounded) fma, on the numerators where the plain
    # product already differs from the quotient or might after the correction -- i.e. all of them, vectorised in chunks
    fast = np.empty_like(q)
    fma = _libm.fma
    for i in range(sums.size):
        r = fma(-q[i], d, sums[i])
        fast[i] = fma(r, inv, q[i])
    assert np.array_equal(fast.view(np.uint64), exact.view(np.uint64))
