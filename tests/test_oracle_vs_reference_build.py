"""The oracle against a build of the REFERENCE ITSELF (oracle/_ref/libmrs_tg_ref.so, made by oracle/build_ref.sh from the
reference's own polynomial.cpp and rpoly/rpoly_ak1.cpp + our C harness).  Dormant in the image this repository was built in:
Eigen3 is not installed there, the library cannot exist, and every test here SKIPS -- which is what "parity unpinned" means
(oracle/REF_BUILD.md).  On an image with Eigen3 the same tests pin oracle/mto_poly.c against the reference's Jenkins-Traub
root finder, its candidate selection (quirk B7 included) and its base-coefficient table."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libmrs_tg_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref is empty: the reference needs Eigen3, absent from this image "
                                                                "(oracle/REF_BUILD.md) -- parity unpinned")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    L.ref_find_roots.restype = C.c_int
    L.ref_find_roots.argtypes = [dp, C.c_int, dp, dp, C.c_int]
    L.ref_min_max_candidates.restype = C.c_int
    L.ref_min_max_candidates.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_int, dp, C.c_int]
    L.ref_base_coeffs.restype = None
    L.ref_base_coeffs.argtypes = [C.c_int, C.c_int, C.c_double, dp]
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_jenkins_traub_roots_equal_the_references(ref):
    rng = np.random.default_rng(1)
    for n in (3, 5, 8, 9):
        for _ in range(50):
            c = rng.normal(size=n + 1)
            re, im = np.zeros(32), np.zeros(32)
            m = ref.ref_find_roots(_dp(c), n + 1, _dp(re), _dp(im), 32)
            ours = po.find_roots(c)
            assert m == ours.size
            theirs = np.sort_complex(re[:m] + 1j * im[:m])
            assert np.array_equal(np.sort_complex(ours), theirs)   # the same algorithm on the same doubles: the same bits


def test_base_coefficients_equal_the_references(ref):
    for d in range(5):
        for t in (0.0, 0.37, 1.0, 2.5):
            out = np.zeros(10)
            ref.ref_base_coeffs(10, d, t, _dp(out))
            ours = np.array([(np.prod([k - j for j in range(d)]) if k >= d else 0.0) * (t ** (k - d) if k >= d else 0.0) for k in range(10)])
            assert np.allclose(out, ours, rtol=1e-15, atol=0)


def test_candidates_of_a_near_double_root_follow_the_imag_filter(ref):
    """quirk B7 on the reference itself: a derivative with a complex pair of tiny imaginary part loses both roots"""
    a = 0.9
    for delta in (1e-6, 0.0, -1e-10):
        v = np.polynomial.polynomial.polyadd([1.0], 0.3 * np.polynomial.polynomial.polysub(
            np.polynomial.polynomial.polypow([-a, 1.0], 3) / 3.0, delta * np.array([-a, 1.0])))
        p = np.polynomial.polynomial.polyint(v)
        out = np.zeros(16)
        m = ref.ref_min_max_candidates(_dp(np.ascontiguousarray(p)), p.size, 0.0, 2.0, 1, _dp(out), 16)
        c = np.zeros((4, 10))
        c[2, :p.size] = p
        assert m >= 2 and abs(np.max(np.abs(np.polynomial.polynomial.polyval(out[:m], v))) - po.segment_max_magnitude(c, 2.0, 1, [2])) < 1e-15
