"""The large-batch GEMM epilogue applies GELU as x (0.5 + x Q(x^2)) with a fixed degree-13 odd polynomial instead of
the erf form (encoder.hip: gelu_poly2).  This pins the tolerance: the coefficients compiled into the kernel, evaluated
in fp32 the way the kernel does (v_pk_fma_f32 steps), stay within 2e-4 absolute of 0.5 x (1 + erf(x / sqrt 2)) for every
x, reach exactly x (or exactly 0) beyond the clamp at |x| = 4, and are the ones tools/gelu_fit.py produces."""
import os
import re
import sys

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gelu_fit  # noqa: E402


def _kernel_coefficients():
    src = open(os.path.join(ROOT, "semantic_query_engine_amd", "csrc", "encoder.hip")).read()
    body = src[src.index("f32x2 gelu_poly2(f32x2 v)"):]
    body = body[:body.index("return __builtin_elementwise_fma(v, h")]
    vals = [float(m) for m in re.findall(r"f32x2\{(-?[0-9.e+-]+)f,", body)]
    assert len(vals) == 7, vals                      # highest power first, as Horner takes them
    return vals[::-1]


def test_polynomial_gelu_tolerance():
    c = _kernel_coefficients()
    x = np.concatenate([np.linspace(-8, 8, 400001), np.linspace(-1000, 1000, 20001), [0.0, 4.0, -4.0, 1e-8]])
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    got = gelu_fit.gelu_poly_f32(x, c)
    assert np.abs(got - ref).max() < 2e-4            # the tolerance of the large-batch GELU
    big = np.array([4.0, 5.0, 37.5, 1000.0])
    assert np.array_equal(gelu_fit.gelu_poly_f32(big, c), big.astype(np.float32))        # h(4) = 0.5 exactly
    assert np.all(gelu_fit.gelu_poly_f32(-big, c) == 0.0)


def test_kernel_coefficients_are_the_fit():
    want = gelu_fit.snap(gelu_fit.fit())
    got = np.array(_kernel_coefficients())
    assert np.allclose(got, want, rtol=1e-6, atol=0)
