"""Seeded input recipes shared by tests/golden/make_golden.py and the tests (the
fixtures hold only expected outputs; inputs are regenerated from these recipes)."""
import numpy as np


def cosine_cases(seed=20240901):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((64, 1024)).astype(np.float32)
    b = rng.standard_normal((64, 1024)).astype(np.float32)
    b[:16] = a[:16] + 0.05 * b[:16]
    a[16] = 0.0
    b[17] = 0.0
    a[18] = 0.0; b[18] = 0.0
    a[19, 5] = np.nan
    b[20] = a[20]
    b[21] = -a[21]
    a[22] = np.float32(1e-30) * a[22]
    a[23] *= np.float32(1e15); b[23] *= np.float32(1e15)
    a[24] = 0.0; a[24, 7] = 3.0; b[24] = 0.0; b[24, 7] = 2.0
    return a, b


def normalize_case(seed=7):
    rngn = np.random.default_rng(seed)
    e = (rngn.standard_normal((32, 1024)) * rngn.uniform(0.01, 50.0, (32, 1))).astype(np.float32)
    e[3] = 0.0
    e[4] = 0.0; e[4, 100] = 1e-20
    e[5] *= np.float32(1e-12)
    return e


def knn_case(seed=11):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((4096, 1024)).astype(np.float32)
    q = rng.standard_normal((16, 1024)).astype(np.float32)
    for i in range(8):
        q[i] = x[37 * i + 5] + 0.1 * q[i]
    x[3000:3040] = x[5]
    x[100] = 0.0
    x[200] = 2.5 * x[42]
    q[15] = 0.0
    return x, q


def cache_base(seed=5):
    return np.random.default_rng(seed).standard_normal((12, 1024)).astype(np.float32)
