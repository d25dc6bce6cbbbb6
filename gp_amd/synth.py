"""Deterministic synthetic inputs for benches and tests: counter-based
SplitMix64, so every rank / box regenerates identical bits from integers (no files).
X ~ U[0,1)^(n x D) column-major; y = sin(2 pi sum_d X) + 0.1 eps, eps by Box-Muller."""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return x ^ (x >> np.uint64(31))


def _u01(seed, idx):
    with np.errstate(over="ignore"):
        k = (np.uint64(seed) * np.uint64(0x100000001B3) + idx.astype(np.uint64)) & _M
    return (_splitmix64(k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synth(n, D, seed=20240601):
    idx = np.arange(n * D, dtype=np.uint64)
    X = np.asfortranarray(_u01(seed, idx).reshape(D, n).T)
    i2 = np.arange(n, dtype=np.uint64) * np.uint64(2)
    u1 = np.maximum(_u01(seed + 1, i2), 1e-300)
    u2 = _u01(seed + 1, i2 + np.uint64(1))
    eps = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    s = np.zeros(n)
    for d in range(D):
        s = s + X[:, d]
    y = np.sin(2.0 * np.pi * s) + 0.1 * eps
    return X, y
