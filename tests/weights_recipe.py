"""Deterministic G / D weights from a numpy RandomState recipe.

Shared by tests/golden/make_golden.py (which loads them into the *reference* modules to make the
goldens) and by the tests (which load them into the build's modules / the oracle), so that 8 MB of
weights need not be committed.  Input: ordered (key, shape) pairs of a state_dict.
"""
import numpy as np


def seeded_state_arrays(keys_shapes, seed):
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in keys_shapes:
        shp = tuple(int(s) for s in shp)
        if k.endswith('weight_u') or k.endswith('weight_v'):
            a = rs.randn(*shp)                      # replaced below by power-iteration vectors
        elif 'gain0' in k:
            a = 1.0 + 0.1 * rs.uniform(-1, 1, shp)
        elif len(shp) == 1 or 'bias0' in k:
            a = 0.1 * rs.uniform(-1, 1, shp)
        elif k.endswith('weight_orig') and len(shp) == 4:
            # positive-mean weights: zero-mean ones average out under the global pool and make D's
            # scores (and input gradients) nearly input-independent, which would be a weak test
            fan_in = int(np.prod(shp[1:]))
            a = rs.uniform(-0.6, 1.0, shp) / np.sqrt(fan_in) * 1.7
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rs.uniform(-1, 1, shp) / np.sqrt(fan_in) * 1.7
        out[k] = np.asarray(a, dtype=np.float64)
    # spectral-norm buffers: 12 float64 power iterations from the seeded start so that
    # sigma = u^T W v is close to the true top singular value (a random u,v would make W/sigma blow up)
    for k in list(out.keys()):
        if k.endswith('weight_orig'):
            base = k[:-len('weight_orig')]
            W = out[k].reshape(out[k].shape[0], -1)
            u = out[base + 'weight_u']
            u = u / np.linalg.norm(u)
            for _ in range(12):
                v = W.T @ u
                v = v / np.linalg.norm(v)
                u = W @ v
                u = u / np.linalg.norm(u)
            out[base + 'weight_u'] = u
            out[base + 'weight_v'] = v
    return {k: v.astype(np.float32) for k, v in out.items()}


def digest(a, n=256):
    """Small fingerprint of a large gradient tensor: (sum, abs-sum, strided sample)."""
    f = np.asarray(a, dtype=np.float64).ravel()
    stride = max(1, f.size // n)
    return np.float64(f.sum()), np.float64(np.abs(f).sum()), f[::stride][:n].astype(np.float32)
