"""Fit of the sigmoid-form GELU used by csrc/mlp.hip:  gelu(x) ~= x / (1 + exp(-x q(x^2))), q of degree 4
(`fit_gelu.py 3`: 3 coefficients with x^2 clamped at 64 -- gelu1 of csrc/mlp_common.h, the un-packed form of mlp32.hip).
Iteratively re-weighted least squares towards the minimax of the absolute GELU error on [0, 9] (the form is
odd-symmetric in the error); prints the coefficients of q and of -log2(e) q (what the kernel holds)."""
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.special import erfc

xs = np.linspace(0, 9, 18001)
NCOEF = int(sys.argv[1]) if len(sys.argv) > 1 else 5
TCLAMP = 64.0 if NCOEF < 5 else np.inf  # a short q has a positive leading term: hold x^2 where q is still negative


def gelu_err(c, x=xs):
    p = x * np.polyval(c[::-1], np.minimum(x * x, TCLAMP))
    with np.errstate(over="ignore"):
        s = 1.0 / (1.0 + np.exp(-p))
    return x * (s - 0.5 * erfc(-x / np.sqrt(2)))


c = np.array([1.5976, 0.07056, 0.0, 0.0, 0.0])[:NCOEF]
w = np.ones_like(xs)
best = (np.inf, c)
for _ in range(60):
    c = least_squares(lambda c: gelu_err(c) * w, c, method="lm", xtol=1e-15, ftol=1e-15).x
    e = np.abs(gelu_err(c))
    if e.max() < best[0]:
        best = (e.max(), c.copy())
    w = w * (1 + 2.0 * e / e.max())
    w /= w.mean()
m, c = best
print("max |gelu error| on [0, 9]:", m, " on [0, 40]:", np.abs(gelu_err(c, np.linspace(0, 40, 400001))).max())
print("q:", ["%.10g" % v for v in c])
print("-log2(e) q:", ["%.9e" % v for v in -np.log2(np.e) * c])
