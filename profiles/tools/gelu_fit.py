"""The degree-7 fit behind gelu_erf (pram_amd/csrc/linear.hip): q(a) ~ -log2(0.5 erfc(a / sqrt 2)) on [0, 6], least squares on
Chebyshev nodes weighted by a * s(a) (the error of GELU = t * Phi(t) is t * ds), coefficients rounded to fp32 and the result
re-evaluated in fp32 Horner form.  Prints the coefficients (constant term first) and the error of s and of t * s.
    python profiles/tools/gelu_fit.py [degree]"""
import sys
import numpy as np
from numpy.polynomial import polynomial as P
from scipy.special import erfc

A, deg = 6.0, int(sys.argv[1]) if len(sys.argv) > 1 else 7
x = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000)
a = (x + 1) * A / 2
s = 0.5 * erfc(a / np.sqrt(2))
c = P.polyfit(a, -np.log2(s), deg, w=np.maximum(s, 1e-12) * np.maximum(a, 0.05)).astype(np.float32)
t = np.linspace(0, A, 200001)
acc = np.full(t.shape, c[-1], np.float32)
for k in range(deg - 1, -1, -1):
    acc = acc * t.astype(np.float32) + c[k]
ds = np.exp2(-acc.astype(np.float64)) - 0.5 * erfc(t / np.sqrt(2))
print("coefficients (fp32, constant first):", [float(v) for v in c])
print(f"max |ds| = {np.abs(ds).max():.2e}, max |t ds| = {np.abs(t * ds).max():.2e} at t = {t[np.abs(t * ds).argmax()]:.3f}")
