#!/usr/bin/env python3
"""Generates the coefficients of fast_atan2 (csrc/nid_device.hpp): atan(t) = t * P(t^2) on t in [0,1],
P = Chebyshev interpolant of atan(sqrt(s))/sqrt(s) on s in [0,1] converted to the monomial basis with
50-digit arithmetic, rounded to double.  Prints the table and the measured max error of the double
Horner evaluation.  Usage: gen_atan_poly.py [terms]"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20  # number of terms (degree n-1 in s)


def g(s):
    if s == 0:
        return mp.mpf(1)
    r = mp.sqrt(s)
    return mp.atan(r) / r


# Chebyshev nodes on [0,1]
nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * n)) + 1) / 2 for k in range(n)]
vals = [g(x) for x in nodes]
# Chebyshev coefficients
c = []
for j in range(n):
    s = mp.mpf(0)
    for k in range(n):
        s += vals[k] * mp.cos(mp.pi * j * (2 * k + 1) / (2 * n))
    c.append(s * 2 / n)
c[0] /= 2
# to monomial in y = 2s - 1, then substitute
T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
for j in range(2, n):
    a = [mp.mpf(0)] + [2 * v for v in T[j - 1]]
    b = T[j - 2] + [mp.mpf(0)] * (len(a) - len(T[j - 2]))
    T.append([x - y for x, y in zip(a, b)])
py = [mp.mpf(0)] * n
for j in range(n):
    for i, v in enumerate(T[j]):
        py[i] += c[j] * v
# y = 2s - 1: expand
ps = [mp.mpf(0)] * n
for i, a in enumerate(py):
    # (2s-1)^i
    for k in range(i + 1):
        ps[k] += a * mp.binomial(i, k) * (mp.mpf(2) ** k) * (mp.mpf(-1) ** (i - k))
coef = [float(v) for v in ps]
print("// %d terms" % n)
for v in coef:
    print("  %.17e," % v)

# error of the double Horner evaluation
t = np.linspace(0.0, 1.0, 200001)
s = t * t
p = np.full_like(s, coef[-1])
for v in coef[-2::-1]:
    p = p * s + v
approx = t * p
exact = np.array([float(mp.atan(mp.mpf(float(x)))) for x in t[::50]])
print("max abs err (double Horner, sampled):", np.max(np.abs(approx[::50] - exact)))
