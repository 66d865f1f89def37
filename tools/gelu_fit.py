#!/usr/bin/env python3
"""Fit of the bf16 decoder kernels' GELU (csrc/decoder_fused.hip gelu_fast) to the erf form the reference uses
(nn.GELU, models/help_funcs.py:57):   gelu(z) ~ z * sigma(z * (k0 + k1 t + k2 t^2)),  t = min(z^2, 36).
Prints the coefficients (minimax over |z| <= 12, Nelder-Mead from the tanh-form constants), the maximum error of the value
and of the derivative the backward kernel uses, the same two figures for the textbook tanh constants, and the fp32 / exp2
evaluation the kernel performs.  CPU only (numpy + scipy)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

T_CLAMP = 36.0
z = np.linspace(-12, 12, 400001)
g = 0.5 * z * (1 + erf(z / np.sqrt(2)))
dg = 0.5 * (1 + erf(z / np.sqrt(2))) + z * np.exp(-z * z / 2) / np.sqrt(2 * np.pi)


def parts(k):
    t = np.minimum(z * z, T_CLAMP)
    a = z * (k[0] + t * (k[1] + t * k[2]))
    s = 1 / (1 + np.exp(-np.clip(a, -87, 87)))
    ap = k[0] + t * (3 * k[1] + t * 5 * k[2])
    return z * s, s + z * ap * (s - s * s)


def objective(k, w=0.25):
    v, d = parts(k)
    return max(np.abs(v - g).max(), w * np.abs(d - dg).max())


tanh_k = np.array([2 * np.sqrt(2 / np.pi), 2 * np.sqrt(2 / np.pi) * 0.044715, 0.0])
v, d = parts(tanh_k)
print("tanh-form constants : max |gelu err| %.3e   max |gelu' err| %.3e" % (np.abs(v - g).max(), np.abs(d - dg).max()))
r = minimize(objective, np.array([1.595, 7.4e-2, -7.03e-4]), method="Nelder-Mead",
             options=dict(xatol=1e-10, fatol=1e-13, maxiter=40000))
v, d = parts(r.x)
print("fitted k0 k1 k2     : %.16g %.16g %.16g" % tuple(r.x))
print("                      max |gelu err| %.3e   max |gelu' err| %.3e" % (np.abs(v - g).max(), np.abs(d - dg).max()))
# the kernel's evaluation: fp32, sigma(a) = 1 / (1 + exp2(-a log2 e)) with the constant folded into the coefficients
nl2e = np.float32(-1.4426950408889634)
c = [np.float32(x) * nl2e for x in r.x]
zf = z.astype(np.float32)
t = np.minimum(zf * zf, np.float32(T_CLAMP))
w = zf * (c[0] + t * (c[1] + t * c[2]))
with np.errstate(over="ignore"):
    s = (np.float32(1) / (np.float32(1) + np.exp2(w.astype(np.float64)).astype(np.float32))).astype(np.float32)
print("fp32 / exp2 form    : max |gelu err| %.3e" % np.abs(zf * s - g).max())
