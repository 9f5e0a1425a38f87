#!/usr/bin/env python3
"""Fit of the 16-bit GEMM epilogues' GELU (svol_amd/csrc/common.h::gelu_parts2): Phi(x) ~ sigmoid(x (p0 + p1 x^2 + p2 x^4)), minimax
against the exact erf form over |x| <= 8 for gelu(x) = x Phi(x) and, at half weight, its derivative.  Prints the coefficients and the
errors of the fitted form, of the usual tanh form (same shape, p2 = 0) and of the fp32 evaluation the kernel does (clamped x^2)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-8, 8, 160001)
Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
gelu, dgelu = x * Phi, Phi + x * np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def errs(p, xs=x, g=gelu, dg=dgelu):
    x2 = np.minimum(xs * xs, 64.0)
    q = np.polyval(p[::-1], x2)
    dq = np.polyval(np.polyder(np.poly1d(p[::-1])), x2) if len(p) > 1 else 0.0
    s = 1 / (1 + np.exp(-xs * q))
    return np.abs(xs * s - g).max(), np.abs(s + xs * s * (1 - s) * (q + 2 * x2 * dq) - dg).max()


std = np.array([2 * np.sqrt(2 / np.pi), 2 * np.sqrt(2 / np.pi) * 0.044715])
print('tanh form          : |gelu err| %.2e  |gelu\' err| %.2e' % errs(std))
r = minimize(lambda p: max(errs(p)[0], 0.5 * errs(p)[1]), np.append(std, 0.0), method='Nelder-Mead',
             options=dict(xatol=1e-10, fatol=1e-11, maxiter=40000))
print('fitted, 3 terms    :', [float(v) for v in r.x], ' |gelu err| %.2e  |gelu\' err| %.2e' % errs(r.x))
xs = np.linspace(-30, 30, 600001)
P = 0.5 * (1 + erf(xs / np.sqrt(2)))
print('same, |x| <= 30    : |gelu err| %.2e  |gelu\' err| %.2e' % errs(r.x, xs, xs * P, P + xs * np.exp(-xs * xs / 2) / np.sqrt(2 * np.pi)))
