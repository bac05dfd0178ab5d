"""Fit of the sigmoid-form GELU used by the 16-bit kernels (csrc/common.h: gelu_phi_fast) and its error scan in fp32 arithmetic."""
import numpy as np
from scipy.special import log_ndtr, ndtr, erf
from scipy.optimize import minimize
def logit_phi(x): return log_ndtr(x) - log_ndtr(-x)
XM = 5.0
xs = np.concatenate([np.linspace(1e-4, XM, 40001)])
t = logit_phi(xs)
best = {}
for deg in (7, 11, 13):
    pw = np.arange(1, deg+1, 2)
    A = np.stack([xs**k for k in pw], 1)
    w = np.ones_like(xs)
    for it in range(1500):
        c, *_ = np.linalg.lstsq(A*w[:,None], t*w, rcond=None)
        e = np.abs(A@c - t)
        w = w*(1+ 1.0*e/e.max()); w/=w.mean()
    print(deg, "max |dp|", e.max(), list(c))
    best[deg] = c
# evaluate in float32 arithmetic
def gelu_sig(x, c):
    x = x.astype(np.float32)
    k = (-np.asarray(c) * np.log2(np.e)).astype(np.float32)
    x2 = x*x
    q = np.float32(k[-1])
    for kk in k[-2::-1]:
        q = (q * x2 + np.float32(kk)).astype(np.float32)
    u = (q * x).astype(np.float32)
    with np.errstate(over='ignore'):
        e = np.exp2(u.astype(np.float64)).astype(np.float32)
    phi = (np.float32(1) / (np.float32(1) + e)).astype(np.float32)
    return (x * phi).astype(np.float32), phi
x = np.linspace(-9, 9, 2000001)
exact = 0.5*x*(1+erf(x/np.sqrt(2)))
exact_d = ndtr(x) + x*np.exp(-x*x/2)/np.sqrt(2*np.pi)
for deg, c in best.items():
    g, phi = gelu_sig(x, c)
    rel = np.abs(g - exact)/np.maximum(np.abs(exact), 1e-30)
    m = np.abs(x) < 5.0
    print("deg", deg, "max rel (|x|<5.5)", rel[m].max(), "2^-9=", 2**-9, "max abs all", np.abs(g-exact).max(), "abs at |x|>5.5", np.abs(g-exact)[~m].max())
    e2 = np.exp2((-(x*x)*0.72134752).astype(np.float32))
    d = (x*0.39894228*e2 + phi).astype(np.float32)
    print("   derivative max abs", np.abs(d-exact_d).max(), "max rel where |d|>1e-3", (np.abs(d-exact_d)/np.abs(exact_d))[np.abs(exact_d)>1e-3].max())
    # monotonic polynomial check
    xx = np.linspace(0, 200, 200001)
    p = sum(cc*xx**k for cc,k in zip(c, np.arange(1,deg+1,2)))
    print("   p monotone:", np.all(np.diff(p)>0))
