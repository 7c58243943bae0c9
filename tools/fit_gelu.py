"""Derives and checks the coefficients of the kernels' erf-GELU (pcdms_amd/csrc/pcdm_device.h: PCDM_GELU_Q0..Q5).

    gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),      Phi(-t) = 2^Q(t),  t = min(|x|, 6)

Q = degree-5 polynomial fitted to log2 Phi(-t) on [0, 6] by iteratively re-weighted least squares in the Chebyshev basis (an
approximate minimax fit of the ABSOLUTE error of the result: weight t Phi(-t)).  Prints the monomial coefficients (fp32), then
evaluates the fp32 Horner form on a dense grid of [-8, 8] against the fp64 erfc form: the bound quoted in pcdm_device.h and
asserted by tests/test_kernels.py::test_gelu_accuracy.  CPU only (numpy / scipy).
"""
import numpy as np
from scipy.special import erfc, log_ndtr

T, DEG = 6.0, 5


def fit():
    t = np.linspace(0, T, 400001)
    q = log_ndtr(-t) / np.log(2)
    w = t * np.exp(log_ndtr(-t)) + 1e-6
    best = None
    for _ in range(200):
        c = np.polynomial.chebyshev.chebfit(2 * t / T - 1, q, DEG, w=w)
        f = np.polynomial.chebyshev.chebval(2 * t / T - 1, c)
        err = np.abs(t * (2.0 ** f - 2.0 ** q))
        if best is None or err.max() < best[0]:
            best = (err.max(), c.copy())
        w = w * (1 + 2 * err / err.max())
    cheb = np.polynomial.chebyshev.Chebyshev(best[1], domain=[0, T])
    return best[0], cheb.convert(kind=np.polynomial.Polynomial).coef


def check(p):
    x = np.linspace(-8, 8, 2000001).astype(np.float32)
    pc = [np.float32(v) for v in p]
    tc = np.minimum(np.abs(x), np.float32(T))
    q = np.full_like(tc, pc[DEG])
    for k in range(DEG - 1, -1, -1):
        q = (q * tc + pc[k]).astype(np.float32)
    g = (np.maximum(x, 0) - tc * np.exp2(q).astype(np.float32)).astype(np.float32)
    ex = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2))
    ae = np.abs(g - ex)
    m = np.abs(ex) >= 2e-3
    return ae.max(), (ae[m] / np.abs(ex[m])).max()


if __name__ == "__main__":
    e, p = fit()
    print("weighted minimax error of the fit (exact arithmetic): %.3e" % e)
    for i, v in enumerate(p):
        print("#define PCDM_GELU_Q%d (%rf)" % (i, float(np.float32(v))))
    a, r = check(p)
    print("fp32 Horner on [-8, 8]: max |err| %.3e; max relative error where |gelu| >= 2e-3: %.3e (2^-11 = %.3e)" % (a, r, 2.0 ** -11))
