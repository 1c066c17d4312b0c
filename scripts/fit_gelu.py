import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares, minimize
x = np.concatenate([np.linspace(0, 2, 6001), np.linspace(2, 9, 6001)])
z = x / np.sqrt(2)
target = 0.5 * x * erfc(z)                      # the part of GELU the approximation produces
def model(c, z, x):
    p = c[0]; a = c[1:]
    t = 1.0 / (1.0 + p * z)
    q = np.zeros_like(t)
    for ak in a[::-1]:
        q = q * t + ak
    return 0.5 * x * t * q * np.exp(-z * z)
best = None
for deg in (5, 6, 7):
    c0 = np.concatenate([[0.3275911], [0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429], np.zeros(deg - 5 + 1)])[:deg + 2]
    r = least_squares(lambda c: (model(c, z, x) - target), c0, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
    c = r.x
    # minimax refinement
    f = lambda c: np.abs(model(c, z, x) - target).max()
    for _ in range(6):
        m = minimize(f, c, method="Nelder-Mead", options=dict(xatol=1e-14, fatol=1e-16, maxiter=40000, maxfev=40000))
        c = m.x
    print(deg, "max abs err (fp64 eval)", f(c), "coeffs", repr(c))
    best = (deg, c)
