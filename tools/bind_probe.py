import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from pypyls_amd.engine import Engine, PLSX_REGRESSION
S, B, T, k = 1000, 100000, 20, 15
rs = np.random.RandomState(0)
X = rs.randn(S, B); Y = rs.randn(S, T) + 0.3 * X[:, :T]
Yc = Y - Y.mean(0)
eng = Engine()
def t(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ms, dX = t(lambda: eng._dev(X, np.float64)); print('H2D X pageable', round(ms, 2))
ms, _ = t(lambda: eng.set_data(dX, eng._dev(Yc, np.float64), np.zeros(S, np.int32), 1, k, PLSX_REGRESSION)); print('set_data from device tensor', round(ms, 2))
ms, _ = t(lambda: eng.set_data_regression(X, Yc, k)); print('set_data_regression from host', round(ms, 2))
ms, _ = t(lambda: bool(torch.isfinite(eng.colmean_dev()).all().item())); print('finite check', round(ms, 2))
eng.k = k
ms, _ = t(lambda: eng.simpls_decompose()); print('simpls_decompose', round(ms, 2))
