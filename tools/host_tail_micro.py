"""Host-side costs of the n x n tail on the GPU box (many-core host: BLAS threading overhead on tiny matrices?)."""
import time
import numpy as np
rng = np.random.default_rng(0)
R1 = np.triu(rng.standard_normal((36, 36))) + 5 * np.eye(36); R2 = rng.standard_normal((36, 13)); z = rng.standard_normal(36)
def t(f, n=2000):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("np.linalg.inv(36x36)            %.1f us" % t(lambda: np.linalg.inv(R1)))
print("inv @ R2 (36x36 @ 36x13)        %.1f us" % t(lambda: R1 @ R2))
print("np.around(.,6)                  %.1f us" % t(lambda: np.around(R2, 6)))
print("np.triu(50x50)                  %.1f us" % t(lambda: np.triu(rng.standard_normal((50, 50)))))
import scipy.linalg as sl
print("scipy solve_triangular 36x13    %.1f us" % t(lambda: sl.solve_triangular(R1, R2)))
try:
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):
        print("np.linalg.inv, 1 BLAS thread    %.1f us" % t(lambda: np.linalg.inv(R1)))
        print("36x36 @ 36x13, 1 BLAS thread    %.1f us" % t(lambda: R1 @ R2))
except Exception as e:
    print("threadpoolctl:", e)
