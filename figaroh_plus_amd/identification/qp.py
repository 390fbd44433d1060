"""Dense strictly convex quadratic programs on the host: the ``quadprog.solve_qp`` call of
``src/figaroh/identification/identification_tools.py:429-463`` (``quadprog`` = Goldfarb & Idnani's dual active-set
method, a third-party dependency of the reference that is not vendored and not installed here; its published
algorithm -- D. Goldfarb, A. Idnani, "A numerically stable dual method for solving strictly convex quadratic
programs", Math. Programming 27 (1983) -- is restated).

The SIP program has 10 unknowns per massed body (400 for the human model) and 14 bound rows per body: microseconds of
data next to the regressor the Gram terms are reduced from, so the solve stays on the host like the reference's
(SURVEY section 8f-3); the device delivers ``W^T W`` and ``W^T tau`` (``figh_tsqr`` / ``figh_regressor_gram``).

    minimize  1/2 x^T G x - a^T x      subject to  C^T x >= b   (the first ``meq`` constraints are equalities)

The strictly convex problem has one solution, characterised by its KKT conditions (tests check those, and an
independent bounded-least-squares solve of the SIP fixture), so any correct solver returns what quadprog returns up to
rounding.
"""
import numpy as np
from scipy import linalg


def solve_qp(G, a, C=None, b=None, meq=0, factorized=False):
    """Same call and return convention as ``quadprog.solve_qp``: returns
    ``(x, f, xu, iterations, lagrangian, iact)`` -- solution, objective value, unconstrained minimiser, (additions,
    removals) of the active set, multipliers of all constraints and the (1-based) indices of the active ones.
    Raises ``ValueError`` like quadprog for a non positive definite G or an infeasible constraint set."""
    if factorized:
        raise NotImplementedError("factorized=True (G given as inverse Cholesky factor) is not used by the reference")
    G = np.asarray(G, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    n = a.shape[0]
    if G.shape != (n, n):
        raise ValueError("G must be %d x %d" % (n, n))
    if C is None:
        C = np.zeros((n, 0))
        b = np.zeros(0)
    C = np.asarray(C, dtype=np.float64).reshape(n, -1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    m = C.shape[1]
    if b.shape[0] != m:
        raise ValueError("C has %d columns, b has %d entries" % (m, b.shape[0]))
    try:
        L = np.linalg.cholesky(G)  # G = L L^T
    except np.linalg.LinAlgError:
        raise ValueError("matrix G is not positive definite")

    def Linv(v):  # L^-1 v
        return linalg.solve_triangular(L, v, lower=True)

    def LinvT(v):  # L^-T v
        return linalg.solve_triangular(L, v, lower=True, trans="T")

    xu = LinvT(Linv(a))
    x = xu.copy()
    f = -0.5 * a.dot(x)
    B = Linv(C)  # every constraint normal in the metric of G: columns L^-1 n_k
    norms = np.sqrt((C * C).sum(axis=0))
    norms[norms == 0.0] = 1.0
    active = []  # constraint indices, in the order they were added
    sgn = np.ones(m)  # an equality enters as n^T x >= b or as -n^T x >= -b, whichever is violated
    u = np.zeros(0)
    added = removed = 0
    eps = np.finfo(float).eps
    max_iter = 50 * (n + m) + 100

    # Q (n x n), R (n x q): L^-1 N = Q R for the active normals N, updated column by column (O(n^2) per change)
    Q, R = np.eye(n), np.zeros((n, 0))
    for _ in range(max_iter):
        s = C.T.dot(x) - b
        viol = s / norms
        # equalities first (either sign counts), then the most violated inequality (quadprog's rule)
        p = -1
        for k in range(meq):
            if k not in active and abs(viol[k]) > 1e3 * eps * (1.0 + abs(b[k]) / norms[k]):
                p = k
                sgn[k] = 1.0 if s[k] < 0 else -1.0
                break
        if p < 0:
            cand = viol.copy()
            cand[:meq] = 0.0
            if active:
                cand[active] = 0.0
            k = int(np.argmin(cand)) if m else -1
            if k < 0 or cand[k] >= -1e3 * eps * (1.0 + np.abs(x).max()):
                lagr = np.zeros(m)
                lagr[active] = u * sgn[active]
                return x, f, xu, np.array([added, removed]), lagr, np.array([i + 1 for i in active])
            p = k
        npl = sgn[p] * B[:, p]  # L^-1 n+
        sp = sgn[p] * s[p]
        uplus = np.append(u, 0.0)
        while True:
            q = len(active)
            d = Q.T.dot(npl)
            z = LinvT(Q[:, q:].dot(d[q:]))  # step direction in the primal space
            r = linalg.solve_triangular(R[:q, :], d[:q]) if q else np.zeros(0)
            zn = d[q:].dot(d[q:])  # z^T n+ = |Q2^T L^-1 n+|^2: zero when n+ depends on the active normals
            t2 = np.inf if zn <= 1e2 * eps * d.dot(d) else -sp / zn
            t1, l = np.inf, -1
            if q:
                ok = (np.asarray(active) >= meq) & (r > 0.0)  # equalities never leave
                if ok.any():
                    ratio = np.where(ok, uplus[:q] / np.where(ok, r, 1.0), np.inf)
                    l = int(np.argmin(ratio))
                    t1 = ratio[l]
            t = min(t1, t2)
            if not np.isfinite(t):
                raise ValueError("constraints are inconsistent, no solution")
            uplus[:q] -= t * r
            uplus[q] += t
            if np.isfinite(t2):
                x = x + t * z
                f += t * zn * (0.5 * t + uplus[q] - t)
            if t == t2:  # full step: the constraint becomes active
                active.append(p)
                u = uplus
                added += 1
                if q == 0:
                    Q, R = linalg.qr(npl.reshape(n, 1), mode="full")
                else:
                    Q, R = linalg.qr_insert(Q, R, npl, q, which="col")
                break
            # partial step (or dual step only): drop the blocking constraint and try again
            del active[l]
            uplus = np.delete(uplus, l)
            removed += 1
            Q, R = linalg.qr_delete(Q, R, l, which="col") if q > 1 else (np.eye(n), np.zeros((n, 0)))
            sp = sgn[p] * (C[:, p].dot(x) - b[p])
    raise ValueError("active-set iteration limit reached")
