"""A second, independent statement of CLIPPER's dense-clique solver — test infrastructure only.

Written from the PAPER (Lusk, Fathian, How: "CLIPPER: A Graph-Theoretic Framework for Robust Data
Association", ICRA 2021, Algorithms 1-2), not from oracle/clipper_oracle.c: dense NumPy matrices, the
penalised matrix M_d = M - d*(11' - C) formed explicitly, gradients by dense mat-vec.  It exists to
catch transcription mistakes in the one C restatement every parity test hangs on
(tests/test_dense_clipper.py cross-checks the two on small problems).

Conventions: M symmetric with its diagonal included (1 for plain CLIPPER, the single scores for the
ROMAN invariant); C symmetric 0/1 with ones on the diagonal; u0 = ones unless given.
"""
import numpy as np


def find_dense_clique(M, C, u0=None, tol_u=1e-8, tol_F=1e-9, maxiniters=200, maxoliters=1000, beta=0.25,
                      maxlsiters=99, eps=1e-9, rescale_u0=True):
    M = np.asarray(M, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    n = M.shape[0]
    Cb = 1.0 - C                                    # complement of the consistency graph; zero diagonal
    u = np.ones(n) if u0 is None else np.asarray(u0, dtype=np.float64).copy()
    if rescale_u0:
        u = M @ u                                    # one power-iteration step
    nu = np.linalg.norm(u)
    if nu > 0:
        u = u / nu
    passes = 2 if rescale_u0 else 1

    def penalty_step(u, absval):
        Mu, Cbu = M @ u, Cb @ u
        act = (Cbu > eps) & (u > eps)
        if not act.any():
            return None
        r = Mu[act] / Cbu[act]
        return float(np.mean(np.abs(r) if absval else r))

    d = penalty_step(u, False) or 0.0
    inner = trials = updates = 0
    for outer in range(maxoliters):
        Md = M - d * Cb
        g = Md @ u
        F = float(u @ g)
        for _ in range(maxiniters):
            alpha = 1.0
            for _k in range(maxlsiters):
                un = np.maximum(u + alpha * g, 0.0)
                nn = np.linalg.norm(un)
                if nn > 0:
                    un = un / nn
                gn = Md @ un
                Fn = float(un @ gn)
                trials += 1
                dF = Fn - F
                if dF < -eps:
                    alpha *= beta
                else:
                    break
            du = np.linalg.norm(un - u)
            u, g, F = un, gn, Fn
            inner += 1
            if du < tol_u or abs(dF) < tol_F:
                break
        inc = penalty_step(u, True)
        updates += 1
        if inc is None:
            outer_done = outer
            break
        d += inc
    else:
        outer_done = maxoliters
    omega = int(np.floor(F + 0.5)) if F >= 1.0 - 0.5 else 0
    omega = max(0, min(omega, n))
    # the omega largest entries, descending; ties by descending index like a (value, index) min-heap pops
    order = sorted(range(n), key=lambda i: (u[i], i), reverse=True)
    nodes = np.array(order[:omega], dtype=np.int64)
    # n_pass: the published count (one product per line-search trial; M u, C u of the accepted trial carried to the d update);
    # n_pass_fused: the count of an implementation that forms M_d u' as ONE product in the line search — as this statement
    # does — and therefore needs M u and C u apart, one more pass, at every d update (the device's stream solver)
    return dict(nodes=nodes, u=u, F=F, d=d, inner_iters=inner, ls_trials=trials, outer_iters=outer_done,
                n_pass=passes + trials, n_pass_fused=passes + trials + updates)
