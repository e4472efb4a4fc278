"""Executable model (numpy, fp64) of the DEVICE algorithm of the solver — the windowed line search
of clipper_amd/csrc/kernels.hip.h, iteration by iteration, with the same phases, tables, point
slots and decision walk. Test infrastructure: tests/test_window_logic.py checks on the CPU that
for every window size V it takes exactly the trials, decisions and result of the oracle's plain
restatement of findDenseClique (clipper.cpp:172-323); the GPU tests check the kernels.

Pass kinds (k_gemv):   window mode  g_v = (M_off + d*C_off) x_v   for the V candidates of a table
                       pair mode    a = M_off x, b = C_off x       for candidate 0
Iteration kinds:       pass iteration (G streams, T = tail) / transition iteration (sweeps)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PH_NORMALIZE, PH_RESCALE, PH_INIT, PH_TRIAL, PH_PENALTY = range(5)


@dataclass
class Params:
    tol_u: float = 1e-8
    tol_F: float = 1e-9
    maxiniters: int = 200
    maxoliters: int = 1000
    beta: float = 0.25
    maxlsiters: int = 99
    eps: float = 1e-9
    rescale_u0: bool = True


@dataclass
class Result:
    u: np.ndarray
    F: float
    d: float
    ifinal: int
    n_passes: int
    n_trials: int
    n_iters: int
    accepted: list = field(default_factory=list)   # window index of every accepted trial


def solve(Moff: np.ndarray, Coff: np.ndarray, u0: np.ndarray, P: Params, V: int) -> Result:
    """Moff, Coff: dense symmetric, zero diagonal."""
    m = len(u0)
    tables = np.zeros((V + 1, m, V))          # candidate tables of the pending set
    nrm, sx = np.ones(V), np.zeros(V)
    # current point and the V point slots the tail fills
    u = np.zeros(m)
    g = np.zeros(m)
    slot_u, slot_g = np.zeros((V, m)), np.zeros((V, m))
    a = b = None
    d = F = s = 0.0
    alpha = 1.0
    i_ = j_ = k_ = 0
    n_passes = n_trials = n_iters = 0
    sel = 0
    accepted = []
    phase = PH_RESCALE if P.rescale_u0 else PH_NORMALIZE
    tables[0][:, 0] = u0
    results_pending = not P.rescale_u0       # PH_NORMALIZE consumes no pass
    sums = None

    def build_window(base_u, base_g, alpha0):
        tab = np.zeros((m, V))
        n, sxs = np.ones(V), np.zeros(V)
        al = alpha0
        for l in range(V):
            t = np.maximum(base_u + al * base_g, 0.0)
            tab[:, l] = t
            z = float(t @ t)
            n[l] = np.sqrt(z) if z > 0 else 1.0
            sxs[l] = float(t.sum()) / n[l]
            al = al * P.beta
        return tab, n, sxs

    while True:
        n_iters += 1
        do_pass = True
        if results_pending:
            # ---- the decision at the head of G ------------------------------------------------
            fast = False
            begin_outer = penalty = need_window = need_pair = finished = False
            if phase == PH_TRIAL:
                jstar = -1
                for v in range(V):
                    n_trials += 1
                    Fnew = sums["F"][v]
                    deltaF = Fnew - F
                    accept = True
                    if deltaF < -P.eps:
                        alpha = alpha * P.beta
                        k_ += 1
                        if k_ < P.maxlsiters:
                            accept = False
                    if accept:
                        jstar = v
                        break
                if jstar < 0:
                    sel, nrm, sx = V, sums["none_nrm"], sums["none_sx"]
                    fast = True
                else:
                    accepted.append(jstar)
                    deltau = np.sqrt(sums["du2"][jstar])
                    s = sx_pass[jstar]
                    F = Fnew
                    u, g = slot_u[jstar].copy(), slot_g[jstar].copy()
                    j_ += 1
                    if deltau < P.tol_u or abs(deltaF) < P.tol_F or j_ >= P.maxiniters:
                        # pair-mode pass straight on the accepted x (its point slot), same iteration
                        need_pair = fast = True
                        phase = PH_PENALTY
                    else:
                        alpha, k_ = 1.0, 0
                        sel, nrm, sx = jstar, sums["nrm"][jstar], sums["sx"][jstar]
                        fast = True
            if not fast:
                # ---- transition iteration (workgroup (0,0)) ---------------------------------
                do_pass = False
                to_init = False
                if phase in (PH_NORMALIZE, PH_RESCALE):
                    u = (a + u0) if phase == PH_RESCALE else u0.copy()
                    u = u / np.sqrt(float(u @ u))
                    tab = np.zeros((m, V))
                    tab[:, 0] = u
                    new_tables = {0: tab}
                    sel, nrm = 0, np.ones(V)
                    next_phase, to_init = PH_INIT, True
                elif phase == PH_INIT:
                    s = float(u.sum())
                    cbu = s - b - u
                    idx = (cbu > P.eps) & (u > P.eps)
                    d = float(np.mean((a[idx] + u[idx]) / cbu[idx])) if idx.any() else 0.0
                    i_ = 0
                    begin_outer = True
                elif phase == PH_PENALTY:
                    penalty = True
                next_phase = PH_TRIAL if not to_init else next_phase
                new_tables = new_tables if to_init else {}
                while not to_init:
                    if penalty:
                        cbu = s - b - u
                        idx = (cbu > P.eps) & (u > P.eps)
                        penalty = False
                        if idx.any():
                            d += float(np.mean(np.abs((a[idx] + u[idx]) / cbu[idx])))
                            i_ += 1
                            begin_outer = True
                        else:
                            finished = True
                            break
                    if begin_outer:
                        begin_outer = False
                        if i_ >= P.maxoliters:
                            finished = True
                            break
                        g = (1 + d) * u - d * s + a + b * d
                        F = float(u @ g)
                        j_ = 0
                        if P.maxiniters <= 0:
                            penalty = True
                            continue
                        alpha, k_ = 1.0, 0
                        need_window = True
                    break
                if finished:
                    return Result(u, F, d, i_, n_passes, n_trials, n_iters, accepted)
                if need_window:
                    tab, nrm, sx = build_window(u, g, alpha)
                    new_tables = {0: tab}
                    sel = 0
                for k, t in new_tables.items():
                    tables[k] = t
                phase = next_phase
                results_pending = False
                continue
        # ---- pass iteration: G streams M against table `sel`, T = tail ----------------------
        n_passes += 1
        X = tables[sel]
        if phase != PH_TRIAL:                       # pair mode, candidate 0 (nrm = 1)
            x = u if phase == PH_PENALTY else X[:, 0]
            a, b = Moff @ x, Coff @ x
            results_pending = True
            continue
        W = Moff + d * Coff                         # window mode
        graw = W @ X                                # (m, V)
        sums = {"F": np.zeros(V), "du2": np.zeros(V), "nrm": {}, "sx": {}}
        sx_pass = sx.copy()
        out_tables = {}
        for v in range(V):
            xi = X[:, v] / nrm[v]
            gn = (1 + d) * xi - d * sx[v] + graw[:, v] / nrm[v]
            slot_u[v], slot_g[v] = xi, gn
            sums["F"][v] = float(xi @ gn)
            du = xi - u
            sums["du2"][v] = float(du @ du)
            out_tables[v], sums["nrm"][v], sums["sx"][v] = build_window(xi, gn, 1.0)
        al = alpha
        for _ in range(V):
            al = al * P.beta
        out_tables[V], sums["none_nrm"], sums["none_sx"] = build_window(u, g, al)
        for k, t in out_tables.items():
            tables[k] = t
        results_pending = True
