"""Executable model (numpy, fp64) of the DEVICE algorithm of the solver — the windowed line search
of clipper_amd/csrc/kernels.hip.h, iteration by iteration, with the same phases, tables, point
slots and decision walk. Test infrastructure: tests/test_window_logic.py checks on the CPU that
for every window size V it takes exactly the trials, decisions and result of the oracle's plain
restatement of findDenseClique (clipper.cpp:172-323); the GPU tests check the kernels.

Pass kinds (k_gemv):   window mode  candidate 0: a = M_off x_0, b = C_off x_0 apart;
                                    candidates v >= 1: g_v = (M_off + d*C_off) x_v
                       pair mode    a = M_off x, b = C_off x of one vector
Iterations (G, T):     pass iteration        G decides, then streams; T = tail of the pass
                       build iteration       G decides "no pass"; T forms gradF, F, first window
                       transition iteration  workgroup (0,0) of G sweeps; T idle
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PH_NORMALIZE, PH_RESCALE, PH_INIT, PH_TRIAL, PH_PENALTY, PH_BUILD = range(6)


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


def solve(Moff: np.ndarray, Coff: np.ndarray, u0: np.ndarray, P: Params, V: int, adaptive: bool = False) -> Result:
    """Moff, Coff: dense symmetric, zero diagonal. adaptive: the window IN USE (SolverState::weff, k_solver.hip.h) —
    a pass multiplies candidate 0 alone while line searches have been accepting their first trial (two in a row, or
    none has run yet), the decision walks exactly the candidates the pass multiplied."""
    m = len(u0)
    tables = np.zeros((V + 1, m, V))          # candidate tables of the pending set
    nrm, sx = np.ones(V), np.zeros(V)
    u, g = np.zeros(m), np.zeros(m)           # the current point slot
    slot_u, slot_g = np.zeros((V, m)), np.zeros((V, m))
    cab = [np.zeros(m), np.zeros(m)]          # (a, b) of the last pair pass / of candidate 0
    d = F = s = 0.0
    alpha = 1.0
    i_ = j_ = k_ = 0
    n_passes = n_trials = n_iters = 0
    sel = 0
    accepted = []
    phase = PH_RESCALE if P.rescale_u0 else PH_NORMALIZE
    tables[0][:, 0] = u0
    results = not P.rescale_u0                # PH_NORMALIZE consumes no pass: decide at once
    sums = None
    from_u = False
    weff, zero_run, redo = V, 2, False

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

    def penalty_sums(uu, a, b, ss):
        cbu = ss - b - uu
        idx = (cbu > P.eps) & (uu > P.eps)
        return float(idx.sum()), float(np.sum(np.abs((a[idx] + uu[idx]) / cbu[idx])))

    def finish():
        return Result(u, F, d, i_, n_passes, n_trials, n_iters, accepted)

    while True:
        n_iters += 1
        action = "pass"
        from_u = False
        if results:
            # ---- the decision at the head of G ------------------------------------------------
            action = "slow"
            if phase == PH_TRIAL:
                jstar = -1
                before = (alpha, k_, n_trials)
                for v in range(weff):
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
                if jstar < 0 and weff < V:
                    # the pass on candidate 0 alone guessed wrong: the same window again, whole (nothing of it is used)
                    alpha, k_, n_trials = before
                    zero_run, redo = 0, True
                    tables = tables_before      # (the device builds a pending window from its point slot, which no tail
                    action = "pass"             # overwrites: the slices path has no tables; this model keeps them)
                elif jstar < 0:
                    sel, nrm, sx = V, sums["none_nrm"], sums["none_sx"]
                    action = "pass"
                else:
                    zero_run = zero_run + 1 if k_ == 0 else 0
                    accepted.append(jstar)
                    deltau = np.sqrt(sums["du2"][jstar])
                    s = sx_pass[jstar]
                    F = Fnew
                    u, g = slot_u[jstar].copy(), slot_g[jstar].copy()
                    j_ += 1
                    if deltau < P.tol_u or abs(deltaF) < P.tol_F or j_ >= P.maxiniters:
                        if jstar == 0:          # (a, b) and the penalty sums came with candidate 0
                            cnt, rs = sums["pen"]
                            if cnt > 0:
                                d += rs / cnt
                                i_ += 1
                                if i_ >= P.maxoliters:
                                    return finish()
                                action = "build"
                            else:
                                return finish()
                        else:                   # pair pass straight on the accepted x
                            action, phase, from_u = "pass", PH_PENALTY, True
                    else:
                        alpha, k_ = 1.0, 0
                        sel, nrm, sx = jstar, sums["nrm"][jstar], sums["sx"][jstar]
                        action = "pass"
            elif phase == PH_BUILD:
                F = sums["F"][0]
                j_ = 0
                if P.maxiniters > 0:
                    alpha, k_, sel = 1.0, 0, 0
                    nrm, sx = sums["nrm"][0], sums["sx"][0]
                    phase = PH_TRIAL
                    action = "pass"
            if action == "slow":
                # ---- sweeps by workgroup (0,0) ----------------------------------------------
                if phase in (PH_NORMALIZE, PH_RESCALE):
                    u = (cab[0] + u0) if phase == PH_RESCALE else u0.copy()
                    u = u / np.sqrt(float(u @ u))
                    tables[0] = 0.0
                    tables[0][:, 0] = u
                    sel, nrm = 0, np.ones(V)
                    phase, results = PH_INIT, False      # a pass was prepared
                    continue
                if phase == PH_INIT:
                    s = float(u.sum())
                    cbu = s - cab[1] - u
                    idx = (cbu > P.eps) & (u > P.eps)
                    d = float(np.mean((cab[0][idx] + u[idx]) / cbu[idx])) if idx.any() else 0.0
                    i_ = 0
                    if i_ >= P.maxoliters:
                        return finish()
                    action = "build"
                else:                                     # PH_PENALTY, or an empty inner loop
                    cnt, rs = penalty_sums(u, cab[0], cab[1], s)
                    if cnt > 0:
                        d += rs / cnt
                        i_ += 1
                        if i_ >= P.maxoliters:
                            return finish()
                        action = "build"
                    else:
                        return finish()
        if action == "build":
            # ---- build iteration: T forms gradF, F and the first window ----------------------
            g = (1 + d) * u - d * s + cab[0] + cab[1] * d
            tab, n0, sx0 = build_window(u, g, 1.0)
            tables[0] = tab
            sums = {"F": np.array([float(u @ g)] + [0.0] * (V - 1)), "nrm": {0: n0}, "sx": {0: sx0}}
            phase, results = PH_BUILD, True
            continue
        # ---- pass iteration: G streams M, T = tail ------------------------------------------
        weff = 1 if (adaptive and not redo and k_ == 0 and zero_run >= 2 and phase == PH_TRIAL) else V
        redo = False
        n_passes += 1
        X = tables[sel]
        if phase != PH_TRIAL:                       # pair mode (nrm = 1)
            x = u if from_u else X[:, 0]
            cab = [Moff @ x, Coff @ x]
            results = True
            continue
        sums = {"F": np.zeros(V), "du2": np.zeros(V), "nrm": {}, "sx": {}}
        sx_pass = sx.copy()
        out_tables = {}
        W = Moff + d * Coff
        for v in range(weff):
            xi = X[:, v] / nrm[v]
            if v == 0:                              # a and b apart, the reference's expression
                an, bn = (Moff @ X[:, 0]) / nrm[0], (Coff @ X[:, 0]) / nrm[0]
                gn = (1 + d) * xi - d * sx[0] + an + bn * d
                cab = [an, bn]
                sums["pen"] = penalty_sums(xi, an, bn, sx[0])
            else:
                gn = (1 + d) * xi - d * sx[v] + (W @ X[:, v]) / nrm[v]
            slot_u[v], slot_g[v] = xi, gn
            sums["F"][v] = float(xi @ gn)
            du = xi - u
            sums["du2"][v] = float(du @ du)
            out_tables[v], sums["nrm"][v], sums["sx"][v] = build_window(xi, gn, 1.0)
        al = alpha
        for _ in range(V):
            al = al * P.beta
        out_tables[V], sums["none_nrm"], sums["none_sx"] = build_window(u, g, al)
        tables_before = tables.copy() if weff < V else None
        for k, t in out_tables.items():
            tables[k] = t
        results = True
