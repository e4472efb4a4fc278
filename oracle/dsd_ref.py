"""TEST INFRASTRUCTURE — CPU restatement of the reference's exact densest-subgraph rounding
(`Rounding::DSD`): Goldberg's flow-based algorithm as /root/reference/src/dsd.cpp implements it.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned by the reference's own known answers (test/dsd_test.cpp:14-43 and :47-80: the 20x20
weighted matrix whose densest subgraph is {3, 5, 12, 14, 15}, on the whole graph and restricted
to S = {0,1,3,5,7,12,14,15,19}), stored in tests/golden/reference_vectors.json.

What is followed, line by line:
  dsd.cpp:274-320  solve(A, S): fully connected graph on S (zero weights included), both
                   directions of every pair in the edge list, n = ALL nodes of A (not |S|)
  dsd.cpp:171-245  densest_subgraph: bisection on the density g in [0, m/2] while
                   n(n-1)(U-L) >= 1; network = source -> v (capacity m/2, integer division),
                   v -> sink (m/2 + 2g - degree(v)), u <-> v (w); the cut is the set reachable
                   from the source in the residual graph; a cut of {source} alone lowers U, anything
                   else raises L and is remembered
  dsd.cpp:39-160   the max-flow itself (Dinic: BFS levels, DFS with current-arc pointers, one
                   augmenting path per DFS call). Only its RESULT matters here — the minimal
                   source side of a minimum cut is unique — so this restatement uses its own
                   Dinic (iterative, adjacency lists) with the same strict `flow < cap` tests.
"""
from __future__ import annotations

from collections import deque

import numpy as np


class _Dinic:
    def __init__(self, nverts: int):
        self.n = nverts
        self.to: list[int] = []
        self.cap: list[float] = []
        self.flow: list[float] = []
        self.adj: list[list[int]] = [[] for _ in range(nverts)]

    def add(self, u: int, v: int, w: float):
        # dsd.cpp:39-51 new_edge: arc u->v with capacity w, and its reverse v->u carried as
        # (capacity w, flow w), i.e. residual 0
        self.adj[u].append(len(self.to))
        self.to.append(v); self.cap.append(w); self.flow.append(0.0)
        self.adj[v].append(len(self.to))
        self.to.append(u); self.cap.append(w); self.flow.append(w)

    def _bfs(self, s: int, t: int):
        dist = [-1] * self.n
        dist[s] = 0
        q = deque([s])
        while q:
            u = q.popleft()
            for e in self.adj[u]:
                v = self.to[e]
                if self.flow[e] < self.cap[e] and dist[v] < 0:
                    dist[v] = dist[u] + 1
                    q.append(v)
        return dist

    def maxflow(self, s: int, t: int) -> float:
        total = 0.0
        while True:
            dist = self._bfs(s, t)
            if dist[t] < 0:
                return total
            it = [0] * self.n
            while True:
                # one augmenting path along level edges (iterative DFS with current arcs)
                path, u = [], s
                while u != t:
                    advanced = False
                    while it[u] < len(self.adj[u]):
                        e = self.adj[u][it[u]]
                        v = self.to[e]
                        if self.flow[e] < self.cap[e] and dist[v] == dist[u] + 1:
                            path.append(e)
                            u = v
                            advanced = True
                            break
                        it[u] += 1
                    if not advanced:
                        if not path:
                            break
                        e = path.pop()          # dead end: retreat and skip that arc
                        u = self.to[e ^ 1]
                        it[u] += 1
                if u != t:
                    break
                df = min(self.cap[e] - self.flow[e] for e in path)
                for e in path:
                    self.flow[e] += df
                    self.flow[e ^ 1] -= df
                total += df

    def source_side(self, s: int):
        seen = [False] * self.n
        seen[s] = True
        st = [s]
        while st:
            u = st.pop()
            for e in self.adj[u]:
                v = self.to[e]
                if self.flow[e] < self.cap[e] and not seen[v]:
                    seen[v] = True
                    st.append(v)
        return seen


def densest_subgraph(A: np.ndarray, S=None):
    """dsd::solve(A, S) — A: dense symmetric weighted adjacency (the diagonal is ignored), S: optional
    list of nodes the search is restricted to. Returns the node list in ascending order."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    S = list(range(n)) if S is None or len(S) == 0 else [int(x) for x in S]
    k = len(S)
    m = k * k - k                      # dsd.cpp:286: directed pairs, both directions
    if k < 2:
        return []
    W = np.array([[A[min(i, j), max(i, j)] if i != j else 0.0 for j in S] for i in S])
    degree = W.sum(axis=1)             # dsd.cpp:190-195 (sum over the directed list = row sum)
    half = float(m // 2)               # dsd.cpp:25 `m / 2` on int64
    L, U = 0.0, half                   # dsd.cpp:200-201
    final = None
    src, dst = 0, k + 1
    while n * (n - 1) * (U - L) >= 1:  # dsd.cpp:219 (n = every node of A, isolated ones included)
        g = (U + L) / 2
        net = _Dinic(k + 2)
        for a in range(k):
            for b in range(k):
                if a != b:
                    net.add(a + 1, b + 1, float(W[a, b]))
        for a in range(k):
            net.add(src, a + 1, half)
        for a in range(k):
            net.add(a + 1, dst, half + 2 * g - float(degree[a]))
        net.maxflow(src, dst)
        cut = net.source_side(src)
        if sum(cut) == 1:              # dsd.cpp:229: only the source is reachable
            U = g
        else:
            L = g
            final = cut
    if final is None:
        return []
    return sorted(S[a] for a in range(k) if final[a + 1])
