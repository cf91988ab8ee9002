"""Synthetic MLIP-3 potential generator (test/bench tooling, not on the product path).

The reference ships no potential files and none exist offline, so
parity tests and the bench need level-L `.mtp` / `.almtp` inputs made here.  This
module

  * enumerates every scalar contraction of moment tensors M_{mu,nu} whose level
    sum(2 + 4 mu + nu) is <= L (SURVEY.md App. C level rule),
  * lowers each contraction to binary `alpha_index_times` rows {a0, a1, mult, a3}
    (M[a3] += mult * M[a0] * M[a1]; the form the reference executes in file order,
    /root/reference/LAMMPS/ML-MTP/pair_mtp.cpp:196-201) with shared intermediates,
  * writes the MLIP-3 text grammar the reference parser accepts
    (pair_mtp.cpp:335-570) plus the optional `#MVS_v1.1` selection block with the raw
    fp64 active set and its inverse (pair_mtp_extrapolation.cpp:550-611).

A symmetric rank-nu tensor is stored by its monomial components (a, b, c),
a + b + c = nu, exactly like `alpha_index_basic` rows {mu, a, b, c}.  Contracting k
indices between symmetric S and U:

    R[alpha, beta] = sum_{|gamma| = k} multinomial(k; gamma) S[alpha+gamma] U[beta+gamma]

which is where the integer multiplicities of the `times` rows come from (the 2s of the
level-8 table in SURVEY.md App. A).
"""
from __future__ import annotations

import itertools
import struct
from dataclasses import dataclass, field
from math import factorial

import numpy as np

# --------------------------------------------------------------------------------------
# small combinatorics helpers


def monomials(nu):
    """Exponent triples (a, b, c) with a+b+c = nu, in the MLIP file order."""
    return [(a, b, nu - a - b) for a in range(nu, -1, -1) for b in range(nu - a, -1, -1)]


def multinomial(g):
    return factorial(sum(g)) // (factorial(g[0]) * factorial(g[1]) * factorial(g[2]))


def tadd(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


# --------------------------------------------------------------------------------------
# enumeration of scalar contraction graphs


def _tensor_types(L):
    return [(mu, nu) for mu in range(0, L // 4 + 1) for nu in range(0, L + 1)
            if 2 + 4 * mu + nu <= L]


def _sym_matrices(nus):
    """All symmetric non-negative integer matrices, zero diagonal, row sums = nus."""
    n = len(nus)
    mat = [[0] * n for _ in range(n)]
    rem = list(nus)
    out = []

    def rec(i, j):
        if i == n:
            out.append(tuple(tuple(r) for r in mat))
            return
        if j == n:
            if rem[i] == 0:
                rec(i + 1, i + 2)
            return
        # remaining capacity of row i must be coverable by columns j..n-1
        cap = sum(rem[k] for k in range(j, n))
        if rem[i] > cap:
            return
        for v in range(min(rem[i], rem[j]), -1, -1):
            mat[i][j] = mat[j][i] = v
            rem[i] -= v
            rem[j] -= v
            rec(i, j + 1)
            rem[i] += v
            rem[j] += v
        mat[i][j] = mat[j][i] = 0

    if n == 0:
        return [()]
    rec(0, 1)
    return out


def _canon(labels, mat):
    """Canonical form of a vertex-labelled multigraph: min over label-preserving
    permutations.  Returns (key, order) where order[p] = original vertex at canonical
    position p."""
    n = len(labels)
    idx = sorted(range(n), key=lambda i: labels[i])
    # groups of identical labels
    groups = []
    for k, grp in itertools.groupby(idx, key=lambda i: labels[i]):
        groups.append(list(grp))
    best = None
    best_order = None
    for perm in itertools.product(*[itertools.permutations(g) for g in groups]):
        order = [v for g in perm for v in g]
        key = tuple(mat[order[a]][order[b]] for a in range(n) for b in range(a + 1, n))
        if best is None or key < best:
            best = key
            best_order = order
    return (tuple(labels[i] for i in best_order), best), best_order


@dataclass(frozen=True)
class Graph:
    """One basis function: isolated nu=0 factors + a contracted part."""
    scalars: tuple          # sorted tuple of mu for the nu=0 tensors
    types: tuple            # (mu, nu) per contracted vertex, canonical order
    mat: tuple              # contraction counts, canonical order
    level: int

    @property
    def ntensors(self):
        return len(self.scalars) + len(self.types)


def enumerate_graphs(L):
    types = _tensor_types(L)
    scal = [t for t in types if t[1] == 0]
    tens = [t for t in types if t[1] > 0]
    graphs = {}

    def lev(t):
        return 2 + 4 * t[0] + t[1]

    # multisets of nu>0 tensors
    tens_sets = []

    def rec_t(start, cur, level):
        tens_sets.append((tuple(cur), level))
        for k in range(start, len(tens)):
            l2 = level + lev(tens[k])
            if l2 <= L:
                cur.append(tens[k])
                rec_t(k, cur, l2)
                cur.pop()

    rec_t(0, [], 0)

    scal_sets = []

    def rec_s(start, cur, level):
        scal_sets.append((tuple(cur), level))
        for k in range(start, len(scal)):
            l2 = level + lev(scal[k])
            if l2 <= L:
                cur.append(scal[k])
                rec_s(k, cur, l2)
                cur.pop()

    rec_s(0, [], 0)

    contracted = {}
    for ts, lv in tens_sets:
        if not ts:
            contracted[((), ())] = ((), (), 0)
            continue
        nus = [t[1] for t in ts]
        if sum(nus) % 2 or max(nus) > sum(nus) - max(nus):
            continue
        for m in _sym_matrices(nus):
            key, order = _canon(list(ts), m)
            if key not in contracted:
                cm = tuple(tuple(m[order[a]][order[b]] for b in range(len(ts)))
                           for a in range(len(ts)))
                contracted[key] = (key[0], cm, lv)
    for (ctypes_, cm, lv) in contracted.values():
        for ss, ls in scal_sets:
            if lv + ls > L or (not ctypes_ and not ss):
                continue
            g = Graph(tuple(sorted(t[0] for t in ss)), ctypes_, cm, lv + ls)
            graphs[(g.scalars, g.types, g.mat)] = g
    out = sorted(graphs.values(), key=lambda g: (g.ntensors, g.level, g.scalars, g.types, g.mat))
    return out


# --------------------------------------------------------------------------------------
# lowering to alpha_index_times


@dataclass
class MTPTable:
    level: int
    basic: list = field(default_factory=list)       # [(mu, a, b, c)]
    times: list = field(default_factory=list)       # [(a0, a1, mult, a3)]
    mapping: list = field(default_factory=list)     # moment index per scalar
    nmoments: int = 0
    graphs: list = field(default_factory=list)
    radial_funcs: int = 0

    @property
    def sizes(self):
        P = 1 + max(b[1] + b[2] + b[3] for b in self.basic)
        return dict(level=self.level, B=len(self.basic), T=len(self.times), S=len(self.mapping),
                    A=self.nmoments, Mu=self.radial_funcs, P=P)


class _Lowerer:
    def __init__(self, L, graphs):
        self.L = L
        self.graphs = graphs
        need = set()
        for g in graphs:
            for mu in g.scalars:
                need.add((mu, 0))
            for t in g.types:
                need.add(t)
        self.basic = []
        self.basic_index = {}
        for (mu, nu) in sorted(need):
            for m in monomials(nu):
                self.basic_index[(mu, m)] = len(self.basic)
                self.basic.append((mu,) + m)
        self.nmom = len(self.basic)
        self.rows = {}          # (a0, a1, a3) -> mult, insertion ordered
        self.node_cache = {}    # canonical key -> dict(comp tuple -> moment index)
        self.prod_cache = {}    # sorted tuple of scalar moment ids -> moment index

    # -- nodes over a subset of vertices of a connected contracted graph ------------
    def _subkey(self, types, mat, U):
        """Canonical key of the partial contraction over vertex subset U (tuple)."""
        free = [types[v][1] - sum(mat[v][w] for w in U) for v in U]
        labels = [(types[v], free[i]) for i, v in enumerate(U)]
        sub = [[mat[v][w] for w in U] for v in U]
        key, order = _canon(labels, sub)
        verts = [U[i] for i in order]           # canonical vertex order (parent labels)
        return key, verts, {U[i]: free[i] for i in range(len(U))}

    def _ncomp(self, freemap):
        n = 1
        for f in freemap.values():
            n *= (f + 1) * (f + 2) // 2
        return n

    def _plan(self, types, mat, U, memo):
        """Cheapest binary contraction tree for subset U: returns (cost, split)."""
        if U in memo:
            return memo[U]
        key, verts, free = self._subkey(types, mat, U)
        if len(U) == 1 or key in self.node_cache:
            memo[U] = (0, None)
            return memo[U]
        best = None
        first = U[0]
        rest = U[1:]
        ncomp = self._ncomp(free)
        for r in range(0, len(rest)):
            for extra in itertools.combinations(rest, r):
                U1 = (first,) + extra
                U2 = tuple(v for v in U if v not in U1)
                terms = 1
                for i in U1:
                    for j in U2:
                        n = mat[i][j]
                        terms *= (n + 1) * (n + 2) // 2
                c = ncomp * terms + self._plan(types, mat, U1, memo)[0] + \
                    self._plan(types, mat, U2, memo)[0]
                if best is None or c < best[0]:
                    best = (c, (U1, U2))
        memo[U] = best
        return best

    def _comp_lookup(self, types, mat, U):
        """Build (or fetch) node(U); return function comp(exponent dict v->triple) -> id."""
        key, verts, free = self._subkey(types, mat, U)
        if len(U) == 1:
            v = U[0]
            mu = types[v][0]
            return lambda ex: self.basic_index[(mu, ex[v])]
        gverts = [v for v in verts if free[v] > 0]
        if key not in self.node_cache:
            self._build(types, mat, U, key, gverts, free)
        table = self.node_cache[key]
        return lambda ex: table[tuple(ex[v] for v in gverts)]

    def _build(self, types, mat, U, key, gverts, free):
        memo = getattr(self, "_memo")
        cost, split = self._plan(types, mat, U, memo)
        U1, U2 = split
        look1 = self._comp_lookup(types, mat, U1)
        look2 = self._comp_lookup(types, mat, U2)
        edges = [(i, j) for i in U1 for j in U2 if mat[i][j] > 0]
        gam_choices = [monomials(mat[i][j]) for (i, j) in edges]
        table = {}
        comp_axes = [monomials(free[v]) for v in gverts]
        zero = (0, 0, 0)
        for comp in itertools.product(*comp_axes):
            out_id = self.nmom
            self.nmom += 1
            table[comp] = out_id
            base = {v: zero for v in U}
            for v, e in zip(gverts, comp):
                base[v] = e
            for gams in itertools.product(*gam_choices):
                ex = dict(base)
                mult = 1
                for (i, j), g in zip(edges, gams):
                    ex[i] = tadd(ex[i], g)
                    ex[j] = tadd(ex[j], g)
                    mult *= multinomial(g)
                a0 = look1(ex)
                a1 = look2(ex)
                if a0 > a1:
                    a0, a1 = a1, a0
                k = (a0, a1, out_id)
                self.rows[k] = self.rows.get(k, 0) + mult
        self.node_cache[key] = table

    # -- connected components ---------------------------------------------------------
    @staticmethod
    def _components(n, mat):
        seen = [False] * n
        comps = []
        for s in range(n):
            if seen[s]:
                continue
            stack = [s]
            seen[s] = True
            comp = []
            while stack:
                v = stack.pop()
                comp.append(v)
                for w in range(n):
                    if mat[v][w] and not seen[w]:
                        seen[w] = True
                        stack.append(w)
            comps.append(tuple(sorted(comp)))
        return comps

    def _scalar_of_component(self, types, mat, U):
        self._memo = {}
        return self._comp_lookup(types, mat, U)({})

    def _product(self, ids):
        """Scalar = product of scalar factors.  Every sub-multiset of the factors is itself a
        basis function made earlier (fewer tensors, lower level), so any split costs exactly
        one row; the split that halves the factor list keeps the dependency depth of the
        times table logarithmic instead of linear in the number of factors."""
        ids = tuple(sorted(ids))
        if len(ids) == 1:
            return ids[0]
        if ids in self.prod_cache:
            return self.prod_cache[ids]
        h = len(ids) // 2
        best = None
        # prefer a balanced split whose halves already exist; fall back to plain halves
        for left in itertools.combinations(range(len(ids)), h):
            a = tuple(ids[i] for i in left)
            b = tuple(ids[i] for i in range(len(ids)) if i not in left)
            have = (len(a) == 1 or a in self.prod_cache) + (len(b) == 1 or b in self.prod_cache)
            if best is None or have > best[0]:
                best = (have, a, b)
            if have == 2:
                break
        _, a, b = best
        ra, rb = self._product(a), self._product(b)
        out = self.nmom
        self.nmom += 1
        a0, a1 = sorted((ra, rb))
        self.rows[(a0, a1, out)] = self.rows.get((a0, a1, out), 0) + 1
        self.prod_cache[ids] = out
        return out

    def run(self):
        mapping = []
        for g in self.graphs:
            factors = [self.basic_index[(mu, (0, 0, 0))] for mu in g.scalars]
            if g.types:
                for U in self._components(len(g.types), g.mat):
                    factors.append(self._scalar_of_component(g.types, g.mat, U))
            mapping.append(self._product(factors))
        times = [(a0, a1, m, a3) for (a0, a1, a3), m in self.rows.items()]
        return self.basic, times, mapping, self.nmom


def build_table(L, max_scalars=None):
    """Complete level-L table (optionally truncated to the first `max_scalars` basis
    functions in (tensor count, level) order; rows that only feed dropped scalars are
    pruned and moments renumbered)."""
    graphs = enumerate_graphs(L)
    if max_scalars is not None:
        graphs = graphs[:max_scalars]
    low = _Lowerer(L, graphs)
    basic, times, mapping, nmom = low.run()
    # topological sanity: operands are produced before use
    produced = set(range(len(basic)))
    done_targets = set()
    for (a0, a1, m, a3) in times:
        assert a0 in produced and a1 in produced, "times rows not topologically ordered"
        done_targets.add(a3)
        # a3 becomes usable only after all of its rows; rows of one node are contiguous
        nxt = a3
        produced.add(nxt)
    t = MTPTable(level=L, basic=basic, times=times, mapping=mapping, nmoments=nmom, graphs=graphs)
    t.radial_funcs = 1 + max(b[0] for b in basic)
    return t


LEVEL8_KNOWN_ANSWER = dict(
    # the MLIP level-8 template (hand-derived there, used here as the
    # known-answer for the generator and as the config-1 table).
    basic=[(0, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (0, 2, 0, 0), (0, 1, 1, 0),
           (0, 1, 0, 1), (0, 0, 2, 0), (0, 0, 1, 1), (0, 0, 0, 2), (1, 0, 0, 0)],
    times=[(0, 0, 1, 11), (1, 1, 1, 12), (2, 2, 1, 12), (3, 3, 1, 12), (4, 4, 1, 13), (5, 5, 2, 13),
           (6, 6, 2, 13), (7, 7, 1, 13), (8, 8, 2, 13), (9, 9, 1, 13), (0, 10, 1, 14), (0, 11, 1, 15),
           (0, 12, 1, 16), (0, 15, 1, 17)],
    mapping=[0, 10, 11, 12, 13, 14, 15, 16, 17],
    nmoments=18,
)


def level8_template():
    k = LEVEL8_KNOWN_ANSWER
    t = MTPTable(level=8, basic=list(k["basic"]), times=list(k["times"]), mapping=list(k["mapping"]),
                 nmoments=k["nmoments"])
    t.radial_funcs = 2
    return t


# --------------------------------------------------------------------------------------
# potential (table + coefficients) and the MLIP-3 writer


@dataclass
class Potential:
    table: MTPTable
    species_count: int
    min_dist: float
    max_dist: float
    radial_basis_size: int
    scaling: float
    radial_coeffs: np.ndarray      # [Sp*Sp, Mu, R]  index (t1*Sp+t2)
    species_coeffs: np.ndarray     # [Sp]
    moment_coeffs: np.ndarray      # [S]
    name: str = "synthetic"
    # optional selection state
    mvs_mode: str | None = None    # "nbh" | "cfg" | None
    active_set: np.ndarray | None = None
    inverse_active_set: np.ndarray | None = None

    @property
    def coeff_count(self):
        t = self.table
        return self.species_count ** 2 * t.radial_funcs * self.radial_basis_size + \
            self.species_count + len(t.mapping)


def random_potential(table, species_count=1, seed=12345, min_dist=2.0, max_dist=5.0,
                     radial_basis_size=8, scaling=1.0, damp=0.25):
    """SURVEY.md section 8(d) synthetic values.  `damp` scales the coefficient of a basis
    function made of n tensors by damp**(n-1) so forces stay O(1-10) eV/A."""
    rng = np.random.default_rng(seed)
    Sp, Mu, R = species_count, table.radial_funcs, radial_basis_size
    rc = rng.uniform(-0.1, 0.1, size=(Sp * Sp, Mu, R))
    sc = rng.uniform(-1.0, 1.0, size=Sp)
    mc = rng.uniform(-0.5, 0.5, size=len(table.mapping))
    if table.graphs:
        nt = np.array([g.ntensors for g in table.graphs])
    else:   # level-8 template order (App. A): tensor counts of its 9 scalars
        nt = np.array([1, 1, 2, 2, 2, 2, 3, 3, 4])
    mc = mc * damp ** (nt - 1)
    return Potential(table, Sp, min_dist, max_dist, R, scaling, rc, sc, mc)


def add_selection_state(pot, mode="nbh", seed=99):
    """Attach a synthetic, well-conditioned active set (A = 2 I + 0.05 U(-1,1)) and its
    inverse."""
    rng = np.random.default_rng(seed)
    C = pot.coeff_count
    A = 2.0 * np.eye(C) + 0.05 * rng.uniform(-1, 1, size=(C, C))
    pot.active_set = A
    pot.inverse_active_set = np.linalg.inv(A)
    pot.mvs_mode = mode
    return pot


def _fmt(v):
    return "%.15e" % v


def write_mtp(pot, path):
    """MLIP-3 text (+ optional binary MVS tail).  Line lengths respect the reference
    reader's buffer resizing (<=20 chars per basic entry, <=32 per
    times entry, later lines < T*32+20)."""
    t = pot.table
    Sp, Mu, R = pot.species_count, t.radial_funcs, pot.radial_basis_size
    lines = ["MTP", "version = 1.1.0", "potential_name = %s" % pot.name]
    if pot.scaling != 1.0:
        lines.append("scaling = %s" % _fmt(pot.scaling))
    lines += ["species_count = %d" % Sp, "potential_tag = ",
              "radial_basis_type = RBChebyshev",
              "\tmin_dist = %s" % _fmt(pot.min_dist),
              "\tmax_dist = %s" % _fmt(pot.max_dist),
              "\tradial_basis_size = %d" % R,
              "\tradial_funcs_count = %d" % Mu,
              "\tradial_coeffs"]
    for t1 in range(Sp):
        for t2 in range(Sp):
            lines.append("\t\t%d-%d" % (t1, t2))
            for mu in range(Mu):
                lines.append("\t\t\t{" + ", ".join(_fmt(v) for v in pot.radial_coeffs[t1 * Sp + t2, mu]) + "}")
    lines.append("alpha_moments_count = %d" % t.nmoments)
    lines.append("alpha_index_basic_count = %d" % len(t.basic))
    lines.append("alpha_index_basic = {" + ", ".join("{%d, %d, %d, %d}" % b for b in t.basic) + "}")
    lines.append("alpha_index_times_count = %d" % len(t.times))
    lines.append("alpha_index_times = {" + ", ".join("{%d, %d, %d, %d}" % r for r in t.times) + "}")
    lines.append("alpha_scalar_moments = %d" % len(t.mapping))
    lines.append("alpha_moment_mapping = {" + ", ".join("%d" % m for m in t.mapping) + "}")
    lines.append("species_coeffs = {" + ", ".join(_fmt(v) for v in pot.species_coeffs) + "}")
    lines.append("moment_coeffs = {" + ", ".join(_fmt(v) for v in pot.moment_coeffs) + "}")
    for b in t.basic:
        assert len("{%d, %d, %d, %d}, " % b) <= 20
    for r in t.times:
        assert len("{%d, %d, %d, %d}, " % r) <= 32
    lim = max(1024, len(t.times) * 32 + 20)
    for ln in lines[-3:]:
        assert len(ln) + 2 < lim, "line would overflow the reference reader's buffer"
    data = ("\n".join(lines) + "\n").encode()
    if pot.mvs_mode is not None:
        cfg = pot.mvs_mode == "cfg"
        tail = ["#MVS_v1.1", "energy_weight = %d" % (1 if cfg else 0), "force_weight = 0",
                "stress_weight = 0", "site_en_weight = %d" % (0 if cfg else 1),
                "weight_scaling = 1"]
        data += ("\n".join(tail) + "\n").encode() + b"#"
        data += np.ascontiguousarray(pot.active_set, dtype="<f8").tobytes()
        data += np.ascontiguousarray(pot.inverse_active_set, dtype="<f8").tobytes()
    with open(path, "wb") as fh:
        fh.write(data)
    return path


# --------------------------------------------------------------------------------------
# synthetic structures


def bcc_lattice(nx, ny, nz, a=3.165, jitter=0.05, seed=777):
    """BCC conventional cells; returns (positions [N,3], box lengths [3])."""
    rng = np.random.default_rng(seed)
    cells = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"),
                     axis=-1).reshape(-1, 3).astype(np.float64)
    pos = np.concatenate([cells, cells + 0.5], axis=0) * a
    # interleave the two sublattices so neighbours in space are near in memory
    order = np.argsort(np.concatenate([np.arange(len(cells)) * 2, np.arange(len(cells)) * 2 + 1]))
    pos = pos[order]
    pos += rng.uniform(-jitter, jitter, size=pos.shape)
    box = np.array([nx, ny, nz], dtype=np.float64) * a
    return pos, box


if __name__ == "__main__":
    import argparse
    import json
    import time

    ap = argparse.ArgumentParser(description="write a synthetic level-L MLIP-3 potential")
    ap.add_argument("--level", type=int, required=True)
    ap.add_argument("--species", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--mvs", choices=["nbh", "cfg"], default=None)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--max-scalars", type=int, default=None)
    args = ap.parse_args()
    t0 = time.time()
    tab = build_table(args.level, args.max_scalars)
    pot = random_potential(tab, args.species, args.seed)
    if args.mvs:
        add_selection_state(pot, args.mvs)
    write_mtp(pot, args.out)
    print(json.dumps(dict(tab.sizes, C=pot.coeff_count, seconds=round(time.time() - t0, 2))))
