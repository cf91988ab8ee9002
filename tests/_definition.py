"""Independent evaluation of MTP site energies straight from the mathematical
definition (full Cartesian tensors + numpy einsum), sharing no code with the oracle or
the product: used to pin the oracle and the table generator.

M_{mu,nu}(i) = sum_j f_mu(|r_ij|, z_i, z_j) * n_ij (x) ... (x) n_ij   (nu factors),
f_mu = sum_k c[z_i][z_j][mu][k] * scaling * T_k(xi) (r - r_max)^2,  xi in [-1, 1]
B_alpha = full contraction of a product of such tensors; E_i = e0[z_i] + sum xi_alpha B_alpha.
"""
import string

import numpy as np


def chebyshev_T(k, xi):
    return np.cos(k * np.arccos(np.clip(xi, -1.0, 1.0)))


def radial_functions(pot, r, zi, zj):
    """f_mu(r) for all mu -> [Mu]."""
    Sp, R = pot.species_count, pot.radial_basis_size
    xi = (2 * r - (pot.min_dist + pot.max_dist)) / (pot.max_dist - pot.min_dist)
    # closed form, valid for xi in [-1,1]; outside use the polynomial definition
    if abs(xi) <= 1:
        Q = np.array([chebyshev_T(k, xi) for k in range(R)])
    else:
        Q = np.array([np.polynomial.chebyshev.Chebyshev.basis(k)(xi) for k in range(R)])
    Q = Q * pot.scaling * (r - pot.max_dist) ** 2
    return pot.radial_coeffs[zi * Sp + zj] @ Q


def site_tensors(pot, rvecs, zi, zjs, need):
    """Full tensors M[(mu,nu)] (shape (3,)*nu) for one atom."""
    out = {}
    for (mu, nu) in need:
        out[(mu, nu)] = np.zeros((3,) * nu)
    for rv, zj in zip(rvecs, zjs):
        r = np.linalg.norm(rv)
        n = rv / r
        f = radial_functions(pot, r, zi, zj)
        for (mu, nu) in need:
            t = np.array(f[mu])
            for _ in range(nu):
                t = np.multiply.outer(t, n)
            out[(mu, nu)] = out[(mu, nu)] + t
    return out


def graph_value(g, tensors):
    """Scalar basis function for one mtpgen.Graph."""
    val = 1.0
    for mu in g.scalars:
        val *= float(tensors[(mu, 0)])
    if g.types:
        letters = iter(string.ascii_letters)
        idx = [[] for _ in g.types]
        n = len(g.types)
        for a in range(n):
            for b in range(a + 1, n):
                for _ in range(g.mat[a][b]):
                    l = next(letters)
                    idx[a].append(l)
                    idx[b].append(l)
        spec = ",".join("".join(s) for s in idx) + "->"
        val *= float(np.einsum(spec, *[tensors[t] for t in g.types]))
    return val


def site_energies(pot, graphs, system, cutoff=None):
    """E_i for every owned atom of a driver.System; also returns basis values [n, S]."""
    rc = pot.max_dist if cutoff is None else cutoff
    need = set()
    for g in graphs:
        for mu in g.scalars:
            need.add((mu, 0))
        need.update(g.types)
    E = np.zeros(system.nlocal)
    Bv = np.zeros((system.nlocal, len(graphs)))
    for i in range(system.nlocal):
        js = system.neigh[system.first[i]:system.first[i + 1]]
        rv = system.x[js] - system.x[i]
        m = (rv ** 2).sum(1) <= rc * rc
        zi = system.types[i] - 1
        ten = site_tensors(pot, rv[m], zi, system.types[js[m]] - 1, need)
        Bv[i] = [graph_value(g, ten) for g in graphs]
        E[i] = pot.species_coeffs[zi] + Bv[i] @ pot.moment_coeffs
    return E, Bv


LEVEL8_GRAPHS_DOC = "scalars of SURVEY.md App. A: M00, M10, M00^2, M01.M01, M02:M02, M00 M10, M00^3, M00 (M01.M01), M00^4"


def level8_basis(ten):
    """The nine level-8 basis functions written out by hand (App. A order)."""
    M00, M10 = float(ten[(0, 0)]), float(ten[(1, 0)])
    M01, M02 = ten[(0, 1)], ten[(0, 2)]
    d11 = float(M01 @ M01)
    d22 = float((M02 * M02).sum())
    return np.array([M00, M10, M00 ** 2, d11, d22, M00 * M10, M00 ** 3, M00 * d11, M00 ** 4])
