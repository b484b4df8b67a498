"""Structure -> graph (host, vectorised numpy).  SURVEY.md 8f row N2.

Mirrors the per-structure rule of /root/reference/matdeeplearn/process/process.py:284-305,385-388:
  distance matrix (minimum image) -> keep <= graph_max_neighbors nearest within graph_max_radius per
  ROW (threshold_sort, :540-559; ordinal rank, ties -> lower column) -> dense_to_sparse (row-major)
  -> add_self_loops(weight 0) appended after the real edges -> one-hot out-degree node feature.
The product then stores every graph's edges SORTED BY TARGET (stable), which is what the kernels'
CSR layout wants; the reference's edge order (by source, loops last) is kept available as
`ref_order` for parity tests.
"""
import json

import numpy as np


def reduce_cell(cell, pbc):
    """Pairwise (Gauss / Minkowski-style) reduction of the PERIODIC cell vectors: repeatedly subtract integer
    multiples of the shorter vector of a pair from the longer one until no vector gets shorter.  For a basis reduced
    this way the minimum image of a wrapped difference vector lies within the 27 neighbouring images — which is how
    ase.geometry.find_mic (behind get_all_distances(mic=True), process.py:258) treats skewed cells."""
    c = np.array(cell, dtype=np.float64)
    per = [k for k in range(3) if pbc[k]]
    for _ in range(64):
        changed = False
        for a in per:
            for b in per:
                if a == b:
                    continue
                nb = c[b] @ c[b]
                if nb < 1e-24:
                    continue
                k = np.rint((c[a] @ c[b]) / nb)
                if k != 0.0:
                    new = c[a] - k * c[b]
                    if new @ new < c[a] @ c[a] - 1e-12:
                        c[a] = new
                        changed = True
        if not changed:
            break
    return c


def distance_matrix(positions, cell=None, pbc=None):
    """All-pairs minimum-image distances, ase get_all_distances(mic=True) semantics (process.py:258): the periodic cell
    vectors are lattice-reduced, difference vectors are wrapped into the reduced cell, then the 27 neighbouring images
    are searched — exact for skewed / thin triclinic cells too (tests/test_host_logic.py compares with a wide brute
    force)."""
    p = np.asarray(positions, dtype=np.float64)
    d = p[None, :, :] - p[:, None, :]                       # d[i, j] = p_j - p_i
    if pbc is None or not np.any(pbc):
        return np.sqrt((d * d).sum(-1))
    pbc = [bool(b) for b in pbc]
    cell = reduce_cell(cell, pbc)
    # fractional coordinates are taken in a basis whose NON-periodic directions are orthogonal to the periodic ones
    # (whatever the file stores there, possibly nothing): wrapping the periodic fractions then brings the component of
    # d inside the periodic subspace into the reduced cell
    full = cell.copy()
    per = [cell[k] for k in range(3) if pbc[k]]
    free = [k for k in range(3) if not pbc[k]]
    if len(per) == 2:
        v = np.cross(per[0], per[1])
        full[free[0]] = v / np.sqrt(v @ v)
    elif len(per) == 1:
        u = per[0] / np.sqrt(per[0] @ per[0])
        e = np.eye(3)[int(np.argmin(np.abs(u)))]
        v1 = np.cross(u, e)
        v1 /= np.sqrt(v1 @ v1)
        full[free[0]], full[free[1]] = v1, np.cross(u, v1)
    frac = d @ np.linalg.inv(full)
    for k in range(3):
        if pbc[k]:
            frac[..., k] -= np.rint(frac[..., k])
    d = frac @ full
    rng = [(-1, 0, 1) if b else (0,) for b in pbc]
    best = None
    for a in rng[0]:
        for b in rng[1]:
            for c in rng[2]:
                v = d + (a * cell[0] + b * cell[1] + c * cell[2])
                r2 = (v * v).sum(-1)
                best = r2 if best is None else np.minimum(best, r2)
    return np.sqrt(best)


def threshold_sort(matrix, threshold, neighbors):
    """process.py:540-559 (adj=False): rank<=neighbors+1 (ordinal, ascending, stable) AND value<=threshold
    keep the value, else 0."""
    m = np.asarray(matrix, dtype=np.float64)
    order = np.argsort(m, axis=1, kind="stable")
    rank = np.empty_like(order)
    rows = np.arange(m.shape[0])[:, None]
    rank[rows, order] = np.arange(1, m.shape[1] + 1)[None, :]
    keep = (rank <= neighbors + 1) & (m <= threshold)
    return np.where(keep, m, 0.0)


def edges_from_trimmed(trimmed):
    """Reference edge order: row-major non-zeros (fp32 values) then one self loop per node, weight 0."""
    t = np.asarray(trimmed, dtype=np.float32)
    r, c = np.nonzero(t)
    n = t.shape[0]
    loop = np.arange(n)
    ei = np.stack([np.concatenate([r, loop]), np.concatenate([c, loop])]).astype(np.int64)
    ew = np.concatenate([t[r, c], np.zeros(n, dtype=np.float32)])
    return ei, ew


def one_hot_degree(edge_index, num_nodes, max_degree):
    """process.py:594-605 — one-hot OUT-degree (edge_index[0]), max_degree+1 classes."""
    deg = np.bincount(edge_index[0], minlength=num_nodes)
    out = np.zeros((num_nodes, max_degree + 1), dtype=np.float32)
    out[np.arange(num_nodes), deg] = 1.0
    return out


def atom_features(numbers, dictionary=None):
    """dictionary_default.json is a one-hot of Z over 100 slots (d[str(Z)][Z-1] == 1)."""
    z = np.asarray(numbers, dtype=np.int64)
    if dictionary is not None:
        return np.asarray([dictionary[str(int(v))] for v in z], dtype=np.float32)
    out = np.zeros((len(z), 100), dtype=np.float32)
    out[np.arange(len(z)), z - 1] = 1.0
    return out


def build_graph(positions, numbers, cell=None, pbc=None, radius=8.0, max_neighbors=12, dictionary=None):
    """Returns dict(x [n,114], edge_index [2,E] (reference order), edge_weight [E])."""
    dm = distance_matrix(positions, cell, pbc)
    ei, ew = edges_from_trimmed(threshold_sort(dm, radius, max_neighbors))
    x = np.concatenate([atom_features(numbers, dictionary), one_hot_degree(ei, len(numbers), max_neighbors + 1)], 1)
    return {"x": x, "edge_index": ei, "edge_weight": ew}


def sort_by_target(edge_index, *edge_arrays):
    """Stable sort of the edges by target node -> CSR order used by the kernels."""
    perm = np.argsort(edge_index[1], kind="stable")
    return (edge_index[:, perm],) + tuple(a[perm] for a in edge_arrays) + (perm,)


def read_ase_json(path):
    """Minimal reader of the ASE-json structure files used by the reference datasets."""
    with open(path) as f:
        rec = json.load(f)
    rec = rec[str(rec["ids"][0])] if "ids" in rec else rec["1"]

    def arr(node):
        node = node["array"] if "array" in node else node
        shape, dtype, flat = node["__ndarray__"]
        return np.array(flat, dtype=dtype).reshape(shape)

    return {"positions": arr(rec["positions"]), "numbers": arr(rec["numbers"]), "cell": arr(rec["cell"]),
            "pbc": arr(rec["pbc"])}
