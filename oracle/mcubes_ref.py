"""TEST INFRASTRUCTURE (CPU oracle) -- marching cubes on a dense scalar volume.

The reference turns the res^3 SDF volume of ``extract_fields`` into a triangle mesh with
``mcubes.marching_cubes(u, threshold)`` (grid_opt/utils/utils_sdf.py:89-101; PyMCubes, an
un-pinned third-party dependency that is absent from /root/reference and from this image:
``environment.yaml`` only lists ``pymcubes`` through pip).  Its published algorithm is
Lorensen & Cline's marching cubes: every cell of 2x2x2 samples is classified by the 8 signs
``u < iso``, a 256-entry table lists the triangles over the cell's 12 edges, and a triangle
corner on an edge sits at the linear zero crossing of ``u - iso`` along it; vertices shared by
neighbouring cells are emitted once, in index coordinates ``(x, y, z)`` of ``u[x, y, z]``.

This file restates that algorithm with numpy.  The case table is *derived* here (``case_table``)
instead of being typed in: on every cell face the crossing edges are joined by segments (a face
with four crossings joins the two edges around each inside corner, a rule that depends on the
face's signs only, so neighbouring cells agree and the surface is closed), the segments are
chained into loops and every loop is fan-triangulated.  The vertex SET of any valid marching
cubes is table-independent, which is what pins this oracle: tests/golden/mcubes.npz holds the
vertex sets, areas and volumes that scikit-image 0.18.3 (``method='lorensen'``, an independent
implementation found under /opt/conda in this image) produced for the volumes of
tools/make_mcubes_golden.py.  Parity with PyMCubes itself is UNPINNED (triangle order, fan
choice, orientation and its resolution of ambiguous faces cannot be checked here).

Orientation: raw triangles are counter-clockwise when seen from the ``u < iso`` side (normals
point towards lower values); ``save_mesh`` flips them like the reference does (``flip_face``).
"""
import numpy as np

# corner c = x + 2*y + 4*z;  edge e = 4*axis + (a + 2*b), (a, b) = the other two coordinates
# of the edge's corners in increasing axis order
CORNERS = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)


def _edge_id(axis, other):
    return 4 * axis + other[0] + 2 * other[1]


def _edges():
    """(12, 2) corner pairs (low end first) and (12, 3) midpoint coordinates."""
    ends = np.zeros((12, 2), dtype=np.int64)
    for axis in range(3):
        o = [a for a in range(3) if a != axis]
        for a in range(2):
            for b in range(2):
                p = [0, 0, 0]
                p[o[0]], p[o[1]] = a, b
                c0 = p[0] + 2 * p[1] + 4 * p[2]
                ends[_edge_id(axis, (a, b))] = (c0, c0 + (1 << axis))
    mid = (CORNERS[ends[:, 0]] + CORNERS[ends[:, 1]]) / 2.0
    return ends, mid


EDGE_ENDS, EDGE_MID = _edges()


def _edge_between(c0, c1):
    lo, hi = min(c0, c1), max(c0, c1)
    for e in range(12):
        if EDGE_ENDS[e, 0] == lo and EDGE_ENDS[e, 1] == hi:
            return e
    raise ValueError((c0, c1))


def _case_loops(case):
    """Directed loops of edge ids for one sign configuration (bit c set <=> corner c inside)."""
    inside = [(case >> c) & 1 == 1 for c in range(8)]
    nxt = {}
    for axis in range(3):
        for side in range(2):
            n = np.zeros(3)
            n[axis] = 1.0 if side else -1.0
            o = [a for a in range(3) if a != axis]
            # the face's corners in cyclic order
            cyc = []
            for a, b in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[o[0]], p[o[1]] = side, a, b
                cyc.append(p[0] + 2 * p[1] + 4 * p[2])
            fe = [_edge_between(cyc[i], cyc[(i + 1) % 4]) for i in range(4)]
            active = [i for i in range(4) if inside[cyc[i]] != inside[cyc[(i + 1) % 4]]]
            segs = []
            if len(active) == 2:
                ref = next(c for c in cyc if inside[c])
                segs.append((fe[active[0]], fe[active[1]], ref))
            elif len(active) == 4:
                for i in range(4):
                    if inside[cyc[i]]:
                        segs.append((fe[(i + 3) % 4], fe[i], cyc[i]))
            for e0, e1, ref in segs:
                p, q, c = EDGE_MID[e0], EDGE_MID[e1], CORNERS[ref].astype(float)
                # inside corner on the left of p -> q when the face is seen from outside the cell
                if np.dot(n, np.cross(q - p, c - p)) < 0:
                    e0, e1 = e1, e0
                assert e0 not in nxt
                nxt[e0] = e1
    loops, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, e = [], start
        while e not in seen:
            seen.add(e)
            loop.append(e)
            e = nxt[e]
        assert e == start
        loops.append(loop)
    return loops


def _on_common_face(e0, e1):
    p, q = EDGE_MID[e0], EDGE_MID[e1]
    return any(p[a] == q[a] and p[a] in (0.0, 1.0) for a in range(3))


def _fan_origin(loop):
    """Rotate the loop so that no fan diagonal lies in a cell face (such a diagonal could coincide with the
    neighbour's and leave an edge with four triangles)."""
    k = len(loop)
    for r in range(k):
        rot = loop[r:] + loop[:r]
        if not any(_on_common_face(rot[0], rot[i]) for i in range(2, k - 1)):
            return rot
    raise AssertionError(loop)


_TABLE = None


def case_table():
    """(256, 3*T) int8 edge ids per triangle corner (-1 padded) and (256,) triangle counts."""
    global _TABLE
    if _TABLE is None:
        rows = []
        for case in range(256):
            tris = []
            for loop in _case_loops(case):
                loop = _fan_origin(loop)
                for i in range(1, len(loop) - 1):
                    tris += [loop[0], loop[i], loop[i + 1]]
            rows.append(tris)
        width = max(len(r) for r in rows)
        tab = -np.ones((256, width), dtype=np.int8)
        for c, r in enumerate(rows):
            tab[c, :len(r)] = r
        _TABLE = (tab, np.array([len(r) // 3 for r in rows], dtype=np.int32))
    return _TABLE


def marching_cubes(u, iso=0.0):
    """vertices (V, 3) float32 in index coordinates of u[x, y, z], triangles (T, 3) int64.

    Vertex v of an edge between samples a (lower index) and b: a + (iso - u_a) / (u_b - u_a)
    along the edge's axis, in float32.  Vertices are unique and sorted by their edge key
    ``((x*ny + y)*3 + axis)*nz + z`` (a = (x, y, z)); triangles are ordered by cell (x-major) then table order."""
    u = np.asarray(u, dtype=np.float32)
    nx, ny, nz = u.shape
    tab, cnt = case_table()
    inside = u < np.float32(iso)
    case = np.zeros((nx - 1, ny - 1, nz - 1), dtype=np.int64)
    for c in range(8):
        dx, dy, dz = CORNERS[c]
        case |= inside[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:nz - 1 + dz].astype(np.int64) << c
    cells = np.argwhere(cnt[case] > 0)                      # x-major order
    ccase = case[cells[:, 0], cells[:, 1], cells[:, 2]]
    keys, pos = [], []
    for k in range(tab.shape[1]):
        e = tab[ccase, k].astype(np.int64)
        live = e >= 0
        ee = np.where(live, e, 0)
        a = cells + CORNERS[EDGE_ENDS[ee, 0]]
        axis = ee // 4
        keys.append(np.where(live, ((a[:, 0] * ny + a[:, 1]) * 3 + axis) * nz + a[:, 2], -1))
        b = a.copy()
        b[np.arange(len(a)), axis] += 1
        ua, ub = u[a[:, 0], a[:, 1], a[:, 2]], u[b[:, 0], b[:, 1], b[:, 2]]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (np.float32(iso) - ua) / (ub - ua)
        p = a.astype(np.float32)
        p[np.arange(len(a)), axis] += np.where(live, t, 0).astype(np.float32)
        pos.append(p)
    keys = np.stack(keys, 1).reshape(-1)                    # cell-major, table order
    pos = np.stack(pos, 1).reshape(-1, 3)
    live = keys >= 0
    keys, pos = keys[live], pos[live]
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    return pos[first].astype(np.float32), inv.reshape(-1, 3).astype(np.int64)


def mesh_area_volume(v, f):
    """Surface area and signed enclosed volume (divergence theorem) of a triangle mesh."""
    a, b, c = (np.asarray(v, dtype=np.float64)[f[:, i]] for i in range(3))
    n = np.cross(b - a, c - a)
    return 0.5 * np.linalg.norm(n, axis=1).sum(), (a * n).sum() / 6.0


def edge_manifold_counts(f):
    """For every undirected edge the number of triangles using it, and for every directed edge its count
    (a closed, consistently oriented surface has 2 and 1 everywhere)."""
    d = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    und = np.sort(d, axis=1)
    _, cu = np.unique(und, axis=0, return_counts=True)
    _, cd = np.unique(d, axis=0, return_counts=True)
    return cu, cd
