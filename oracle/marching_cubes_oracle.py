"""TEST INFRASTRUCTURE -- not part of the product.  Only tests/ may import this module.

Float64 restatement of the marching-cubes step of the reference's mesh extraction:

    extract_geometry (lib/models/renderers/NeuS.py:31-40):  u = extract_fields(...);  vertices, triangles = mcubes.marching_cubes(u, threshold)
    vertices = vertices / (resolution - 1) * (bound_max - bound_min) + bound_min                                  (NeuS.py:36-39)

`mcubes` is PyMCubes 0.1.4 (requirements.txt:13), a third-party C++ extension that is neither installed nor vendored in the reference
checkout, and the reference holds no mesh fixture.  What is restated here is therefore the PUBLISHED algorithm it implements
(W. E. Lorensen, H. E. Cline, "Marching Cubes: A High Resolution 3D Surface Construction Algorithm", SIGGRAPH 1987):
  * a lattice cell is classified by which of its 8 corners lie inside the surface (here: u > threshold, the `-sdf` convention of NeuS.py:416);
  * every lattice edge whose two end points are classified differently carries ONE vertex, at the linear interpolation of the level
    (paper, section 4, step 5) -- shared by the (up to four) cells around that edge;
  * inside a cell the vertices are joined into closed polygons whose sides lie in the cell faces, and the polygons are cut into triangles.
The 256-entry triangle table of the paper (and of PyMCubes) is a tabulation of the third step.  This module does NOT use a table and shares no
code or data with tools/gen_mc_table.py (which builds the device table): it contours every cell FACE on the fly from the corner
classification and follows the segments around the cell -- an independent construction of the same polygons, in float64.

Parity pinned / unpinned:
  * vertex positions, which lattice edges carry a vertex, and the polygons of every cell WITHOUT an ambiguous face (a face whose two
    diagonal corner pairs are classified differently) are determined by the published algorithm alone: the device mesh must agree exactly;
  * on an ambiguous face the paper's table (complement symmetry) can leave holes, and implementations differ; this oracle and the device
    table both separate the INSIDE corners of such a face (the face's two segments each cut off one inside corner).  PyMCubes' own table
    is not available here, so the polygons of cells WITH an ambiguous face are "parity unpinned" with respect to PyMCubes;
  * how a polygon with more than three sides is cut into triangles is a free choice that changes neither the polygon boundary nor the
    topology; comparisons are therefore made on directed polygon boundaries per cell (cell_boundaries), and on area / volume with a tolerance.
"""
import numpy as np

# corner c of a cell at (x, y, z): offset ((c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1)
_CORNER = np.array([[(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)
# the six faces as corner cycles (any consistent cyclic order of the four corners of the face)
_FACES = []
for axis in range(3):
    a, b = [i for i in range(3) if i != axis]
    for side in (0, 1):
        cyc = []
        for da, db in ((0, 0), (1, 0), (1, 1), (0, 1)):
            off = [0, 0, 0]
            off[axis], off[a], off[b] = side, da, db
            cyc.append(off[0] | (off[1] << 1) | (off[2] << 2))
        _FACES.append(tuple(cyc))


def _edge_key(cell, c0, c1):
    """Global key of the lattice edge between corners c0 and c1 of `cell`: (x, y, z, axis) of its lower end point."""
    p0 = np.asarray(cell) + _CORNER[c0]
    p1 = np.asarray(cell) + _CORNER[c1]
    lo = np.minimum(p0, p1)
    axis = int(np.nonzero(p0 != p1)[0][0])
    return (int(lo[0]), int(lo[1]), int(lo[2]), axis)


def marching_cubes(u, threshold=0.0, bound_min=None, bound_max=None):
    """u[x][y][z] -> (vertices [V, 3] float64, triangles [F, 3] int64, info).

    vertices are in lattice-index coordinates (like mcubes.marching_cubes) unless bound_min / bound_max are given, in which case the affine
    map of NeuS.py:36-39 is applied.  Triangles are oriented so that normals point from the inside (u > threshold) to the outside.
    info: {"edge_of_vertex": [(x, y, z, axis)], "ambiguous_cells": set of cells with an ambiguous face, "cell_of_triangle": [F] cells}."""
    u = np.asarray(u, dtype=np.float64)
    nx, ny, nz = u.shape
    inside = u > threshold
    # cells that the surface passes through
    cnt = np.zeros((nx - 1, ny - 1, nz - 1), dtype=np.int64)
    for c in range(8):
        ox, oy, oz = _CORNER[c]
        cnt += inside[ox:nx - 1 + ox, oy:ny - 1 + oy, oz:nz - 1 + oz]
    active = np.argwhere((cnt > 0) & (cnt < 8))
    vid, verts, edge_of_vertex = {}, [], []

    def vertex(key):
        if key not in vid:
            x, y, z, axis = key
            p0 = np.array([x, y, z], dtype=np.float64)
            q = [x, y, z]
            q[axis] += 1
            v0, v1 = u[x, y, z], u[q[0], q[1], q[2]]
            t = (threshold - v0) / (v1 - v0)              # linear interpolation of the level along the edge (paper, step 5)
            p = p0.copy()
            p[axis] += t
            vid[key] = len(verts)
            verts.append(p)
            edge_of_vertex.append(key)
        return vid[key]

    tris, cell_of_tri, ambiguous = [], [], set()
    for cell in active:
        cx, cy, cz = (int(t) for t in cell)
        ins = [bool(inside[cx + o[0], cy + o[1], cz + o[2]]) for o in _CORNER]
        # ---- contour every face: DIRECTED segments between crossing face edges.  Orientation rule (normals point from the inside to the
        # outside, polygons counter-clockwise seen from outside): walking along a segment in a face whose outward normal is f, the inside
        # corners of that face lie on the side d x f.
        nxt = {}           # edge key -> next edge key around its polygon
        centre = np.asarray(cell, dtype=np.float64) + 0.5
        for cyc in _FACES:
            corners = np.array([np.asarray(cell) + _CORNER[c] for c in cyc], dtype=np.float64)
            f = corners.mean(0) - centre                   # outward normal of the face (length 0.5)
            cross = []     # (index i of the face edge cyc[i] -> cyc[i+1], key)
            for i in range(4):
                a, b = cyc[i], cyc[(i + 1) % 4]
                if ins[a] != ins[b]:
                    cross.append((i, _edge_key(cell, a, b)))

            def segment(k0, k1, inside_corner):
                p0, p1 = verts[vertex(k0)], verts[vertex(k1)]
                side = float(np.dot(np.cross(p1 - p0, f), inside_corner - 0.5 * (p0 + p1)))
                assert side != 0.0
                if side < 0:
                    k0, k1 = k1, k0
                assert k0 not in nxt
                nxt[k0] = k1
            if len(cross) == 2:
                (_, k0), (_, k1) = cross
                segment(k0, k1, corners[[i for i in range(4) if ins[cyc[i]]][0]])
            elif len(cross) == 4:
                # ambiguous face: inside corners on one diagonal.  Each inside corner is cut off by a segment joining its two face edges.
                ambiguous.add((cx, cy, cz))
                keys = dict(cross)
                for i in range(4):
                    if ins[cyc[i]]:
                        segment(keys[(i - 1) % 4], keys[i], corners[i])          # the face edges ending in corner cyc[i]
        # ---- follow the segments around the cell: closed, oriented polygons
        seen = set()
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, cur = [], start
            while cur not in seen:
                seen.add(cur)
                loop.append(cur)
                cur = nxt[cur]
            assert cur == start and len(loop) >= 3, "the segments of a cell close up into polygons"
            ids = [vertex(k) for k in loop]
            for i in range(1, len(ids) - 1):      # fan
                tris.append((ids[0], ids[i], ids[i + 1]))
                cell_of_tri.append((cx, cy, cz))
    v = np.array(verts, dtype=np.float64).reshape(-1, 3)
    if bound_min is not None:
        bmin, bmax = np.asarray(bound_min, dtype=np.float64), np.asarray(bound_max, dtype=np.float64)
        v = v / (np.array([nx, ny, nz], dtype=np.float64) - 1.0) * (bmax - bmin) + bmin
    t = np.array(tris, dtype=np.int64).reshape(-1, 3)
    return v, t, {"edge_of_vertex": edge_of_vertex, "ambiguous_cells": ambiguous, "cell_of_triangle": cell_of_tri}


def match_vertices(v, v_ref, tol):
    """Index of the reference vertex within `tol` (max-norm) of every vertex of `v`; asserts a one-to-one correspondence of the two sets."""
    from scipy.spatial import cKDTree
    v, v_ref = np.asarray(v, dtype=np.float64), np.asarray(v_ref, dtype=np.float64)
    assert len(v) == len(v_ref), (len(v), len(v_ref))
    if len(v) == 0:
        return np.zeros(0, dtype=np.int64)
    dist, idx = cKDTree(v_ref).query(v, k=1, p=np.inf)
    assert float(dist.max()) <= tol, float(dist.max())
    assert len(np.unique(idx)) == len(v_ref), "two vertices of the mesh map to the same reference vertex"
    return idx


def cell_boundaries(t, edge_keys, cells=None):
    """Directed boundary edges of the triangles of every cell: {cell: frozenset((key_a, key_b))}.  Two triangulations of the same oriented
    polygons have the same boundaries.  cells: the cell of every triangle if the caller knows it (this oracle's info["cell_of_triangle"]);
    otherwise it is inferred: the lattice cell that contains the lattice edges of the triangle's three vertices -- unique unless all three
    edges lie in one lattice face (a fan triangle of a polygon that visits three edges of a face), where the triangle takes the cell of its
    neighbour in the output order (an extractor emits the triangles of a cell together)."""
    def cells_of(key):
        x, y, z, axis = key
        others = [i for i in range(3) if i != axis]
        out = set()
        for da in (0, -1):
            for db in (0, -1):
                c = [x, y, z]
                c[others[0]] += da
                c[others[1]] += db
                out.add(tuple(c))
        return out
    t = np.asarray(t)
    keys = [[edge_keys[int(i)] for i in tri] for tri in t]
    if cells is None:
        cand = [cells_of(k[0]) & cells_of(k[1]) & cells_of(k[2]) for k in keys]
        cells = [next(iter(c)) if len(c) == 1 else None for c in cand]
        for sweep in (range(len(t)), reversed(range(len(t)))):       # forward pass: previous neighbour; backward pass: next neighbour
            last = None
            for i in sweep:
                if cells[i] is None and last in cand[i]:
                    cells[i] = last
                last = cells[i] if cells[i] is not None else last
        assert all(c is not None and c in k for c, k in zip(cells, cand)), "triangle whose vertices do not lie on the edges of one lattice cell"
    per_cell = {}
    for ks, cell in zip(keys, cells):
        d = per_cell.setdefault(tuple(cell), {})
        for a, b in ((0, 1), (1, 2), (2, 0)):
            e = (ks[a], ks[b])
            d[e] = d.get(e, 0) + 1
    out = {}
    for cell, d in per_cell.items():
        bd = set()
        for (a, b), n in d.items():
            if d.get((b, a), 0) == 0:        # an interior diagonal of a polygon appears in both directions
                assert n == 1
                bd.add((a, b))
        out[cell] = frozenset(bd)
    return out


def area_volume(v, t):
    v = np.asarray(v, dtype=np.float64)
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    area = float(np.linalg.norm(np.cross(b - a, c - a), axis=1).sum() / 2.0)
    vol = float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
    return area, vol
