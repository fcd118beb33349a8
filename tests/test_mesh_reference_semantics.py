"""What the reference's mesh call pins WITHOUT PyMCubes (the package is not in the image and no fixture of its output exists):

    vertices, triangles = mcubes.marching_cubes(u, threshold)                                        (NeuS.py:35)
    vertices = vertices / (resolution - 1.0) * (b_max_np - b_min_np)[None, :] + b_min_np[None, :]    (NeuS.py:36-39)

Marching cubes, whatever its triangle table, places ONE vertex on every lattice edge whose end points lie on different sides of the level
(in index space, at the linear interpolation point of the two lattice values); the reference then maps index space to the box by the affine
formula above; and the triangles of a level set that stays off the lattice boundary form a closed, consistently oriented 2-manifold.
Those three facts are triangulation-independent.  They are computed here directly from the lattice with numpy -- nothing shared with
oracle/marching_cubes_oracle.py, tools/gen_mc_table.py or the library -- and asserted for the device mesh (HIP under -m gpu, the CPU
emulation otherwise).  What stays UNPINNED without a PyMCubes vector: which way a cell with an ambiguous face is contoured, how a polygon
is cut into triangles, and the order of vertices / triangles.

The PLY files (NeuS_Trainer.py:287-307 writes them through trimesh) are checked by a byte-level parser written here from the PLY format
description -- header grammar, record sizes, little-endian scalars -- not by color-neus_amd/meshio.read_ply."""
import os
import struct

import numpy as np
import pytest
import torch

import _native as N


def _renderer(library, device):
    from oracle import colorneus_oracle as O     # weights only: marching cubes needs a renderer object, not its networks
    cfg = O.tiny_config()
    return N.make_renderer(cfg, O.init_params(cfg, seed=1, trained_like=True), library, device)


def _crossing_vertices(u, thr, bmin, bmax):
    """Index-space crossing points of every lattice edge (axis 0, 1, 2), mapped like NeuS.py:36-39.  float64 from the float32 lattice."""
    u = u.astype(np.float64)
    res = u.shape[0]
    inside = u > thr
    idx = np.stack(np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij"), -1).astype(np.float64)
    pts = []
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax], hi[ax] = slice(0, res - 1), slice(1, res)
        lo, hi = tuple(lo), tuple(hi)
        cross = inside[lo] != inside[hi]
        a, b = u[lo][cross], u[hi][cross]
        p = idx[lo][cross].copy()
        p[:, ax] += (thr - a) / (b - a)
        pts.append(p)
    p = np.concatenate(pts, 0)
    bmin, bmax = np.asarray(bmin, np.float64), np.asarray(bmax, np.float64)
    return p / (res - 1.0) * (bmax - bmin)[None, :] + bmin[None, :]


def _match_sets(a, b, tol):
    """Two point sets are the same multiset within tol: equal counts, every point of either set has a partner in the other, and around
    every point both sets hold the same number of points (several crossing points coincide where a lattice value equals the level)."""
    assert a.shape == b.shape, (a.shape, b.shape)
    if len(a) == 0:
        return
    from scipy.spatial import cKDTree
    ta, tb = cKDTree(a), cKDTree(b)
    da, _ = tb.query(a)
    db, _ = ta.query(b)
    assert da.max() <= tol and db.max() <= tol, (da.max(), db.max(), tol)
    na = ta.query_ball_point(b, r=4.0 * tol, return_length=True)
    nb = tb.query_ball_point(b, r=4.0 * tol, return_length=True)
    assert np.array_equal(na, nb), "two mesh vertices on one lattice edge (or one missing)"


def _closed_oriented_manifold(t, nv):
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], 0).astype(np.int64)
    key, rkey = e[:, 0] * (nv + 1) + e[:, 1], e[:, 1] * (nv + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key), "a directed edge used twice"
    assert np.array_equal(np.sort(key), np.sort(rkey)), "an edge without its opposite: the surface is not closed"
    assert (t[:, 0] != t[:, 1]).all() and (t[:, 1] != t[:, 2]).all() and (t[:, 0] != t[:, 2]).all(), "degenerate triangle"


def _fields():
    g = torch.Generator().manual_seed(11)
    lin = lambda r: torch.linspace(-1.0, 1.0, r)
    out = []
    # (name, lattice, threshold, bmin, bmax)
    x, y, z = torch.meshgrid(lin(33), lin(33), lin(33), indexing="ij")
    out.append(("sphere", 0.55 - torch.sqrt(x * x + y * y + z * z), 0.0, [-1.01] * 3, [1.01] * 3))
    out.append(("torus_shifted_level_anisotropic_box", 0.2 - torch.sqrt((torch.sqrt(x * x + y * y) - 0.55) ** 2 + z * z), 0.03125, [-1.0, -2.0, 0.5], [1.5, 2.0, 4.0]))
    u = torch.randn(22, 22, 22, generator=g)                       # rough field: every cell pattern, ambiguous faces included
    u[0], u[-1], u[:, 0], u[:, -1], u[:, :, 0], u[:, :, -1] = -2, -2, -2, -2, -2, -2
    # (levels that float32 holds exactly: the ABI takes the level as a float, PyMCubes as a double -- the reference itself only ever passes 0.0)
    out.append(("noise", u, 0.125, [0.0, 0.0, 0.0], [21.0, 10.5, 42.0]))
    u2 = u.clone()
    u2[5, 5, 5] = 0.125                                              # a lattice value EQUAL to the level: "inside" is u > threshold, so this corner is outside
    out.append(("noise_value_on_level", u2, 0.125, [0.0, 0.0, 0.0], [21.0, 21.0, 21.0]))
    return out


def _run(library, device):
    r = _renderer(library, device)
    for name, u, thr, bmin, bmax in _fields():
        u = u.float().contiguous()
        v, t = r.marching_cubes(u.to(device), bmin, bmax, thr)
        v, t = v.cpu().numpy().astype(np.float64), t.cpu().numpy()
        want = _crossing_vertices(u.numpy(), thr, bmin, bmax)
        box = float(np.max(np.asarray(bmax) - np.asarray(bmin)))
        _match_sets(v, want, 3e-6 * box)                             # float32 vertex coordinates against the float64 formula
        _closed_oriented_manifold(t, len(v))
        assert len(np.unique(t.reshape(-1))) == len(v), "unreferenced vertex"
        # outward orientation w.r.t. the u > threshold region: positive enclosed volume (divergence theorem)
        a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
        assert float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) > 0.0, name


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_mesh_vertices_are_the_reference_crossing_points_emu():
    _run(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_mesh_vertices_are_the_reference_crossing_points_hip():
    _run(None, "cuda:0")


# ------------------------------------------------------------------------------------------------------------------------------------
# PLY, parsed from the bytes
# ------------------------------------------------------------------------------------------------------------------------------------
_SCALAR = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
           "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f", "float64": "d"}


def _parse_ply(raw):
    """A PLY reader from the format description: header lines up to end_header, then for every element its records in order; scalar
    properties by their declared type, list properties as <count type><count x item type>.  Returns {element: {property: list}}."""
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").split("\n")
    assert lines[0] == "ply"
    fmt = [ln for ln in lines if ln.startswith("format ")]
    assert fmt == ["format binary_little_endian 1.0"], fmt
    elements, cur = [], None
    for ln in lines[1:]:
        tok = ln.split()
        if not tok or tok[0] in ("format", "comment", "obj_info", "end_header"):
            continue
        if tok[0] == "element":
            cur = (tok[1], int(tok[2]), [])
            elements.append(cur)
        elif tok[0] == "property":
            assert cur is not None
            if tok[1] == "list":
                cur[2].append((tok[4], ("list", _SCALAR[tok[2]], _SCALAR[tok[3]])))
            else:
                cur[2].append((tok[2], _SCALAR[tok[1]]))
        else:
            raise AssertionError("unknown header line: " + ln)
    off = end
    data = {}
    for name, count, props in elements:
        cols = {p: [] for p, _ in props}
        for _ in range(count):
            for p, ty in props:
                if isinstance(ty, tuple):
                    (n,) = struct.unpack_from("<" + ty[1], raw, off)
                    off += struct.calcsize("<" + ty[1])
                    vals = struct.unpack_from("<%d%s" % (n, ty[2]), raw, off)
                    off += struct.calcsize("<%d%s" % (n, ty[2]))
                    cols[p].append(vals)
                else:
                    (val,) = struct.unpack_from("<" + ty, raw, off)
                    off += struct.calcsize("<" + ty)
                    cols[p].append(val)
        data[name] = cols
    assert off == len(raw), "trailing or missing bytes: %d of %d consumed" % (off, len(raw))
    return data, [(n, c, [p for p, _ in ps]) for n, c, ps in elements]


def test_ply_bytes_against_an_independent_parser(tmp_path):
    """The two files of validate_mesh (NeuS_Trainer.py:287-307): `*_mesh.ply` (geometry) and `*_color.ply` (per-vertex colours through
    (colors * 255).astype(np.uint8), :292).  trimesh writes binary little-endian PLY with float x y z (+ uchar red green blue alpha) and
    `list uchar int vertex_indices` faces; a parser written from the format description must recover exactly what was passed in."""
    from color_neus_amd import meshio
    g = np.random.default_rng(5)
    v = g.standard_normal((37, 3)) * 3.0
    t = g.integers(0, 37, (61, 3))
    c = g.random((37, 3)) * 1.2 - 0.1                                # some values outside [0, 1]: clipped before the cast
    p1, p2 = str(tmp_path / "m_mesh.ply"), str(tmp_path / "m_color.ply")
    meshio.write_ply(p1, v, t)
    meshio.write_ply(p2, v, t, c)
    for path, colored in ((p1, False), (p2, True)):
        data, layout = _parse_ply(open(path, "rb").read())
        assert [n for n, _, _ in layout] == ["vertex", "face"] and layout[0][1] == 37 and layout[1][1] == 61
        assert layout[0][2] == ["x", "y", "z"] + (["red", "green", "blue", "alpha"] if colored else [])
        assert layout[1][2] == ["vertex_indices"]
        got_v = np.stack([data["vertex"][k] for k in ("x", "y", "z")], -1)
        assert np.array_equal(got_v.astype(np.float32), v.astype(np.float32))
        assert all(len(f) == 3 for f in data["face"]["vertex_indices"])
        assert np.array_equal(np.asarray(data["face"]["vertex_indices"]), t)
        if colored:
            want = (np.clip(c, 0.0, 1.0) * 255.0).astype(np.uint8)
            got_c = np.stack([data["vertex"][k] for k in ("red", "green", "blue")], -1)
            assert np.array_equal(got_c, want) and set(data["vertex"]["alpha"]) == {255}
    # an empty mesh is still a valid file
    meshio.write_ply(p1, np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64))
    data, layout = _parse_ply(open(p1, "rb").read())
    assert layout[0][1] == 0 and layout[1][1] == 0
