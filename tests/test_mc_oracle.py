"""Marching cubes (replaces mcubes.marching_cubes of NeuS.py:35): the device extractor against the independent float64
restatement of the published algorithm in oracle/marching_cubes_oracle.py (no table, no code shared with tools/gen_mc_table.py).

What is compared (see the oracle's header for what the published algorithm pins and what it leaves open):
  * the vertex SET: same lattice edges, positions within 2e-6 of the bounding-box size;
  * the oriented polygons of EVERY cell (directed boundary of the cell's triangles): identical -- hence identical topology, orientation and
    triangle count; on cells with an ambiguous face this holds because both sides use the same stated policy (inside corners separated);
  * area and enclosed volume: equal up to the free choice of how a polygon is cut into triangles (1e-3 on smooth surfaces)."""
import os

import numpy as np
import pytest
import torch

import _native as N
from oracle import marching_cubes_oracle as M


def _renderer(library, device):
    from oracle import colorneus_oracle as O
    cfg = O.tiny_config()
    return N.make_renderer(cfg, O.init_params(cfg, seed=1, trained_like=True), library, device)


def _lattice(res, f):
    lin = torch.linspace(-1.0, 1.0, res)
    x, y, z = torch.meshgrid(lin, lin, lin, indexing="ij")
    return f(x, y, z).float().contiguous()


def _cases():
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(18, 18, 18, generator=g)
    noise[0], noise[-1], noise[:, 0], noise[:, -1], noise[:, :, 0], noise[:, :, -1] = -1, -1, -1, -1, -1, -1
    return {
        "sphere": (_lattice(28, lambda x, y, z: 0.6 - torch.sqrt(x * x + y * y + z * z)), [-1, -1, -1], [1, 1, 1], 0.0, 1e-3),
        "torus": (_lattice(36, lambda x, y, z: 0.2 - torch.sqrt((torch.sqrt(x * x + y * y) - 0.55) ** 2 + z * z)), [-1, -1, -1], [1, 1, 1], 0.0, 1e-3),
        "two_blobs": (_lattice(32, lambda x, y, z: torch.maximum(0.3 - torch.sqrt((x - 0.45) ** 2 + y * y + z * z),
                                                                  0.25 - torch.sqrt((x + 0.5) ** 2 + y * y + z * z))), [-1, -1, -1], [1, 1, 1], 0.0, 1e-3),
        "shifted_level_box": (_lattice(20, lambda x, y, z: 0.5 * x + 0.3 * torch.sin(3 * y) * torch.cos(2 * z)), [0, 0, 0], [2, 4, 8], 0.1, 1e-3),
        "noise": (noise, [0, 0, 0], [17, 17, 17], 0.0, 3e-2),     # every cell pattern, ~1/3 of the active cells with an ambiguous face
    }


def _compare(r, device):
    for name, (u, bmin, bmax, thr, tol_av) in _cases().items():
        v, t, info = M.marching_cubes(u.numpy(), thr, bmin, bmax)
        hv, ht = r.marching_cubes(u.to(device), bmin, bmax, thr)
        hv, ht = hv.cpu().numpy().astype(np.float64), ht.cpu().numpy().astype(np.int64)
        assert len(ht) == len(t), (name, len(ht), len(t))                         # triangle count
        size = max(abs(b - a) for a, b in zip(bmin, bmax))
        idx = M.match_vertices(hv, v, 2e-6 * size)                                # vertex set (one-to-one, positions)
        keys = [info["edge_of_vertex"][i] for i in idx]
        want = M.cell_boundaries(t, info["edge_of_vertex"], info["cell_of_triangle"])
        got = M.cell_boundaries(ht, keys)
        diff = [c for c in set(want) | set(got) if want.get(c) != got.get(c)]
        assert not diff, (name, len(diff), diff[:3])                              # oriented polygons of every cell
        if name == "noise":
            assert len(info["ambiguous_cells"]) > 200                             # the ambiguous-face policy is really exercised
        else:
            assert not info["ambiguous_cells"], name
        (a0, v0), (a1, v1) = M.area_volume(v, t), M.area_volume(hv, ht)
        assert abs(a1 - a0) <= tol_av * abs(a0) and abs(v1 - v0) <= tol_av * max(abs(v0), 1e-12), (name, a0, a1, v0, v1)


# ---------------------------------------------------------------- the oracle against known answers (CPU)
def test_oracle_single_corner_and_plane():
    u = -np.ones((2, 2, 2))
    u[0, 0, 0] = 3.0                       # one inside corner: one triangle cutting it off at t = 3 / 4 along each edge
    v, t, info = M.marching_cubes(u, 0.0)
    assert len(t) == 1 and sorted(map(tuple, np.round(v, 12).tolist())) == [(0.0, 0.0, 0.75), (0.0, 0.75, 0.0), (0.75, 0.0, 0.0)]
    a, b, c = v[t[0]]
    assert np.dot(np.cross(b - a, c - a), np.ones(3)) > 0                       # normal points away from the inside corner
    # complement: same vertices, opposite orientation
    v2, t2, _ = M.marching_cubes(-u, 0.0)
    a, b, c = v2[t2[0]]
    assert np.dot(np.cross(b - a, c - a), np.ones(3)) < 0
    # a plane x = 0.25 through an 8^3 lattice on [0, 1]^3 scaled to a box: (n - 1)^2 quads -> 2 (n - 1)^2 triangles, exact area
    lin = np.linspace(0.0, 1.0, 8)
    x = np.broadcast_to(lin[:, None, None], (8, 8, 8))
    v, t, info = M.marching_cubes(0.25 - x, 0.0, [0, 0, 0], [2, 3, 5])
    assert len(t) == 2 * 7 * 7 and np.allclose(v[:, 0], 0.5)
    area, _ = M.area_volume(v, t)
    assert abs(area - 15.0) < 1e-12
    assert not info["ambiguous_cells"]


def test_oracle_sphere_convergence_and_topology():
    errs = []
    for res in (17, 33):
        lin = np.linspace(-1.0, 1.0, res)
        x, y, z = np.meshgrid(lin, lin, lin, indexing="ij")
        v, t, _ = M.marching_cubes(0.6 - np.sqrt(x * x + y * y + z * z), 0.0, [-1, -1, -1], [1, 1, 1])
        e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], 0)
        fwd = set(map(tuple, e.tolist()))
        assert len(fwd) == len(e) and all((b, a) in fwd for a, b in fwd)          # closed, consistently oriented
        assert len(v) - len(e) // 2 + len(t) == 2                                # Euler characteristic of a sphere
        errs.append(abs(M.area_volume(v, t)[1] - 4.0 / 3.0 * np.pi * 0.6 ** 3))
    assert errs[1] < 0.3 * errs[0]                                               # second order in the lattice spacing


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_emulation_mesh_matches_oracle():
    _compare(_renderer(N.EMU_LIB, "cpu"), "cpu")


@pytest.mark.gpu
def test_hip_mesh_matches_oracle():
    _compare(_renderer(None, "cuda:0"), "cuda:0")
