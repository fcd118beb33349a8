"""Device iso-surface extraction (cnr_mc_count / cnr_mc_emit): mesh INVARIANTS.  The comparison with the independent
float64 restatement of the published algorithm lives in tests/test_mc_oracle.py; here are the properties any correct marching-cubes mesh
of a closed level set satisfies:
every edge is shared by exactly two triangles with opposite orientation, the Euler characteristic of a sphere is 2 (a torus 0),
vertices lie on the lattice edges at the linear zero crossing, normals point out of the inside region, and the enclosed volume
converges to the analytic one.  The CPU-emulation build and the HIP kernels must emit the identical mesh."""
import os

import numpy as np
import pytest
import torch

import _native as N
import color_neus_amd as cn


def _renderer(library, device):
    from oracle import colorneus_oracle as O
    cfg = O.tiny_config()
    return N.make_renderer(cfg, O.init_params(cfg, seed=1, trained_like=True), library, device)


def _lattice(res, f, device):
    lin = torch.linspace(-1.0, 1.0, res)
    x, y, z = torch.meshgrid(lin, lin, lin, indexing="ij")
    return f(x, y, z).float().contiguous().to(device)


def _mesh_checks(v, t, closed=True):
    assert t.min() >= 0 and t.max() < len(v)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], 0)
    key = e[:, 0].astype(np.int64) * (len(v) + 1) + e[:, 1]
    rkey = e[:, 1].astype(np.int64) * (len(v) + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key), "a directed edge appears twice: inconsistent orientation"
    if closed:
        assert np.array_equal(np.sort(key), np.sort(rkey)), "every edge must be shared by two triangles, opposite directions"
    und = np.unique(np.sort(e, 1), axis=0)
    assert len(np.unique(t.reshape(-1))) == len(v), "unreferenced vertices"
    return len(v) - len(und) + len(t)    # Euler characteristic


def _volume(v, t):
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)


def _run(library, device):
    r = _renderer(library, device)
    out = {}
    # sphere of radius 0.6: u = r0 - |x| > 0 inside
    for res in (24, 41):
        u = _lattice(res, lambda x, y, z: 0.6 - torch.sqrt(x * x + y * y + z * z), device)
        v, t = r.marching_cubes(u, [-1, -1, -1], [1, 1, 1], 0.0)
        v, t = v.cpu().numpy().astype(np.float64), t.cpu().numpy()
        assert _mesh_checks(v, t) == 2
        rad = np.linalg.norm(v, axis=1)
        h = 2.0 / (res - 1)
        assert np.abs(rad - 0.6).max() < 0.5 * h * h / 0.6 + 1e-5      # linear interpolation of a distance field: O(h^2 / r)
        vol = _volume(v, t)
        assert vol > 0, "normals must point out of the u > threshold region"
        assert abs(vol - 4.0 / 3.0 * np.pi * 0.6 ** 3) < 3.0 * h * h   # second-order convergence of the enclosed volume
        # vertices sit on lattice edges: two coordinates are lattice values
        g = (v + 1.0) / h
        on_lattice = np.abs(g - np.round(g)) < 1e-3
        assert (on_lattice.sum(1) >= 2).all()
        out[("sphere", res)] = (v, t)
    # torus (genus 1) and two disjoint blobs (two components): Euler characteristic 0 and 4
    u = _lattice(40, lambda x, y, z: 0.2 - torch.sqrt((torch.sqrt(x * x + y * y) - 0.55) ** 2 + z * z), device)
    v, t = r.marching_cubes(u, [-1, -1, -1], [1, 1, 1], 0.0)
    assert _mesh_checks(v.cpu().numpy().astype(np.float64), t.cpu().numpy()) == 0
    u = _lattice(36, lambda x, y, z: torch.maximum(0.3 - torch.sqrt((x - 0.45) ** 2 + y * y + z * z), 0.25 - torch.sqrt((x + 0.5) ** 2 + y * y + z * z)), device)
    v, t = r.marching_cubes(u, [-1, -1, -1], [1, 1, 1], 0.0)
    assert _mesh_checks(v.cpu().numpy().astype(np.float64), t.cpu().numpy()) == 4
    # a rough random field (every one of the 256 cell patterns, ambiguous faces included): closed, consistently oriented surface
    g = torch.Generator().manual_seed(3)
    u = torch.randn(20, 20, 20, generator=g)
    u[0], u[-1], u[:, 0], u[:, -1], u[:, :, 0], u[:, :, -1] = -1, -1, -1, -1, -1, -1   # level set stays off the boundary
    v, t = r.marching_cubes(u.to(device), [0, 0, 0], [19, 19, 19], 0.0)
    _mesh_checks(v.cpu().numpy().astype(np.float64), t.cpu().numpy())
    out["noise"] = (v.cpu().numpy(), t.cpu().numpy())
    # non-zero threshold, anisotropic bounds, empty result
    u = _lattice(16, lambda x, y, z: x, device)
    v, t = r.marching_cubes(u, [0, 0, 0], [2, 4, 8], 0.25)
    assert np.allclose(v.cpu().numpy()[:, 0], (0.25 + 1.0) / 2.0 * 2.0, atol=1e-5) and len(t) == 2 * 15 * 15
    v, t = r.marching_cubes(u, [0, 0, 0], [1, 1, 1], 5.0)
    assert len(v) == 0 and len(t) == 0
    return out


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_marching_cubes_emu():
    _run(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_marching_cubes_hip_matches_emulation_and_properties():
    got = _run(None, "cuda:0")
    if os.path.isfile(N.EMU_LIB):
        ref = _run(N.EMU_LIB, "cpu")
        for k in got:
            assert np.array_equal(got[k][1], ref[k][1]), k                     # identical triangles (same ownership / scan order)
            assert np.abs(got[k][0] - ref[k][0]).max() < 1e-6, k


def test_ply_round_trip(tmp_path):
    from color_neus_amd import meshio
    g = np.random.default_rng(0)
    v, t, c = g.standard_normal((50, 3)), g.integers(0, 50, (80, 3)), g.random((50, 3))
    p = str(tmp_path / "m.ply")
    meshio.write_ply(p, v, t, c)
    v2, t2, c2 = meshio.read_ply(p)
    assert np.allclose(v2, v.astype(np.float32)) and np.array_equal(t2, t) and np.array_equal(c2, (np.clip(c, 0, 1) * 255).astype(np.uint8))
    head = open(p, "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0") and "property list uchar int vertex_indices" in open(p, "rb").read(400).decode("ascii", "ignore")
    meshio.write_ply(p, v, t)
    v3, t3, c3 = meshio.read_ply(p)
    assert c3 is None and np.array_equal(t3, t)
