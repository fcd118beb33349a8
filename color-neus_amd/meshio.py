"""Mesh files of the evaluation path: what NeuS_Trainer.validate_mesh writes through trimesh (NeuS_Trainer.py:287-307) --
binary little-endian PLY, vertices float32 x y z (+ uchar red green blue alpha when coloured), faces ``list uchar int vertex_indices``."""
import numpy as np


def write_ply(path, vertices, triangles, colors=None):
    """vertices (V,3) float, triangles (F,3) int, colors (V,3) float in [0,1] or uint8 (optional)."""
    v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32).reshape(-1, 3))
    f = np.ascontiguousarray(np.asarray(triangles, dtype=np.int32).reshape(-1, 3))
    head = ["ply", "format binary_little_endian 1.0", "comment color-neus_amd", f"element vertex {v.shape[0]}",
            "property float x", "property float y", "property float z"]
    vdt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if colors is not None:
        c = np.asarray(colors)
        if c.dtype != np.uint8:
            c = (np.clip(c, 0.0, 1.0) * 255.0).astype(np.uint8)     # NeuS_Trainer.py:292 (colors * 255).astype(np.uint8)
        c = c.reshape(-1, 3)
        head += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
        vdt += [("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")]
    head += [f"element face {f.shape[0]}", "property list uchar int vertex_indices", "end_header"]
    vrec = np.empty(v.shape[0], dtype=vdt)
    vrec["x"], vrec["y"], vrec["z"] = v[:, 0], v[:, 1], v[:, 2]
    if colors is not None:
        vrec["red"], vrec["green"], vrec["blue"], vrec["alpha"] = c[:, 0], c[:, 1], c[:, 2], 255
    frec = np.empty(f.shape[0], dtype=[("n", "u1"), ("i", "<i4", (3,))])
    frec["n"], frec["i"] = 3, f
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        fh.write(vrec.tobytes())
        fh.write(frec.tobytes())


def read_ply(path):
    """Reader for the files write_ply produces (tests / round trips): (vertices, triangles, colors or None)."""
    with open(path, "rb") as fh:
        lines = []
        while True:
            ln = fh.readline().decode("ascii").strip()
            lines.append(ln)
            if ln == "end_header":
                break
        nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        nf = int([l for l in lines if l.startswith("element face")][0].split()[-1])
        colored = any(l.startswith("property uchar red") for l in lines)
        vdt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")] + ([("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")] if colored else [])
        v = np.frombuffer(fh.read(nv * np.dtype(vdt).itemsize), dtype=vdt)
        f = np.frombuffer(fh.read(nf * 13), dtype=[("n", "u1"), ("i", "<i4", (3,))])
    verts = np.stack([v["x"], v["y"], v["z"]], -1)
    cols = np.stack([v["red"], v["green"], v["blue"]], -1) if colored else None
    return verts, f["i"].copy(), cols
