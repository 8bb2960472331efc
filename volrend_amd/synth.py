"""Seeded synthetic PlenOctree assets (SURVEY.md 8(d) recipe).

No real ``tree.npz`` / pose files exist offline, so every BASELINE.json config is
restated on deterministic synthetic inputs written in the exact on-disk formats
the reference consumes:

* ``tree.npz``  -- svox key layout read by ``N3Tree::load_npz``
  (reference ``src/n3tree.cpp:228-362``): ``data_dim`` int64 0-d, ``data_format``
  ``<U``, ``child`` int32 ``[cap,2,2,2]`` (relative offsets, 0 = leaf),
  ``data`` float16 ``[cap,2,2,2,data_dim]``, ``invradius3`` f32[3], ``offset`` f32[3]
  (+ the svox bookkeeping keys the loader ignores, ``scripts/compress_octree.py:62-66``).
* ``pose/%04d.txt`` 4x4 c2w + ``intrinsics.txt`` as written by
  ``scripts/extract_test_poses.py:22-30`` and read by ``main_headless.cpp:40-74``.

Pure numpy; used by tests, bench.py and the CLI demo.  Not on the hot path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

SH_C0 = 0.28209479177387814

# the five configurations, restated on synthetic inputs.
CONFIGS = {
    # depth: finest leaves = 2**depth per axis; shell = half-thickness of the occupied
    # surface shell in finest-leaf widths; shape_size = primitive radius range (world units).
    # Shell / shape sizes are tuned so the node counts land on BASELINE.md's targets
    # (C1 ~2 M nodes ~1.5 GB like the real lego tree; C0 under the 171 k-node texture limit).
    "C0": dict(depth=7, fmt="SH", basis_dim=16, seed=1001, width=400, height=400, focal=555.5556,
               sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=2.0),
    "C1": dict(depth=9, fmt="SH", basis_dim=16, seed=1002, width=800, height=800, focal=1111.111,
               sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=7.5),
    # Workload-sensitivity variants of C1 (same poses / image / options; profiles/r02_workload_*.json):
    #   C1r: the SURVEY 8(d) recipe as written -- 12 small primitives, shell of 2 leaf widths (0.55 M nodes)
    #   C1t: a THIN shell (2.5 leaf widths) grown to ~2 M nodes by more primitives instead of a thicker shell
    "C1r": dict(depth=9, fmt="SH", basis_dim=16, seed=1002, width=800, height=800, focal=1111.111,
                sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.25, 0.6), shell_leaves=2.0),
    "C1t": dict(depth=9, fmt="SH", basis_dim=16, seed=1002, width=800, height=800, focal=1111.111,
                sigma=(5.0, 200.0), n_shapes=64, shape_size=(0.3, 0.7), shell_leaves=2.5,
                center_range=1.3),
    # cache-policy experiments (tools/variant_cfg.sh): C1's topology with SH9 records, C3's with SH16
    "X1": dict(depth=9, fmt="SH", basis_dim=9, seed=1002, width=800, height=800, focal=1111.111,
               sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=7.5),
    "X3": dict(depth=10, fmt="SH", basis_dim=16, seed=1004, width=1920, height=1080, focal=1166.0,
               sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=4.0),
    "C2": dict(depth=9, fmt="SH", basis_dim=25, seed=1003, width=800, height=800, focal=1111.111,
               sigma=(1.0, 10.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=7.5),
    "C3": dict(depth=10, fmt="SH", basis_dim=9, seed=1004, width=1920, height=1080, focal=1166.0,
               sigma=(5.0, 200.0), n_shapes=12, shape_size=(0.5, 1.0), shell_leaves=4.0),
}


@dataclass
class SynthTree:
    """Host-side tree in the reference's flat layout (``child_`` / ``data_``)."""

    child: np.ndarray  # int32 [cap, 2, 2, 2]
    data: np.ndarray  # float16 [cap, 2, 2, 2, data_dim]
    offset: np.ndarray  # float32 [3]
    invradius3: np.ndarray  # float32 [3]  (= TreeSpec.scale)
    data_format: str  # "SH16", "RGBA", "SG8", ...
    extra: np.ndarray | None = None  # float32, SG/ASG lobes
    depth: int = 0
    meta: dict = field(default_factory=dict)

    @property
    def capacity(self) -> int:
        return int(self.child.shape[0])

    @property
    def data_dim(self) -> int:
        return int(self.data.shape[-1])

    @property
    def N(self) -> int:
        return int(self.child.shape[1])

    @property
    def basis_dim(self) -> int:
        digits = "".join(ch for ch in self.data_format if ch.isdigit())
        return int(digits) if digits else -1

    @property
    def format_name(self) -> str:
        return "".join(ch for ch in self.data_format if ch.isalpha())

    def nbytes(self) -> int:
        return self.child.nbytes + self.data.nbytes


def _random_scene(rng: np.random.Generator, n_shapes: int, size=(0.25, 0.6), center_range=0.8):
    """Union of random spheres / axis-aligned boxes inside world radius 1.5."""
    shapes = []
    for _ in range(n_shapes):
        kind = "sphere" if rng.random() < 0.6 else "box"
        r = rng.uniform(size[0], size[1])
        c = rng.uniform(-center_range, center_range, size=3)
        if kind == "sphere":
            shapes.append(("sphere", c, np.array([r, r, r])))
        else:
            shapes.append(("box", c, rng.uniform(0.5, 1.0, size=3) * r))
    return shapes


def _sdf(shapes, p: np.ndarray) -> np.ndarray:
    """Signed distance (world units) of points p[n,3] to the union of shapes."""
    d = np.full(p.shape[0], np.inf, dtype=np.float32)
    for kind, c, h in shapes:
        q = p - c.astype(np.float32)
        if kind == "sphere":
            di = np.sqrt((q * q).sum(axis=1)) - np.float32(h[0])
        else:
            a = np.abs(q) - h.astype(np.float32)
            outside = np.sqrt((np.maximum(a, 0.0) ** 2).sum(axis=1))
            inside = np.minimum(a.max(axis=1), 0.0)
            di = outside + inside
        np.minimum(d, di, out=d)
    return d


# child slot -> (i, j, k) with x the most significant digit (n3tree_query.hpp:26-33)
_SLOT_IJK = np.array([[(s >> 2) & 1, (s >> 1) & 1, s & 1] for s in range(8)], dtype=np.int64)


def make_tree(depth: int, basis_dim: int = 16, fmt: str = "SH", seed: int = 0, n_shapes: int = 12,
              sigma=(5.0, 200.0), shape_size=(0.25, 0.6), world_radius: float = 1.5,
              shell_leaves: float = 2.0, topology_only: bool = False,
              center_range: float = 0.8) -> SynthTree:
    """Build a sparse octree refined around the surface shell of a random scene.

    ``depth`` = number of tree levels: the finest leaves tile 2**depth cells per
    axis (cube_sz = 2**depth in ``query_single_from_root``).  Nodes are numbered
    breadth-first; ``child[n,i,j,k] = child_id - n`` (relative, 0 = leaf).
    """
    assert depth >= 1
    rng = np.random.default_rng(seed)
    shapes = _random_scene(rng, n_shapes, shape_size, center_range)
    invr = np.float32(0.5 / world_radius)
    leaf_w_world = (1.0 / (1 << depth)) / float(invr)
    shell = shell_leaves * leaf_w_world

    # level 0 = root.  A node at level d has children cells of size 2**-(d+1).
    level_nodes_ijk = [np.zeros((1, 3), dtype=np.int64)]  # integer cell coords of each node
    child_blocks = []  # per level: int64 [n,8] absolute child ids (0 = leaf)
    n_total = 1
    shell_leaf_records = None
    for d in range(depth):
        ijk = level_nodes_ijk[d]
        n = ijk.shape[0]
        cell = 1.0 / (1 << (d + 1))  # child cell size in tree units
        # child cell integer coords at level d+1: 2*ijk + slot
        cijk = (ijk[:, None, :] * 2 + _SLOT_IJK[None, :, :]).reshape(-1, 3)
        centers_tree = (cijk.astype(np.float32) + 0.5) * np.float32(cell)
        centers_world = (centers_tree - 0.5) / invr
        dist = _sdf(shapes, centers_world)
        half_diag = 0.5 * np.sqrt(3.0) * cell / float(invr)
        if d + 1 < depth:
            refine = np.abs(dist) <= (shell + half_diag)
            ids = np.zeros(n * 8, dtype=np.int64)
            k = int(refine.sum())
            ids[refine] = n_total + np.arange(k, dtype=np.int64)
            n_total += k
            child_blocks.append(ids.reshape(n, 8))
            level_nodes_ijk.append(cijk[refine])
        else:
            child_blocks.append(np.zeros((n, 8), dtype=np.int64))
            in_shell = np.abs(dist) <= shell
            shell_leaf_records = (in_shell, centers_world)
    cap = n_total
    child = np.zeros((cap, 8), dtype=np.int32)
    base = 0
    for d in range(depth):
        blk = child_blocks[d]
        n = blk.shape[0]
        node_ids = (base + np.arange(n, dtype=np.int64))[:, None]
        rel = np.where(blk != 0, blk - node_ids, 0)
        child[base:base + n] = rel.astype(np.int32)
        base += n
    assert base == cap

    fmt = fmt.upper()
    if fmt == "RGBA":
        data_dim, data_format = 4, "RGBA"
    else:
        data_dim, data_format = 3 * basis_dim + 1, f"{fmt}{basis_dim}"
    meta = dict(seed=seed, depth=depth, n_nodes=cap, shell_leaves=0)
    if topology_only:
        data = np.zeros((0, 2, 2, 2, data_dim), dtype=np.float16)
        return SynthTree(child.reshape(cap, 2, 2, 2), data, np.full(3, 0.5, np.float32),
                         np.full(3, invr, np.float32), data_format, None, depth, meta)

    data = np.zeros((cap, 8, data_dim), dtype=np.float16)
    in_shell, centers_world = shell_leaf_records
    finest_base = cap - child_blocks[-1].shape[0]
    flat_idx = np.nonzero(in_shell)[0]
    nsh = flat_idx.size
    meta["shell_leaves"] = int(nsh)
    if nsh:
        node = finest_base + flat_idx // 8
        slot = flat_idx % 8
        p = centers_world[flat_idx]
        sig = np.exp(rng.uniform(np.log(sigma[0]), np.log(sigma[1]), size=nsh)).astype(np.float32)
        # smooth base colour in (0.1, 0.9)
        phase = rng.uniform(0, 2 * np.pi, size=(3, 3))
        freq = rng.uniform(1.0, 3.0, size=(3, 3))
        col = np.empty((nsh, 3), dtype=np.float32)
        for c in range(3):
            col[:, c] = 0.5 + 0.4 * np.sin(
                freq[c, 0] * p[:, 0] + phase[c, 0]) * np.sin(
                freq[c, 1] * p[:, 1] + phase[c, 1]) * np.sin(freq[c, 2] * p[:, 2] + phase[c, 2])
        rec = np.zeros((nsh, data_dim), dtype=np.float32)
        if fmt == "RGBA":
            rec[:, 0:3] = col
        elif fmt == "SH":
            band_of = np.floor(np.sqrt(np.arange(basis_dim))).astype(np.int64)
            band_scale = (0.3 / (band_of + 1.0)).astype(np.float32)
            for c in range(3):
                coeff = rng.standard_normal((nsh, basis_dim), dtype=np.float32) * band_scale[None, :]
                coeff[:, 0] = np.log(col[:, c] / (1.0 - col[:, c])) / np.float32(SH_C0)
                rec[:, c * basis_dim:(c + 1) * basis_dim] = coeff
        else:  # SG / ASG: positive lobe weights
            for c in range(3):
                coeff = rng.standard_normal((nsh, basis_dim), dtype=np.float32) * 0.5
                coeff[:, 0] += np.log(col[:, c] / (1.0 - col[:, c])) * basis_dim
                rec[:, c * basis_dim:(c + 1) * basis_dim] = coeff
        rec[:, data_dim - 1] = sig
        data[node, slot] = rec.astype(np.float16)
    extra = None
    if fmt == "SG":
        # lumisphere.hpp:30-37: per lobe [lambda, mu_x, mu_y, mu_z]
        mu = rng.standard_normal((basis_dim, 3)).astype(np.float32)
        mu /= np.linalg.norm(mu, axis=1, keepdims=True)
        lam = rng.uniform(1.0, 8.0, size=(basis_dim, 1)).astype(np.float32)
        extra = np.concatenate([lam, mu], axis=1).astype(np.float32)
    elif fmt == "ASG":
        # lumisphere.hpp:14-29: per lobe [lambda_x, lambda_y, mu_x(3), mu_y(3), mu_z(3)]
        rows = []
        for _ in range(basis_dim):
            q, _r = np.linalg.qr(rng.standard_normal((3, 3)))
            rows.append(np.concatenate([rng.uniform(1.0, 6.0, size=2), q[:, 0], q[:, 1], q[:, 2]]))
        extra = np.asarray(rows, dtype=np.float32)
    return SynthTree(child.reshape(cap, 2, 2, 2), data.reshape(cap, 2, 2, 2, data_dim),
                     np.full(3, 0.5, np.float32), np.full(3, invr, np.float32), data_format,
                     extra, depth, meta)


def make_config_tree(name: str, **overrides) -> SynthTree:
    cfg = dict(CONFIGS[name])
    cfg.update({k: v for k, v in overrides.items() if v is not None})
    t = make_tree(cfg["depth"], cfg["basis_dim"], cfg["fmt"], cfg["seed"], cfg["n_shapes"],
                  cfg["sigma"], shape_size=cfg.get("shape_size", (0.25, 0.6)),
                  shell_leaves=cfg.get("shell_leaves", 2.0),
                  topology_only=cfg.get("topology_only", False),
                  center_range=cfg.get("center_range", 0.8))
    t.meta.update(config=name)
    return t


def save_npz(tree: SynthTree, path: str, compressed: bool = False) -> None:
    """Write the svox/PlenOctree ``tree.npz`` layout (reference src/n3tree.cpp:228-362)."""
    arrays = dict(
        data_dim=np.int64(tree.data_dim),
        data_format=np.array(tree.data_format),
        child=tree.child.astype(np.int32),
        data=tree.data.astype(np.float16),
        invradius3=tree.invradius3.astype(np.float32),
        offset=tree.offset.astype(np.float32),
        # svox bookkeeping keys (ignored by the loader, dropped by compress_octree.py:62-66)
        parent_depth=np.zeros((tree.capacity, 2), dtype=np.int32),
        n_internal=np.int32(tree.capacity),
        n_free=np.int32(0),
        depth_limit=np.int32(tree.depth),
        geom_resize_fact=np.float64(1.0),
    )
    if tree.extra is not None:
        arrays["extra_data"] = tree.extra.astype(np.float32)
    (np.savez_compressed if compressed else np.savez)(path, **arrays)


def pose_spherical(theta_deg: float, phi_deg: float = -30.0, radius: float = 4.0) -> np.ndarray:
    """NeRF-synthetic camera-to-world (OpenGL convention: camera looks down -z)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    trans = np.eye(4)
    trans[2, 3] = radius
    rot_phi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                        [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], dtype=np.float64)
    rot_theta = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                          [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=np.float64)
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return swap @ rot_theta @ rot_phi @ trans


def make_poses(n: int = 200, phi_deg: float = -30.0, radius: float = 4.0) -> np.ndarray:
    """theta_i = -180 + (360/n)*i  (1.8 degrees apart for the 200-pose test set)."""
    return np.stack([pose_spherical(-180.0 + 360.0 / n * i, phi_deg, radius) for i in range(n)])


def c2w_to_transform(c2w: np.ndarray) -> np.ndarray:
    """4x4 (or 3x4) row-major c2w -> the 12-float column-major 4x3 ``CameraSpec.transform``
    (right, up, back, centre), as ``read_transform_matrices`` builds it (main_headless.cpp:40-63)."""
    m = np.asarray(c2w, dtype=np.float32)[:3, :4]
    return np.ascontiguousarray(m.T).reshape(12)


def write_pose_dir(root: str, poses: np.ndarray, width: int, focal: float) -> list[str]:
    """pose/%04d.txt + intrinsics.txt as scripts/extract_test_poses.py writes them."""
    pose_dir = os.path.join(root, "pose")
    os.makedirs(pose_dir, exist_ok=True)
    paths = []
    for i, m in enumerate(poses):
        p = os.path.join(pose_dir, f"{i:04d}.txt")
        np.savetxt(p, m)
        paths.append(p)
    K = np.diag([focal, focal, 1.0, 1.0])
    K[:2, 2] = [width / 2.0, width / 2.0]
    np.savetxt(os.path.join(root, "intrinsics.txt"), K)
    return paths
