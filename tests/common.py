"""Shared helpers for the parity tests: seeded scenes + oracle/kernel runners."""
from __future__ import annotations

import numpy as np

from oracle import binding as ob
from volrend_amd import synth


def small_scene(depth=5, basis_dim=16, fmt="SH", seed=11, **kw):
    tree = synth.make_tree(depth=depth, basis_dim=basis_dim, fmt=fmt, seed=seed, **kw)
    return tree


def camera_for(pose_idx=1, n_poses=8, size=96, focal=None, phi=-30.0, radius=4.0):
    poses = synth.make_poses(n_poses, phi_deg=phi, radius=radius)
    focal = focal if focal is not None else size * 1111.111 / 800.0
    return synth.c2w_to_transform(poses[pose_idx]), size, size, focal


def oracle_frame(tree, transform, w, h, focal, fp_mode=ob.FP_STRICT, ndc=None, region=None,
                 offscreen=True, rgba_init=None, depth_init=None, **opt_kw):
    th = ob.TreeHandle(tree, ndc=ndc)
    cam = ob.make_camera(transform, w, h, focal)
    opt = ob.default_options(**opt_kw)
    return ob.render(th, cam, opt, fp_mode, region=region, offscreen=offscreen,
                     rgba_init=rgba_init, depth_init=depth_init)


def ulp_diff(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in units-in-the-last-place between two float32 arrays."""
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def random_tree_general_n(N=4, depth=3, basis_dim=4, fmt="SH", seed=0, p_refine=0.35,
                          p_occupied=0.5):
    """A random N^3-tree with arbitrary branching factor (the svox format allows any N;
    upstream warns 'N != 2 probably doesn't work', our kernels take the literal float
    descent for it).  Breadth-first numbering, relative child offsets."""
    from volrend_amd import synth
    rng = np.random.default_rng(seed)
    N3 = N ** 3
    data_dim = 4 if fmt == "RGBA" else 3 * basis_dim + 1
    levels = [1]
    child_rows = []
    n_total = 1
    for d in range(depth):
        n = levels[d]
        if d + 1 < depth:
            refine = rng.random((n, N3)) < p_refine
        else:
            refine = np.zeros((n, N3), dtype=bool)
        ids = np.zeros((n, N3), dtype=np.int64)
        k = int(refine.sum())
        ids[refine] = n_total + np.arange(k)
        n_total += k
        child_rows.append(ids)
        levels.append(k)
        if k == 0:
            break
    cap = n_total
    child = np.zeros((cap, N3), dtype=np.int32)
    base = 0
    for ids in child_rows:
        n = ids.shape[0]
        node = (base + np.arange(n))[:, None]
        child[base:base + n] = np.where(ids != 0, ids - node, 0)
        base += n
    data = np.zeros((cap, N3, data_dim), dtype=np.float32)
    occ = (rng.random((cap, N3)) < p_occupied) & (child == 0)
    data[..., :-1] = rng.standard_normal((cap, N3, data_dim - 1)) * 0.6
    if fmt == "RGBA":
        data[..., :3] = rng.uniform(0.05, 0.95, size=(cap, N3, 3))
    data[..., -1] = np.where(occ, np.exp(rng.uniform(np.log(2), np.log(60), size=(cap, N3))), 0)
    data[child != 0] = 0
    name = "RGBA" if fmt == "RGBA" else f"{fmt}{basis_dim}"
    return synth.SynthTree(child.reshape(cap, N, N, N),
                           data.astype(np.float16).reshape(cap, N, N, N, data_dim),
                           np.full(3, 0.5, np.float32), np.full(3, np.float32(1 / 3), np.float32),
                           name, None, depth)


def write_quantised_npz(tree, path, n_retain=1, compressed=True):
    """The compress_octree.py layout (reference scripts/compress_octree.py:106-119) of a
    synthetic tree: the first ``n_retain`` basis functions uncompressed, one exact
    codebook (all distinct RGB triples, <= 65536) per remaining basis function."""
    import numpy as np
    cap, dd = tree.capacity, tree.data_dim
    nb = (dd - 1) // 3
    data = tree.data.reshape(-1, dd)
    n_slots = data.shape[0]
    coeff = data[:, :-1].reshape(n_slots, 3, nb)            # [slot, channel, basis]
    n_q = nb - n_retain
    qc = np.zeros((n_q, 65536, 3), np.float16)
    qm = np.zeros((n_q, n_slots), np.uint16)
    for j in range(n_retain, nb):
        uniq, inv = np.unique(coeff[:, :, j], axis=0, return_inverse=True)
        assert len(uniq) <= 65536
        qc[j - n_retain, :len(uniq)] = uniq
        qm[j - n_retain] = inv.reshape(-1).astype(np.uint16)
    N = tree.child.shape[1]
    arrays = dict(data_dim=np.int64(dd), data_format=np.array(tree.data_format),
                  child=tree.child, invradius3=tree.invradius3, offset=tree.offset,
                  quant_colors=qc, quant_map=qm.reshape(n_q, cap, N, N, N),
                  sigma=data[:, -1].reshape(cap, N, N, N))
    if n_retain:
        ret = np.stack([coeff[:, :, j] for j in range(n_retain)])  # [n_ret, slot, 3]
        arrays["data_retained"] = ret.reshape(n_retain, cap, N, N, N, 3)
    (np.savez_compressed if compressed else np.savez)(path, **arrays)


def random_configuration(seed):
    """Seeded random (tree, transform, w, h, focal, ndc, option kwargs): formats, basis sizes,
    odd image sizes, cameras inside the volume, degenerate thresholds, bbox / basis range /
    view rotation / depth mode, NDC."""
    import numpy as np
    rng = np.random.default_rng(9000 + seed)
    fmt, bd = [("SH", 1), ("SH", 4), ("SH", 9), ("SH", 16), ("SH", 25), ("RGBA", 0), ("SG", 4),
               ("SG", 16), ("ASG", 9), ("SG", 7)][int(rng.integers(10))]
    tree = small_scene(depth=int(rng.integers(3, 7)), basis_dim=bd, fmt=fmt,
                       seed=int(rng.integers(1 << 30)))
    w, h = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    tr = np.array(camera_for(pose_idx=int(rng.integers(8)), size=64)[0], dtype=np.float32)
    tr[9:12] *= np.float32(rng.choice([1.0, 0.6, 0.25, 0.05]))  # move towards / into the volume
    focal = float(rng.uniform(0.4, 2.5) * w)
    lo = rng.uniform(0.0, 0.4, 3)
    hi = rng.uniform(0.6, 1.0, 3)
    bmin = int(rng.integers(0, 3))
    kw = dict(step_size=float(10 ** rng.uniform(-5, -2)),
              sigma_thresh=float(rng.choice([0.0, 1e-2, 0.5, 20.0])),
              stop_thresh=float(rng.choice([0.0, 1e-3, 1e-2, 0.3])),
              background_brightness=float(rng.choice([0.0, 0.5, 1.0])),
              render_bbox=tuple(lo) + tuple(hi) if rng.random() < 0.5 else (0, 0, 0, 1, 1, 1),
              basis_minmax=(bmin, int(rng.integers(bmin, 25))) if rng.random() < 0.4 else (0, 24),
              rot_dirs=tuple(rng.normal(size=3) * 0.7) if rng.random() < 0.3 else (0, 0, 0),
              render_depth=int(rng.random() < 0.2))
    ndc = (float(w), float(h), float(focal)) if rng.random() < 0.25 else None
    return tree, tr, w, h, focal, ndc, kw, (fmt, bd)


def deep_chain_tree_n2(depth=28, basis_dim=4, fmt="SH", seed=0, target=(3.1e-4, 5.3e-4, 4.2e-4)):
    """An N = 2 tree that is a CHAIN towards `target`: at every level the child that contains the
    target is refined again (and one more child of the node into a node of leaves), `depth` levels
    deep -- a few dozen nodes with leaves of every depth 1..depth.  Deeper than 24 levels the
    integer lookup of the kernels (exact digits of a binary32 coordinate) does not apply and
    vr_tree_upload routes the tree to the literal float descent, which the reference runs for
    every tree (n3tree_query.hpp:22-47).  offset 0 / scale 1: tree coordinates = world
    coordinates, so that a camera placed AT the target (coordinates ~1e-4: binary32 resolves
    2^-36 there) starts every ray inside the deepest leaf.  Returns (tree, target as float32)."""
    from volrend_amd import synth
    rng = np.random.default_rng(seed)
    T = np.asarray(target, dtype=np.float32)
    data_dim = 4 if fmt == "RGBA" else 3 * basis_dim + 1
    child_rows, level_of = [np.zeros(8, np.int64)], [0]   # absolute child ids, 0 = leaf
    x = T.astype(np.float64).copy()
    node = 0
    for lvl in range(depth - 1):
        x *= 2.0
        k = np.floor(x).astype(np.int64)
        x -= k
        slot = int(k[0] * 4 + k[1] * 2 + k[2])            # x is the most significant digit
        nxt = len(child_rows)
        child_rows.append(np.zeros(8, np.int64))
        level_of.append(lvl + 1)
        child_rows[node][slot] = nxt
        other = int((slot + 1 + rng.integers(7)) % 8)     # one sibling becomes a node of leaves
        side = len(child_rows)
        child_rows.append(np.zeros(8, np.int64))
        level_of.append(lvl + 1)
        child_rows[node][other] = side
        node = nxt
    cap = len(child_rows)
    ids = np.stack(child_rows)
    child = np.where(ids != 0, ids - np.arange(cap)[:, None], 0).astype(np.int32)
    data = np.zeros((cap, 8, data_dim), dtype=np.float32)
    data[..., :-1] = rng.standard_normal((cap, 8, data_dim - 1)) * 0.6
    if fmt == "RGBA":
        data[..., :3] = rng.uniform(0.05, 0.95, size=(cap, 8, 3))
    lvl = np.asarray(level_of)[:, None]
    occ = (rng.random((cap, 8)) < 0.6) & (child == 0)
    # steps near the target are ~1e-8 long: deep leaves get large densities so that they still weigh in
    sig = np.where(lvl >= 14, np.exp(rng.uniform(np.log(2e3), np.log(6e4), size=(cap, 8))),
                   np.exp(rng.uniform(np.log(2.0), np.log(200.0), size=(cap, 8))))
    data[..., -1] = np.where(occ, sig, 0)
    data[child != 0] = 0
    name = "RGBA" if fmt == "RGBA" else f"{fmt}{basis_dim}"
    tree = synth.SynthTree(child.reshape(cap, 2, 2, 2), data.astype(np.float16).reshape(cap, 2, 2, 2, data_dim),
                           np.zeros(3, np.float32), np.ones(3, np.float32), name, None, depth)
    return tree, T


def camera_at(position, look_at=(0.5, 0.5, 0.5), size=40, focal=28.0):
    """12-float transform (columns right, up, back, centre) of a camera AT `position` (float32,
    kept bit for bit) looking at `look_at`."""
    c = np.asarray(position, dtype=np.float32)
    back = c.astype(np.float64) - np.asarray(look_at, np.float64)
    back /= np.linalg.norm(back)
    right = np.cross([0.0, 0.0, 1.0], back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    tr = np.concatenate([right, up, back]).astype(np.float32)
    return np.concatenate([tr, c]).astype(np.float32), size, size, focal
