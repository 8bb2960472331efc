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
