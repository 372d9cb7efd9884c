"""Seeded synthetic weights and point-cloud pairs (pure numpy; PCG64 is stable across machines).

There are no pretrained weights or datasets in the container or on the GPU box
(reference README.md:12-20 points at external downloads), so parity fixtures,
tests and ``bench.py`` use what this module generates:

* ``make_state_dict``   -- the 190-tensor PointTransformer state_dict layout of the
                           reference (models/pointnet.py:9-36, models/transformer.py:47-72,110-115).
* ``make_3dmatch_pair`` -- indoor-room-like clouds voxelised at 0.0625 m
                           (process_3d_match.py:30) -> about 5k points per cloud.
* ``make_kitti_pair``   -- LiDAR-ring-like clouds voxelised at 0.7 m (process_kitti.py:55-56).
* ``make_uniform_pair`` -- N = M uniform points in the unit ball (roofline config).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Tuple

import numpy as np
import torch

__all__ = ["state_dict_keys", "make_state_dict", "random_rotation", "voxel_downsample",
           "make_3dmatch_pair", "make_kitti_pair", "make_uniform_pair", "synthetic_info_matrix"]


def _mha_keys(prefix: str, d: int):
    return [
        (prefix + "q_proj.weight", (d, d)), (prefix + "k_proj.weight", (d, d)),
        (prefix + "v_proj.weight", (d, d)), (prefix + "merge.weight", (d, d)),
        (prefix + "mlp.0.weight", (4 * d, d)), (prefix + "mlp.2.weight", (d, 4 * d)),
        (prefix + "norm1.weight", (d,)), (prefix + "norm1.bias", (d,)),
        (prefix + "norm2.weight", (d,)), (prefix + "norm2.bias", (d,)),
    ]


def state_dict_keys(d_model: int = 256, n_self: int = 6, n_cross: int = 6):
    """(name, shape) in the reference's state_dict order."""
    d = d_model
    keys = [("embedding.weight", (d, 3, 1)), ("embedding.bias", (d,)),
            ("pre_norm.weight", (d,)), ("pre_norm.bias", (d,))]
    for i in range(n_self):
        keys += _mha_keys("stem.%d." % i, d)
    for i in range(2 * n_cross):
        keys += _mha_keys("cross.%d." % i if i % 2 == 0 else "cross.%d.layer." % i, d)
    keys += [("coor_mlp.0.weight", (d, d, 1)), ("coor_mlp.0.bias", (d,)),
             ("coor_mlp.2.weight", (d, d, 1)), ("coor_mlp.2.bias", (d,)),
             ("coor_mlp.4.weight", (3, d, 1)), ("coor_mlp.4.bias", (3,))]
    return keys


def dem_state_dict_keys(d_model: int = 256, n_self: int = 6, n_cross: int = 6):
    """(name, shape) of DEMTransformer's state_dict (models/pointnet.py:104-131) in registration order."""
    d = d_model
    keys = [("embedding.weight", (d, 3, 1)), ("embedding.bias", (d,)),
            ("pre_norm.weight", (d,)), ("pre_norm.bias", (d,))]
    for stem in ("stem_dsm", "stem_dem"):
        for i in range(n_self):
            keys += _mha_keys("%s.%d." % (stem, i), d)
    for i in range(2 * n_cross):
        keys += _mha_keys("cross.%d." % i if i % 2 == 0 else "cross.%d.layer." % i, d)
    keys += [("coor_mlp.0.weight", (d, d, 1)), ("coor_mlp.0.bias", (d,)),
             ("coor_mlp.2.weight", (d, d, 1)), ("coor_mlp.2.bias", (d,)),
             ("coor_mlp.4.weight", (3, d, 1)), ("coor_mlp.4.bias", (3,))]
    return keys


def make_state_dict(seed: int, d_model: int = 256, n_self: int = 6, n_cross: int = 6, dem: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 weights: matrices U(-1/sqrt(fan_in), +), LN gain 1 +- 0.1, LN/conv bias +- 0.1."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape in (dem_state_dict_keys if dem else state_dict_keys)(d_model, n_self, n_cross):
        if "norm" in name and name.endswith("weight"):
            w = 1.0 + 0.1 * rng.uniform(-1, 1, size=shape)
        elif name.endswith("bias"):
            w = 0.1 * rng.uniform(-1, 1, size=shape)
        else:
            fan_in = shape[1]
            b = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-b, b, size=shape)
        sd[name] = torch.from_numpy(w.astype(np.float32))
    return sd


def make_trained_like_state_dict(seed: int, d_model: int = 256, n_self: int = 6, n_cross: int = 6, dem: bool = False,
                                 model_scale_log2: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 weights with the statistics training leaves behind -- the stand-in for params/point-generator.pth
    (evaluate_3d_match.py:188-191), which is not in the image -- where make_state_dict gives the gentlest case default
    initialisation can: LayerNorm gains log-uniform in [0.05, 8] with three channels per norm a further x 30, LayerNorm
    biases +- 2, other biases +- 0.5; weight matrices U(-1/sqrt(fan_in), +) with log-normally scaled rows (sigma 0.7) and
    0.2 % outlier elements x 8 .. 20; every matrix finally times 2^model_scale_log2 (whole-model rescaling).  These are the
    weights the fp16 x 2 split's weight-derived exponents (scream_amd/scales.py) have to carry."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape in (dem_state_dict_keys if dem else state_dict_keys)(d_model, n_self, n_cross):
        if "norm" in name and name.endswith("weight"):
            w = np.exp(rng.uniform(math.log(0.05), math.log(8.0), size=shape))
            w[rng.choice(shape[0], size=3, replace=False)] *= 30.0
            w *= rng.choice([-1.0, 1.0], size=shape, p=[0.1, 0.9])  # gains do change sign in trained nets
        elif "norm" in name and name.endswith("bias"):
            w = rng.uniform(-2.0, 2.0, size=shape)
        elif name.endswith("bias"):
            w = rng.uniform(-0.5, 0.5, size=shape)
        else:
            fan_in = shape[1]
            b = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-b, b, size=shape)
            w *= np.exp(rng.normal(0.0, 0.7, size=(shape[0],) + (1,) * (len(shape) - 1)))
            out = rng.uniform(size=shape) < 0.002
            w = np.where(out, w * rng.uniform(8.0, 20.0, size=shape), w)
            w *= 2.0 ** model_scale_log2
        sd[name] = torch.from_numpy(w.astype(np.float32))
    return sd


def random_rotation(rng: np.random.Generator, max_angle_deg: float = 180.0) -> np.ndarray:
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = math.radians(max_angle_deg) * rng.uniform(-1, 1)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def voxel_downsample(pts: np.ndarray, voxel: float) -> np.ndarray:
    """One centroid per occupied voxel (what open3d's voxel_down_sample does)."""
    key = np.floor(pts / voxel).astype(np.int64)
    key -= key.min(axis=0)
    dims = key.max(axis=0) + 1
    lin = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    order = np.argsort(lin, kind="stable")
    lin_s = lin[order]
    first = np.concatenate([[True], lin_s[1:] != lin_s[:-1]])
    group = np.cumsum(first) - 1
    n_groups = group[-1] + 1
    sums = np.zeros((n_groups, 3))
    np.add.at(sums, group, pts[order])
    cnt = np.bincount(group, minlength=n_groups)[:, None]
    return sums / cnt


def _room_points(rng: np.random.Generator, n_samples: int) -> np.ndarray:
    """Three box faces meeting at a corner of a 3.0 x 2.5 x 2.5 m room + two interior 1.5 x 1.6 m planar patches."""
    lx, ly, lz = 3.0, 2.5, 2.5
    areas = np.array([lx * ly, lx * lz, ly * lz, 2.4, 2.4])
    counts = np.maximum((areas / areas.sum() * n_samples).astype(int), 1)
    parts = []
    u = rng.uniform(size=(counts[0], 2)); parts.append(np.stack([u[:, 0] * lx, u[:, 1] * ly, np.zeros(counts[0])], 1))
    u = rng.uniform(size=(counts[1], 2)); parts.append(np.stack([u[:, 0] * lx, np.zeros(counts[1]), u[:, 1] * lz], 1))
    u = rng.uniform(size=(counts[2], 2)); parts.append(np.stack([np.zeros(counts[2]), u[:, 0] * ly, u[:, 1] * lz], 1))
    for k in (3, 4):
        origin = rng.uniform([0.5, 0.5, 0.3], [lx - 1.0, ly - 1.0, lz - 1.0])
        Rp = random_rotation(rng, 60.0)
        u = rng.uniform(size=(counts[k], 2))
        parts.append(origin + (np.stack([1.5 * u[:, 0], 1.6 * u[:, 1], np.zeros(counts[k])], 1) @ Rp.T))
    pts = np.concatenate(parts, 0) + 0.03125  # walls sit mid-voxel, as a scanned wall rarely straddles two cells
    return pts + rng.normal(scale=0.005, size=pts.shape)


def _arc_crop(pts: np.ndarray, centre: np.ndarray, start: float, frac: float) -> np.ndarray:
    ang = np.mod(np.arctan2(pts[:, 1] - centre[1], pts[:, 0] - centre[0]) - start, 2 * math.pi)
    return pts[ang < frac * 2 * math.pi]


def synthetic_info_matrix(rng: np.random.Generator) -> np.ndarray:
    """A 6x6 SPD matrix with the block structure of the benchmark's gt.info entries
    (datasets/three_d_match.py:11-27 parses them): [[n*I, G],[G^T, J]]."""
    n = 5000.0
    p = rng.uniform(-2.5, 2.5, size=3) * n
    G = np.array([[0, p[2], -p[1]], [-p[2], 0, p[0]], [p[1], -p[0], 0]])
    J = np.diag(rng.uniform(2.5e4, 4e4, size=3)) + rng.uniform(-3e3, 3e3, size=(3, 3))
    J = 0.5 * (J + J.T)
    M = np.zeros((6, 6))
    M[:3, :3] = n * np.eye(3)
    M[:3, 3:] = G
    M[3:, :3] = G.T
    M[3:, 3:] = J + G.T @ G / n
    return M.astype(np.float32)


def make_3dmatch_pair(seed: int, kind: str = "3dmatch", voxel: float = 0.0625, n_samples: int = 120000):
    """Returns raw metric-frame (src [N,3] f64, tgt [M,3] f64, T [4,4] f64, idx [2] i64,
    covariance [6,6] f32, scene_idx int) in the on-disk convention of
    process_3d_match.py:38-40,199-200 (T registers src onto tgt)."""
    rng = np.random.default_rng(seed)
    world = _room_points(rng, n_samples)
    centre = np.array([1.5, 1.25, 1.2]) + rng.uniform(-0.2, 0.2, size=3)
    start = rng.uniform(0, 2 * math.pi)
    if kind == "3dmatch":
        shift = rng.uniform(0.15, 0.45)
    elif kind == "lo":
        shift = rng.uniform(0.5, 0.62)
    else:  # "zero": disjoint crops
        shift = 0.5
    frac = 0.7 if kind != "zero" else 0.45
    tgt = voxel_downsample(_arc_crop(world, centre, start, frac), voxel)
    src_world = voxel_downsample(_arc_crop(world, centre, start + shift * 2 * math.pi, frac), voxel)
    R = random_rotation(rng, 60.0)
    t = rng.uniform(-1.5, 1.5, size=(3, 1))
    src = (src_world - t.T) @ R  # R @ src + t == src_world
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3:] = t
    return src, tgt, T, np.array([0, 2], dtype=np.int64), synthetic_info_matrix(rng), int(seed % 8)


def make_kitti_pair(seed: int, voxel: float = 0.7, n_rings: int = 64, pts_per_ring: int = 2000):
    """Ring-structured LiDAR-like ground/obstacle cloud in a 120 x 120 x 6 m box, two poses <= 12 m / 10 deg apart."""
    rng = np.random.default_rng(seed)
    n_obs = 60
    obs_c = rng.uniform(-55, 55, size=(n_obs, 2))
    obs_r = rng.uniform(1.0, 4.0, size=n_obs)
    obs_h = rng.uniform(1.5, 5.5, size=n_obs)

    def scan(origin_xy, yaw):
        az = rng.uniform(0, 2 * math.pi, size=(n_rings, pts_per_ring))
        elev = np.linspace(math.radians(-24), math.radians(2), n_rings)[:, None]
        h = 1.8
        rng_ground = np.where(elev < -0.01, h / np.tan(-elev), 80.0)
        r = np.minimum(np.broadcast_to(rng_ground, az.shape), 60.0)
        x = origin_xy[0] + r * np.cos(az + yaw)
        y = origin_xy[1] + r * np.sin(az + yaw)
        z = np.where(r < 60.0, 0.0, h + r * np.tan(elev))
        pts = np.stack([x.ravel(), y.ravel(), np.broadcast_to(z, az.shape).ravel()], 1)
        d = np.linalg.norm(pts[:, None, :2] - obs_c[None], axis=2)  # obstacles lift points
        hit = d < obs_r[None]
        lift = np.where(hit, obs_h[None] * rng.uniform(0, 1, size=hit.shape), 0).max(axis=1)
        pts[:, 2] = np.clip(pts[:, 2] + lift, 0, 6.0)
        pts += rng.normal(scale=0.02, size=pts.shape)
        keep = (np.abs(pts[:, 0]) < 60) & (np.abs(pts[:, 1]) < 60)
        return pts[keep]

    tgt_w = scan(np.zeros(2), 0.0)
    yaw = math.radians(rng.uniform(-10, 10))
    dt = rng.uniform(-1, 1, size=2)
    dt = dt / np.linalg.norm(dt) * rng.uniform(2, 12)
    src_w = scan(dt, yaw)
    tgt = voxel_downsample(tgt_w, voxel)
    src_world = voxel_downsample(src_w, voxel)
    R = np.array([[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1.0]])
    t = np.array([[dt[0]], [dt[1]], [0.0]])
    src = (src_world - t.T) @ R
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3:] = t
    return src, tgt, T


def make_uniform_pair(seed: int, n: int, m: int = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """n src / m tgt points uniform in a ball, small relative pose."""
    rng = np.random.default_rng(seed)
    m = n if m is None else m

    def ball(k):
        v = rng.normal(size=(k, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v * rng.uniform(size=(k, 1)) ** (1 / 3)

    tgt = ball(m)
    src_world = ball(n)
    R = random_rotation(rng, 30.0)
    t = rng.uniform(-0.2, 0.2, size=(3, 1))
    src = (src_world - t.T) @ R
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3:] = t
    return src, tgt, T
