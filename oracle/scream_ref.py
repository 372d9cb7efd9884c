"""CPU oracle for the SCREAM registration hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU / numpy restatement of the arithmetic of the
reference hot path.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``scream_amd/`` (the product path)
imports, links or calls anything in ``oracle/``.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's own
code in the build container (never on the GPU box) and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors, and ``tests/test_oracle_live.py`` re-checks against
the live reference whenever ``/root/reference`` is present.

Every function cites the reference file:line it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

NHEAD = 8  # models/pointnet.py:19,23-24 (nhead=8 everywhere)
LN_EPS = 1e-5  # torch.nn.LayerNorm default, models/transformer.py:71-72
ATTN_EPS = 1e-6  # models/transformer.py:12


# --------------------------------------------------------------------------- A1
def pe_dim_t(d_model: int, n_dim: int = 3, temperature: float = 10000.0) -> torch.Tensor:
    """Frequency table of PositionEmbeddingCoordsSine (models/transformer.py:148-150,168-170)."""
    num_pos_feats = d_model // n_dim // 2 * 2
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    return temperature ** (2 * torch.trunc(torch.div(i, 2)) / num_pos_feats)


def pe_sine(xyz: torch.Tensor, d_model: int) -> torch.Tensor:
    """models/transformer.py:157-179: per axis, interleaved (sin, cos) pairs, zero padded."""
    n_dim = xyz.shape[-1]
    num_pos_feats = d_model // n_dim // 2 * 2
    padding = d_model - num_pos_feats * n_dim
    dim_t = pe_dim_t(d_model, n_dim)
    p = (xyz * (1.0 * 2 * math.pi)).unsqueeze(-1) / dim_t
    emb = torch.stack([p[..., 0::2].sin(), p[..., 1::2].cos()], dim=-1)
    emb = emb.reshape(*xyz.shape[:-1], -1)
    return F.pad(emb, (0, padding))


def embed_prenorm(xyz_pe: torch.Tensor, xyz_embed: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """models/pointnet.py:45-48: pe(xyz) + Conv1d(3->d,k=1)(xyz_embed), then pre_norm LayerNorm."""
    w = sd["embedding.weight"][:, :, 0]  # [d,3]
    b = sd["embedding.bias"]
    d = w.shape[0]
    feats = pe_sine(xyz_pe, d) + (xyz_embed @ w.t() + b)
    return F.layer_norm(feats, (d,), sd["pre_norm.weight"], sd["pre_norm.bias"], LN_EPS)


# --------------------------------------------------------------------------- A3
def linear_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, want: Optional[dict] = None) -> torch.Tensor:
    """models/transformer.py:17-44 with masks unused: q [1,L,H,D], k/v [1,S,H,D]."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    S = v.size(1)
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + ATTN_EPS)
    out = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S
    if want is not None:
        want.update(Q=Q, K=K, KV=KV, Ksum=K.sum(dim=1), Z=Z)
    return out.contiguous()


# ------------------------------------------------------------------------ A2, A4
def mh_attention(xq: torch.Tensor, xk: torch.Tensor, xv: torch.Tensor, sd: Dict[str, torch.Tensor],
                 prefix: str, want: Optional[dict] = None) -> torch.Tensor:
    """models/transformer.py:74-90.  Residual of norm2 is the block input (line 88)."""
    d = xq.shape[-1]
    dim = d // NHEAD
    bs = xq.shape[0]
    q = (xq @ sd[prefix + "q_proj.weight"].t()).view(bs, -1, NHEAD, dim)
    k = (xk @ sd[prefix + "k_proj.weight"].t()).view(bs, -1, NHEAD, dim)
    v = (xv @ sd[prefix + "v_proj.weight"].t()).view(bs, -1, NHEAD, dim)
    att = linear_attention(q, k, v, want)
    msg = att.view(bs, -1, d) @ sd[prefix + "merge.weight"].t()
    m1 = F.layer_norm(msg + xq, (d,), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], LN_EPS)
    hid = torch.relu(m1 @ sd[prefix + "mlp.0.weight"].t())
    ffn = hid @ sd[prefix + "mlp.2.weight"].t()
    out = F.layer_norm(xq + ffn, (d,), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], LN_EPS)
    if want is not None:
        want.update(q=q, k=k, v=v, att=att, msg=msg, m1=m1, hid=hid, ffn=ffn, out=out)
    return out


# ------------------------------------------------------------------------ A5, A6
def layer_counts(sd: Dict[str, torch.Tensor]) -> Tuple[int, int]:
    n_self = len({k.split(".")[1] for k in sd if k.startswith("stem.")})
    n_cross2 = len({k.split(".")[1] for k in sd if k.startswith("cross.")})
    return n_self, n_cross2 // 2


def point_transformer_forward(src: torch.Tensor, tgt: torch.Tensor, sd: Dict[str, torch.Tensor],
                              src_center: Optional[torch.Tensor] = None, wants: Optional[list] = None) -> torch.Tensor:
    """models/pointnet.py:38-60: returns src_ [1,N,3] (predicted registered src coordinates).
    wants: a list that receives (prefix, side, intermediates of mh_attention) per block application, for tests that look at
    the operand ranges inside the network."""
    assert src.shape[0] == 1 and tgt.shape[0] == 1
    n_self, n_cross = layer_counts(sd)
    if src_center is None:
        src_center = torch.mean(src, dim=1, keepdim=True)  # pointnet.py:43-44
    sf = embed_prenorm(src, src - src_center, sd)
    tf = embed_prenorm(tgt, tgt, sd)
    def block(xq, xkv, p, side):
        w = {} if wants is not None else None
        out = mh_attention(xq, xkv, xkv, sd, p, w)
        if wants is not None:
            w.update(xq=xq, xkv=xkv)
            wants.append((p, side, w))
        return out

    for i in range(n_self):  # pointnet.py:50-52 -- shared weights, tgt first
        p = "stem.%d." % i
        tf = block(tf, tf, p, "tgt")
        sf = block(sf, sf, p, "src")
    for i in range(2 * n_cross):  # pointnet.py:53-57
        if i % 2 == 0:
            sf = block(sf, sf, "cross.%d." % i, "src")
        else:
            sf = block(sf, tf, "cross.%d.layer." % i, "src")
    # coor_mlp, pointnet.py:27-33,60 (Conv1d k=1 == per-point Linear with bias)
    h = torch.relu(sf @ sd["coor_mlp.0.weight"][:, :, 0].t() + sd["coor_mlp.0.bias"])
    h = torch.relu(h @ sd["coor_mlp.2.weight"][:, :, 0].t() + sd["coor_mlp.2.bias"])
    return h @ sd["coor_mlp.4.weight"][:, :, 0].t() + sd["coor_mlp.4.bias"]


def dem_transformer_forward(dsm: torch.Tensor, dem_coarse: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """models/pointnet.py:134-153 (DEMTransformer): separate stems, raw coordinates embedded for both clouds."""
    n_self = len({k.split(".")[1] for k in sd if k.startswith("stem_dsm.")})
    n_cross2 = len({k.split(".")[1] for k in sd if k.startswith("cross.")})
    sf = embed_prenorm(dsm, dsm, sd)
    tf = embed_prenorm(dem_coarse, dem_coarse, sd)
    for i in range(n_self):
        sf = mh_attention(sf, sf, sf, sd, "stem_dsm.%d." % i)
        tf = mh_attention(tf, tf, tf, sd, "stem_dem.%d." % i)
    for i in range(n_cross2):
        if i % 2 == 0:
            sf = mh_attention(sf, sf, sf, sd, "cross.%d." % i)
        else:
            sf = mh_attention(sf, tf, tf, sd, "cross.%d.layer." % i)
    h = torch.relu(sf @ sd["coor_mlp.0.weight"][:, :, 0].t() + sd["coor_mlp.0.bias"])
    h = torch.relu(h @ sd["coor_mlp.2.weight"][:, :, 0].t() + sd["coor_mlp.2.bias"])
    return h @ sd["coor_mlp.4.weight"][:, :, 0].t() + sd["coor_mlp.4.bias"]


# --------------------------------------------------------------------------- A7
def square_distance(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """utils.py:72-78 (expanded form; materialises [B,N,M])."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def nn_search(src_pred: torch.Tensor, tgt: torch.Tensor, s: float, dis_thresh: float):
    """evaluate_3d_match.py:94-95: thresholded 1-NN in metric units (squared distance vs thresh)."""
    d, idx = square_distance(src_pred / s, tgt / s)[0].min(dim=1)
    return d, idx, d < dis_thresh


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def nn_search_exact(src_pred: np.ndarray, tgt: np.ndarray, s: float, chunk: int = 1024):
    """Bit-exact numpy model of nn_search's fp32 arithmetic, O(chunk*M) memory.

    Restates what torch-CPU computes for utils.py:75-77 (verified bitwise by
    tests/test_oracle_golden.py::test_nn_exact_matches_torch):
      a = x / fp32(s) (IEEE division), dot = fma(a2,b2, fma(a1,b1, a0*b0)),
      |a|^2 = (a0^2 + a1^2) + a2^2 with every op rounded to fp32,
      dist = ((-2*dot) + |a|^2) + |b|^2, argmin ties -> lowest index.
    Products of two fp32 are exact in fp64 and |x| stays far from the fp64
    double-rounding boundary, so fp64 + round-to-fp32 models fma exactly here
    except on measure-zero ties, which the torch cross-check would flag.
    """
    s32 = np.float64(np.float32(s))
    a = _f32(np.asarray(src_pred, dtype=np.float32).astype(np.float64) / s32)
    b = _f32(np.asarray(tgt, dtype=np.float32).astype(np.float64) / s32)
    sb = _f32(_f32(_f32(b[:, 0] * b[:, 0]) + _f32(b[:, 1] * b[:, 1])) + _f32(b[:, 2] * b[:, 2]))
    n = a.shape[0]
    dmin = np.empty(n, dtype=np.float32)
    idx = np.empty(n, dtype=np.int64)
    d2nd = np.empty(n, dtype=np.float32)
    for i0 in range(0, n, chunk):
        ac = a[i0:i0 + chunk]
        sa = _f32(_f32(_f32(ac[:, 0] * ac[:, 0]) + _f32(ac[:, 1] * ac[:, 1])) + _f32(ac[:, 2] * ac[:, 2]))
        dot = _f32(ac[:, 0:1] * b[None, :, 0])
        dot = _f32(ac[:, 1:2] * b[None, :, 1] + dot)
        dot = _f32(ac[:, 2:3] * b[None, :, 2] + dot)
        d = _f32(_f32(-2.0 * dot + sa[:, None]) + sb[None, :])
        j = np.argmin(d, axis=1)  # first occurrence == torch CPU min tie-break
        idx[i0:i0 + chunk] = j
        rows = np.arange(d.shape[0])
        dmin[i0:i0 + chunk] = d[rows, j]
        if d.shape[1] > 1:
            d[rows, j] = np.inf
            d2nd[i0:i0 + chunk] = d.min(axis=1)
        else:
            d2nd[i0:i0 + chunk] = np.inf
    return dmin, idx, d2nd


# --------------------------------------------------------------------------- A8
def gather_correspondences(src: torch.Tensor, tgt: torch.Tensor, src_pred: torch.Tensor, idx: torch.Tensor,
                           valid: torch.Tensor, s: float, c: torch.Tensor, corr: str = "tgt"):
    """evaluate_3d_match.py:96-101: metric-frame correspondences A (src) and B (tgt or src_pred)."""
    A = src[:, valid] / s + c
    if corr == "tgt":
        B = tgt[:, idx[valid]] / s + c
    else:
        B = src_pred[:, valid] / s + c
    return A, B


# --------------------------------------------------------------------------- A9
def integrate_trans(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """utils.py:112-135 (batched torch branch)."""
    T = torch.eye(4)[None].repeat(R.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3:4] = t.view(-1, 3, 1)
    return T


def rigid_transform_3d(A: torch.Tensor, B: torch.Tensor, weights: Optional[torch.Tensor] = None,
                       weight_threshold: float = 0) -> torch.Tensor:
    """utils.py:138-178: weighted Kabsch.  H is formed without the dense K x K diag_embed
    (utils.py:165-166), which is the same sum; ``torch.svd`` returns V (named Vt there)."""
    bs = A.shape[0]
    if weights is None:
        weights = torch.ones_like(A[:, :, 0])
    weights = weights.clone()
    weights[weights < weight_threshold] = 0
    wsum = torch.sum(weights, dim=1, keepdim=True)[:, :, None] + 1e-6
    cA = torch.sum(A * weights[:, :, None], dim=1, keepdim=True) / wsum
    cB = torch.sum(B * weights[:, :, None], dim=1, keepdim=True) / wsum
    Am, Bm = A - cA, B - cB
    H = Am.permute(0, 2, 1) @ (weights[:, :, None] * Bm)
    U, S, V = torch.svd(H)
    delta = torch.det(V @ U.permute(0, 2, 1))
    eye = torch.eye(3)[None].repeat(bs, 1, 1)
    eye[:, -1, -1] = delta
    R = V @ eye @ U.permute(0, 2, 1)
    t = cB.permute(0, 2, 1) - R @ cA.permute(0, 2, 1)
    return integrate_trans(R, t)


# -------------------------------------------------------------------------- A10
def transformation_error(pred: torch.Tensor, gt: torch.Tensor):
    """utils.py:181-189."""
    tr = torch.trace(pred[:3, :3].T @ gt[:3, :3])
    RE = torch.acos(torch.clamp((tr - 1) / 2.0, min=-1, max=1)) * 180 / np.pi
    TE = torch.norm(pred[:3, 3:4] - gt[:3, 3:4])
    return RE, TE


# -------------------------------------------------------------------------- A11
def gt_pose_metric(rot: torch.Tensor, trans: torch.Tensor, s: float, c: torch.Tensor) -> torch.Tensor:
    """evaluate_3d_match.py:90: GT pose moved back to the metric frame. rot [3,3], trans [3,1], c [3]."""
    t = trans / s + c.view(3, 1) - torch.matmul(rot, c.view(3, 1))
    return torch.cat([torch.cat([rot, t], dim=1), torch.tensor([[0.0, 0.0, 0.0, 1.0]])], dim=0)


def rotmat2quat(R: np.ndarray) -> np.ndarray:
    """wxyz quaternion with w >= 0 -- the convention of lie/torch/so3_common.py:91-129 and of
    nibabel.quaternions.mat2quat (absent here: third-party, nibabel==3.2.1 per requirements.txt:1;
    parity with nibabel itself is UNPINNED, the lie/ implementation is the pinned stand-in)."""
    R = np.asarray(R, dtype=np.float64)
    r = math.sqrt(max(1.0 + R[0, 0] + R[1, 1] + R[2, 2], 0.0))
    if not np.isclose(r, 0.0):
        sc = 0.5 / r
        return np.array([0.5 * r, (R[2, 1] - R[1, 2]) * sc, (R[0, 2] - R[2, 0]) * sc, (R[1, 0] - R[0, 1]) * sc])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    r = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    sc = 0.5 / r
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) * sc
    q[i + 1] = 0.5 * r
    q[j + 1] = (R[i, j] + R[j, i]) * sc
    q[k + 1] = (R[k, i] + R[i, k]) * sc
    return q


def rmse_metric(trans: np.ndarray, info: np.ndarray) -> float:
    """evaluate_3d_match.py:31-50 (``RMSE``): er = [t, q_xyz]; p = er' info er / info[0,0]."""
    er = np.concatenate([trans[:3, 3], rotmat2quat(trans[:3, :3])[1:]], axis=0)
    return (er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]).item()


def point_loss(src_pred: torch.Tensor, src: torch.Tensor, rot: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """models/pointnet.py:93-99."""
    reg = (torch.matmul(rot, src.permute(0, 2, 1)) + trans).permute(0, 2, 1)
    return torch.mean(torch.sum(torch.abs(src_pred - reg), dim=-1), dim=1).mean(dim=0)


# -------------------------------------------------------------------------- A12
def normalize_pair(src: np.ndarray, tgt: np.ndarray, T: np.ndarray):
    """datasets/three_d_match.py:228-242: centre/scale to the unit ball; fp64 numpy -> fp32 tensors."""
    rot, trans = T[:3, :3], T[:3, 3:]
    reg = np.concatenate([(rot.dot(src.T) + trans).T, tgt], axis=0)
    c = np.mean(reg, axis=0)
    reg = reg - c.reshape(1, 3)
    s = 1 / np.max(np.linalg.norm(reg, axis=1)).item()
    src_n = s * (src - c)
    tgt_n = s * (tgt - c)
    trans_n = s * (trans - c.reshape(3, 1) + rot.dot(c.reshape(3, 1)))
    return (torch.Tensor(src_n), torch.Tensor(tgt_n), torch.Tensor(rot), torch.Tensor(trans_n), s, torch.Tensor(c))


# ------------------------------------------------------------- whole pair, A1-A10
def register_pair(src, tgt, rot, trans, s, c, sd, dis_thresh=0.1, corr="tgt"):
    """evaluate_3d_match.py:83-104 for one pair, without ICP: returns dict of all stage outputs.
    src/tgt [1,N,3] normalised; rot [1,3,3]; trans [1,3,1]; c [3]."""
    src_pred = point_transformer_forward(src, tgt, sd, trans.permute(0, 2, 1))
    d, idx, valid = nn_search(src_pred, tgt, s, dis_thresh)
    A, B = gather_correspondences(src, tgt, src_pred, idx, valid, s, c, corr)
    T = rigid_transform_3d(A, B)[0]
    Tgt = gt_pose_metric(rot[0], trans[0], s, c)
    re, te = transformation_error(T, Tgt)
    return dict(src_pred=src_pred, d=d, idx=idx, valid=valid, T=T, Tgt=Tgt, re=re, te=te,
                loss=point_loss(src_pred, src, rot, trans))
