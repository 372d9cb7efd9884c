"""Thin host wrappers: torch device tensors in, one C-ABI call (include/scream_hip.h) each.

torch is plumbing here (device memory + the current HIP stream); all arithmetic happens in
libscream_hip.so.  There is no CPU fallback: a tensor that is not on a HIP device is an error.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib, scales
from ._lib import SPLIT_BF3, SPLIT_H1, SPLIT_H2, check

EPI_NONE, EPI_ELU1, EPI_RELU, EPI_BIAS_RELU, EPI_RES_LN, EPI_QKV = 0, 1, 2, 3, 4, 5
ROW_TILE = 128
KV_CHUNK = 256
KV_ELEMS = 33 * 32
D_MODEL = 256


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.ScreamHipError("scream_amd ops need tensors on the MI355X (got device %s); there is no CPU path" % t.device)
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def gemm_f32(A: torch.Tensor, W: torch.Tensor, epilogue: int = EPI_NONE, n_act: int = 0,
             bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(A @ W.T); A [M,K] (M % 128 == 0), W [N,K] (N % 256 == 0, K % 32 == 0)."""
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    lib = _lib.load()
    check(lib.scream_gemm_f32(_p(A), A.stride(0), _p(W), _p(out), out.stride(0), M, N, K, epilogue, n_act,
                              _p(bias), _p(residual), residual.stride(0) if residual is not None else 0,
                              _p(gamma), _p(beta), _stream()), "scream_gemm_f32")
    return out


@dataclass
class PackedW:
    """A weight matrix [N,K] in the split GEMM's operand image (scream_pack_w_split): `split` planes of 16-bit values,
    [split, K/32, N, 32]; SPLIT_H2: of W * 2^w_exp."""
    data: torch.Tensor
    split: int
    w_exp: int
    N: int
    K: int
    row_l1: Optional[torch.Tensor] = None  # [N] sum_k |W[n, k]| on the host: bounds |A W^T| by max|A| * row_l1 (gemm_qkv's K^T V exponents)

    def data_ptr(self) -> int:
        return self.data.data_ptr()


@dataclass
class PackedTail:
    """merge + mlp.0 + mlp.2 in the layer-tail kernel's 72-stage image (scream_pack_tail) with the exponents it was built for."""
    data: torch.Tensor
    split: int
    exps: _lib.TailExpsT
    next_q: bool = False  # eight more stages: the next layer's query projection (pack_tail(Wq_next=...))
    q_first: bool = False  # eight more stages IN FRONT: this layer's own query projection (pack_tail(Wq_own=...)); layer_tail(Q=None)

    def data_ptr(self) -> int:
        return self.data.data_ptr()


def default_split() -> int:
    """The operand split behind gemm_backend: 'h2' (default) -> SPLIT_H2, 'x3' -> SPLIT_BF3."""
    return SPLIT_BF3 if os.environ.get("SCREAM_GEMM", "h2") == "x3" else SPLIT_H2


def pack_w(W: torch.Tensor, split: Optional[int] = None, w_exp: Optional[int] = None) -> PackedW:
    """[N,K] fp32 (on the GPU) -> the packed operand of the split GEMMs (scream_pack_w_split).  SPLIT_BF3: planes
    p0 + p1 + p2 == W exactly.  SPLIT_H2: two fp16 planes of W * 2^w_exp (default: the largest exponent max|W| allows)."""
    split = default_split() if split is None else split
    W = W.detach().to(torch.float32).contiguous()
    N, K = W.shape
    if split in (SPLIT_H2, SPLIT_H1) and w_exp is None:
        w_exp = scales.w_exp(W)
    w_exp = int(w_exp or 0)
    dt = torch.float16 if split in (SPLIT_H2, SPLIT_H1) else torch.bfloat16
    out = torch.empty(split, K // 32, N, 32, device=W.device, dtype=dt)
    check(_lib.load().scream_pack_w_split(_p(W), N, K, split, w_exp, _p(out, dt), _stream()), "scream_pack_w_split")
    return PackedW(out, split, w_exp, N, K, W.abs().sum(dim=1).cpu() if split != SPLIT_BF3 else None)


LAYOUT_A_FRAG, LAYOUT_C_FRAG = 1, 2


def act_layout(X: torch.Tensor, to_fragment: bool) -> torch.Tensor:
    """[M,256] fp32 row-major <-> fragment-major (SCREAM_ACT_FRAG, include/scream_hip.h); returns a new tensor."""
    M = X.shape[0]
    assert X.shape[1] == D_MODEL
    out = torch.empty_like(X)
    check(_lib.load().scream_act_layout(_p(X), _p(out), M, int(to_fragment), _stream()), "scream_act_layout")
    return out


def _a_exp(A: torch.Tensor, Wp, a_exp: Optional[int]) -> int:
    """SPLIT_H2 needs |A| 2^a_exp <= 2^15.  Callers that know a bound pass it; the convenience default measures max|A|
    (a device synchronisation: tests and tools only -- the forward gets its exponents from the weights, scream_amd/scales.py)."""
    if Wp.split == SPLIT_BF3:
        return 0
    return scales.exp_for(A.abs().max().item()) if a_exp is None else int(a_exp)


def gemm_split(A: torch.Tensor, Wp: PackedW, epilogue: int = EPI_NONE, n_act: int = 0,
               bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, layout: int = 0, a_exp: Optional[int] = None) -> torch.Tensor:
    """gemm_f32's contract on the 16-bit matrix cores by operand splitting (fp32-level accuracy); Wp = pack_w(W).
    layout: LAYOUT_A_FRAG (A is fragment-major) | LAYOUT_C_FRAG (the activated query tile is written fragment-major)."""
    M, K = A.shape
    assert K == Wp.K
    if out is None:
        out = torch.empty(M, Wp.N, device=A.device, dtype=torch.float32)
    check(_lib.load().scream_gemm_split_f32(_p(A), A.stride(0), Wp.data_ptr(), _p(out), out.stride(0), M, Wp.N, K,
                                            epilogue, n_act, _p(bias), _p(residual),
                                            residual.stride(0) if residual is not None else 0, _p(gamma), _p(beta), layout,
                                            Wp.split, _a_exp(A, Wp, a_exp), Wp.w_exp, _stream()),
          "scream_gemm_split_f32")
    return out


def tail_exps(**kw) -> _lib.TailExpsT:
    e = _lib.TailExpsT()
    for k in ("e_att", "e_wm", "e_m1", "e_w1", "e_h", "e_w2", "e_y", "e_wq", "e_x", "e_q"):
        setattr(e, k, int(kw.get(k, 0)))
    return e


def pack_tail(Wm: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, split: Optional[int] = None,
              exps: Optional[_lib.TailExpsT] = None, Wq_next: Optional[torch.Tensor] = None,
              Wq_own: Optional[torch.Tensor] = None) -> PackedTail:
    """merge.weight [256,256], mlp.0.weight [1024,256], mlp.2.weight [256,1024] -> the weight image of layer_tail.
    SPLIT_H2 needs `exps` (scales.layer_exps / ops.tail_exps): the image carries e_wm, e_w1, e_w2, the kernel the rest.
    Wq_next (fp16 splits): q_proj.weight of the NEXT layer -- layer_tail(..., q_next=...) then also produces that layer's Q'
    (exps.e_y: exponent of this block's output as an operand, exps.e_wq: of Wq_next).
    Wq_own (fp16 splits, instead of Wq_next): q_proj.weight of THIS layer, its stages in front -- layer_tail(Q=None, ...) then computes
    Q' = elu(x Wq^T) + 1 itself (exps.e_x: exponent of the block input x as an operand, exps.e_wq: of Wq_own)."""
    assert Wq_next is None or Wq_own is None
    q_first = Wq_own is not None
    Wq_next = Wq_own if q_first else Wq_next
    split = default_split() if split is None else split
    Wm, W1, W2 = (w.detach().to(torch.float32).contiguous() for w in (Wm, W1, W2))
    assert Wm.shape == (D_MODEL, D_MODEL) and W1.shape == (4 * D_MODEL, D_MODEL) and W2.shape == (D_MODEL, 4 * D_MODEL)
    if split != SPLIT_BF3 and exps is None:
        raise ValueError("pack_tail on an fp16 split needs the operand exponents (scream_amd/scales.py)")
    exps = exps if exps is not None else tail_exps()
    lib = _lib.load()
    if Wq_next is not None:
        Wq_next = Wq_next.detach().to(torch.float32).contiguous()
        assert Wq_next.shape == (D_MODEL, D_MODEL)
    nbytes = lib.scream_tail_image_bytes(split, int(Wq_next is not None))
    if nbytes < 0:
        raise _lib.ScreamHipError("scream_tail_image_bytes returned %d (a next-layer query projection needs an fp16 split)" % nbytes)
    out = torch.empty(nbytes, device=Wm.device, dtype=torch.uint8)
    check(lib.scream_pack_tail(_p(Wm), _p(W1), _p(W2), _p(Wq_next), int(q_first), split, C.byref(exps), _p(out, torch.uint8), _stream()), "scream_pack_tail")
    return PackedTail(out, split, exps, Wq_next is not None and not q_first, q_first)


def kv_finalize_image(partial: torch.Tensor, cloud_row0, cloud_len, row_base: int, cloud_begin: int, n_kv: int,
                   n_clouds: int, out: Optional[torch.Tensor] = None, split: Optional[int] = None) -> torch.Tensor:
    """K^T V partials of gemm_qkv -> the per-cloud operand image of layer_tail ([n_clouds, kv_image_bytes] uint8) for the tail of
    the same `split` (default: default_split()): three bf16 planes, or two fp16 planes with a per-head exponent chosen on the device.
    partial [L, M/128, 8, 1056] (a batched key/value projection of L layers): returns [L, n_clouds, kv_image_bytes]."""
    split = default_split() if split is None else split
    lib = _lib.load()
    L = partial.shape[0] if partial.dim() == 4 else 1
    img = lib.scream_kv_image_bytes()
    if out is None:
        out = torch.zeros((L, n_clouds, img) if partial.dim() == 4 else (n_clouds, img), device=partial.device, dtype=torch.uint8)
    check(lib.scream_kv_finalize_image(_p(partial), _p(cloud_row0, torch.int32), _p(cloud_len, torch.int32), row_base,
                                    cloud_begin, n_kv, _p(out, torch.uint8), L, partial[0].numel() if L > 1 else 0,
                                    n_clouds * img if L > 1 else 0, split, _stream()), "scream_kv_finalize_image")
    return out


def layer_tail(Q: torch.Tensor, kv_image: torch.Tensor, tile_cloud, kv_cloud_offset: int, cloud_len, x: torch.Tensor,
               tail: PackedTail, g1, b1, g2, b2, out: Optional[torch.Tensor] = None,
               q_next: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention apply + merge + norm1 + FFN + norm2 of one block in one launch (scream_layer_tail_f32).
    Q, x and the result are FRAGMENT-major [M,256] matrices (act_layout converts).  q_next (images packed with Wq_next): also
    receives elu(y Wq_next^T) + 1, fragment-major; it may be Q itself."""
    M = x.shape[0]
    assert x.shape[1] == D_MODEL and (Q is None or x.shape == Q.shape)
    assert (Q is None) == tail.q_first, "Q = None goes with an image that carries this layer's own query stages (pack_tail(Wq_own=...))"
    assert (q_next is not None) == tail.next_q, "q_next goes with an image that carries the next layer's query stages"
    if out is None:
        out = torch.empty(M, D_MODEL, device=x.device, dtype=torch.float32)
    check(_lib.load().scream_layer_tail_f32(_p(Q), _p(kv_image, torch.uint8), _p(tile_cloud, torch.int32),
                                            kv_cloud_offset, _p(cloud_len, torch.int32), _p(x),
                                            tail.data_ptr(), _p(g1), _p(b1), _p(g2), _p(b2), _p(out), _p(q_next),
                                            M, tail.split, C.byref(tail.exps), _stream()), "scream_layer_tail_f32")
    return out


def gemm_qkv(A: torch.Tensor, W, n_q: int, tile_cloud, cloud_row0, cloud_len, row_base: int, layout: int = 0,
             a_exp: Optional[int] = None, k_exp: Optional[int] = None, v_exp: Optional[int] = None):
    """Fused q/k/v projection (scream_gemm_qkv_f32 / scream_gemm_qkv_split_f32 for W = pack_w(...)).  Returns
    (Q' [M,256] or None, kv_partial [M/128,8,1056]).  layout (split kernel only): LAYOUT_A_FRAG | LAYOUT_C_FRAG."""
    M, K = A.shape
    sp = isinstance(W, PackedW)  # packed operand planes -> the split kernel
    N = W.N if sp else W.shape[0]
    Q = torch.empty(M, n_q, device=A.device, dtype=torch.float32) if n_q else None
    L = (N - n_q) // 512  # key/value tile pairs: > 1 for a batched projection of several layers (split kernel, n_q == 0)
    part = torch.empty((L, M // ROW_TILE, 8, KV_ELEMS) if L > 1 else (M // ROW_TILE, 8, KV_ELEMS), device=A.device, dtype=torch.float32)
    args = (_p(A), A.stride(0), W.data_ptr() if sp else _p(W), _p(Q), n_q, M, N, K, n_q,
            _p(tile_cloud, torch.int32), _p(cloud_row0, torch.int32), _p(cloud_len, torch.int32), row_base, _p(part))
    if sp:
        if W.split != SPLIT_BF3 and (k_exp is None or v_exp is None):
            # the K^T V epilogue's operands: |k|, |v| <= max|A| * (L1 norm of their weight rows); key rows are the first 128 of
            # every 256 behind the queries (include/scream_hip.h).  Convenience default (a device sync), as for a_exp.
            amax = float(A.abs().max().item())
            rl = W.row_l1[n_q:].view(-1, 2, 128)
            k_exp = scales.exp_for(1.0 + amax * float(rl[:, 0].max())) if k_exp is None else k_exp
            v_exp = scales.exp_for(amax * float(rl[:, 1].max())) if v_exp is None else v_exp
        check(_lib.load().scream_gemm_qkv_split_f32(*args, layout, W.split, _a_exp(A, W, a_exp), W.w_exp, int(k_exp or 0),
                                                    int(v_exp or 0), _stream()),
              "scream_gemm_qkv_split_f32")
    else:
        assert layout == 0
        check(_lib.load().scream_gemm_qkv_f32(*args, _stream()), "scream_gemm_qkv")
    return Q, part


@dataclass
class PackedProj:
    """A q/k/v (n_q = 256, N = 768) or stacked key/value (n_q = 0, N = 512 L) weight matrix in the ring projection kernel's stage
    image (scream_pack_proj): fp16 planes of W * 2^w_exp, one 32-column chunk over K = 256 per stage."""
    data: torch.Tensor
    split: int
    w_exp: int
    N: int
    n_q: int
    row_l1: Optional[torch.Tensor] = None

    def data_ptr(self) -> int:
        return self.data.data_ptr()


def pack_proj(W: torch.Tensor, n_q: int, split: Optional[int] = None, w_exp: Optional[int] = None) -> PackedProj:
    """[N,256] fp32 in the row order of gemm_qkv -> the stage image of proj_qkv (fp16 splits only)."""
    split = SPLIT_H2 if split is None else split
    W = W.detach().to(torch.float32).contiguous()
    N, K = W.shape
    assert K == D_MODEL
    w_exp = scales.w_exp(W) if w_exp is None else int(w_exp)
    nbytes = _lib.load().scream_proj_image_bytes(N, split)
    if nbytes <= 0:
        raise _lib.ScreamHipError("scream_proj_image_bytes(%d, %d) = %d" % (N, split, nbytes))
    out = torch.empty(nbytes, device=W.device, dtype=torch.uint8)
    check(_lib.load().scream_pack_proj(_p(W), N, n_q, split, w_exp, _p(out, torch.uint8), _stream()), "scream_pack_proj")
    return PackedProj(out, split, w_exp, N, n_q, W.abs().sum(dim=1).cpu())


def proj_qkv(x_frag: torch.Tensor, P: PackedProj, tile_cloud, cloud_row0, cloud_len, row_base: int,
             a_exp: Optional[int] = None, k_exp: Optional[int] = None, v_exp: Optional[int] = None):
    """The fused q/k/v projection on the ring kernel (scream_proj_qkv_f32): x and Q' fragment-major.  Returns (Q' or None,
    kv_partial) like gemm_qkv."""
    M = x_frag.shape[0]
    Q = torch.empty(M, D_MODEL, device=x_frag.device, dtype=torch.float32) if P.n_q else None
    L = (P.N - P.n_q) // 512
    part = torch.empty((L, M // ROW_TILE, 8, KV_ELEMS) if L > 1 else (M // ROW_TILE, 8, KV_ELEMS), device=x_frag.device, dtype=torch.float32)
    if a_exp is None or k_exp is None or v_exp is None:  # convenience defaults (a device sync): tests and tools only
        amax = float(x_frag.abs().max().item())
        rl = P.row_l1[P.n_q:].view(-1, 2, 128)
        a_exp = scales.exp_for(amax) if a_exp is None else a_exp
        k_exp = scales.exp_for(1.0 + amax * float(rl[:, 0].max())) if k_exp is None else k_exp
        v_exp = scales.exp_for(amax * float(rl[:, 1].max())) if v_exp is None else v_exp
    check(_lib.load().scream_proj_qkv_f32(_p(x_frag), P.data_ptr(), _p(Q), M, P.N, P.n_q, _p(tile_cloud, torch.int32),
                                          _p(cloud_row0, torch.int32), _p(cloud_len, torch.int32), row_base, _p(part), P.split,
                                          int(a_exp), P.w_exp, int(k_exp), int(v_exp), _stream()), "scream_proj_qkv_f32")
    return Q, part


def kv_finalize(part, cloud_row0, cloud_len, row_base: int, cloud_begin: int, n_kv: int, n_clouds: int) -> torch.Tensor:
    kv = torch.zeros(n_clouds, 8, KV_ELEMS, device=part.device, dtype=torch.float32)
    check(_lib.load().scream_kv_finalize(_p(part), _p(cloud_row0, torch.int32), _p(cloud_len, torch.int32), row_base,
                                         cloud_begin, n_kv, _p(kv), _stream()), "scream_kv_finalize")
    return kv


def pe_embed_ln(xyz, tile_cloud, center, dim_t, emb_w, emb_b, gamma, beta, frag: bool = False) -> torch.Tensor:
    """A1.  frag=True: the same values in the fragment-major layout (include/scream_hip.h SCREAM_ACT_FRAG; act_layout undoes it)."""
    rows = xyz.shape[0]
    feats = torch.empty(rows, D_MODEL, device=xyz.device, dtype=torch.float32)
    lib = _lib.load()
    fn, name = (lib.scream_pe_embed_ln_frag, "scream_pe_embed_ln_frag") if frag else (lib.scream_pe_embed_ln, "scream_pe_embed_ln")
    check(fn(_p(xyz), _p(tile_cloud, torch.int32), _p(center), _p(dim_t), _p(emb_w),
             _p(emb_b), _p(gamma), _p(beta), _p(feats), rows, _stream()), name)
    return feats


def kv_reduce(Kf: torch.Tensor, Vf: torch.Tensor, ld: int, row_base: int, cloud_row0, cloud_len, cloud_begin: int,
              n_kv: int, max_chunks: int, n_clouds: int) -> torch.Tensor:
    """Kf / Vf: views whose data_ptr is column 0 of the keys / values (row stride ld floats)."""
    dev = Kf.device
    partial = torch.empty(max(n_kv * max_chunks * 8 * KV_ELEMS, 1), device=dev, dtype=torch.float32)
    kv = torch.zeros(n_clouds, 8, KV_ELEMS, device=dev, dtype=torch.float32)
    for t in (Kf, Vf):
        if not t.is_cuda or t.dtype != torch.float32:
            raise TypeError("kv_reduce needs fp32 device tensors")
    check(_lib.load().scream_kv_reduce(Kf.data_ptr(), Vf.data_ptr(), ld, row_base, _p(cloud_row0, torch.int32),
                                       _p(cloud_len, torch.int32), cloud_begin, n_kv, max_chunks, _p(partial),
                                       _p(kv), _stream()), "scream_kv_reduce")
    return kv


def attn_apply(Qf: torch.Tensor, ldq: int, kv, tile_cloud, kv_cloud_offset: int, cloud_len, rows: int) -> torch.Tensor:
    out = torch.empty(rows, D_MODEL, device=kv.device, dtype=torch.float32)
    check(_lib.load().scream_attn_apply(Qf.data_ptr(), ldq, _p(kv), _p(tile_cloud, torch.int32), kv_cloud_offset,
                                        _p(cloud_len, torch.int32), _p(out), D_MODEL, rows, _stream()),
          "scream_attn_apply")
    return out


def coor_head(X: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    rows = X.shape[0]
    out = torch.empty(rows, 3, device=X.device, dtype=torch.float32)
    check(_lib.load().scream_coor_head(_p(X), _p(W), _p(b), _p(out), rows, _stream()), "scream_coor_head")
    return out


def nn_search(query: torch.Tensor, ref: torch.Tensor, q_row0, q_len, r_row0, r_len, s: torch.Tensor,
              max_q_len: int, max_r_len: int, thresh: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Packed thresholded 1-NN.  Returns (idx int32, dmin fp32, valid uint8), one entry per packed query row."""
    dev = query.device
    qn, rn = query.shape[0], ref.shape[0]
    n_pairs = s.shape[0]
    ref_prep = torch.empty(max(rn, 1), 4, device=dev, dtype=torch.float32)
    keys = torch.empty(max(qn, 1), device=dev, dtype=torch.int64)
    idx = torch.empty(qn, device=dev, dtype=torch.int32)
    dmin = torch.empty(qn, device=dev, dtype=torch.float32)
    valid = torch.empty(qn, device=dev, dtype=torch.uint8)
    check(_lib.load().scream_nn_search(_p(query), _p(ref), _p(q_row0, torch.int32), _p(q_len, torch.int32),
                                       _p(r_row0, torch.int32), _p(r_len, torch.int32), _p(s), n_pairs, max_q_len,
                                       max_r_len, qn, rn, float(thresh), _p(ref_prep), _p(keys, torch.int64),
                                       _p(idx, torch.int32), _p(dmin), _p(valid, torch.uint8), _stream()),
          "scream_nn_search")
    return idx, dmin, valid


def kabsch_corr(src, ref, src_row0, src_len, ref_row0, idx, valid, s, c) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (T [n_pairs,4,4], n_corr int32 [n_pairs])."""
    n_pairs = s.shape[0]
    T = torch.empty(n_pairs, 4, 4, device=src.device, dtype=torch.float32)
    n_corr = torch.empty(n_pairs, device=src.device, dtype=torch.int32)
    check(_lib.load().scream_kabsch_corr(_p(src), _p(ref), _p(src_row0, torch.int32), _p(src_len, torch.int32),
                                         _p(ref_row0, torch.int32), _p(idx, torch.int32), _p(valid, torch.uint8),
                                         _p(s), _p(c), n_pairs, _p(T), _p(n_corr, torch.int32), _stream()),
          "scream_kabsch_corr")
    return T, n_corr


def rigid_transform_3d_dense(A: torch.Tensor, B: torch.Tensor, w: Optional[torch.Tensor], thr: float) -> torch.Tensor:
    bs, K = A.shape[0], A.shape[1]
    T = torch.empty(bs, 4, 4, device=A.device, dtype=torch.float32)
    check(_lib.load().scream_rigid_transform_3d(_p(A) if K else None, _p(B) if K else None, _p(w), float(thr), bs, K,
                                                _p(T), _stream()), "scream_rigid_transform_3d")
    return T


def transformation_error_batched(T_pred: torch.Tensor, T_gt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    n = T_pred.shape[0]
    re = torch.empty(n, device=T_pred.device, dtype=torch.float32)
    te = torch.empty(n, device=T_pred.device, dtype=torch.float32)
    check(_lib.load().scream_transformation_error(_p(T_pred), _p(T_gt), n, _p(re), _p(te), _stream()),
          "scream_transformation_error")
    return re, te


def point_loss(src_pred: torch.Tensor, src: torch.Tensor, src_row0, src_len, rot: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """models/pointnet.py:93-99 per pair of a packed batch: loss[p] = mean_n sum_xyz |src_pred_n - (rot_p src_n + trans_p)|.
    src_pred / src packed [rows,3]; rot [B,3,3]; trans [B,3,1] or [B,3]."""
    B = rot.shape[0]
    out = torch.empty(B, device=src_pred.device, dtype=torch.float32)
    check(_lib.load().scream_point_loss(_p(src_pred), _p(src), _p(src_row0, torch.int32), _p(src_len, torch.int32),
                                        _p(rot.reshape(B, 9).contiguous()), _p(trans.reshape(B, 3).contiguous()), B, _p(out), _stream()),
          "scream_point_loss")
    return out


def icp_p2p(src, ref, src_row0, src_len, ref_row0, ref_len, s, c, T_init, max_src_len: int, max_ref_len: int,
            max_corr_dist: float, max_iter: int = 30, rel_fitness: float = 1e-6, rel_rmse: float = 1e-6):
    """Batched point-to-point ICP (scream_icp_p2p).  Returns (T [n,4,4], fitness_rmse [n,2], iters int32 [n])."""
    lib = _lib.load()
    n_pairs = s.shape[0]
    dev = src.device
    T = T_init.detach().clone().contiguous().float()
    fr = torch.empty(n_pairs, 2, device=dev, dtype=torch.float32)
    iters = torch.empty(n_pairs, device=dev, dtype=torch.int32)
    need = lib.scream_icp_workspace_bytes(src.shape[0], ref.shape[0], n_pairs)
    ws = torch.empty(need, device=dev, dtype=torch.uint8)
    check(lib.scream_icp_p2p(_p(src), _p(ref), _p(src_row0, torch.int32), _p(src_len, torch.int32),
                             _p(ref_row0, torch.int32), _p(ref_len, torch.int32), _p(s), _p(c), n_pairs, max_src_len,
                             max_ref_len, src.shape[0], ref.shape[0], float(max_corr_dist), int(max_iter),
                             float(rel_fitness), float(rel_rmse), _p(T), _p(fr), _p(iters, torch.int32),
                             ws.data_ptr(), need, _stream()), "scream_icp_p2p")
    return T, fr, iters


ICP_ASYNC_ITERS = 64  # schedules up to this many iterations are enqueued whole (icp_p2p); longer ones in pieces (IcpRun)


class IcpRun:
    """A LONG ICP schedule (KITTI: up to 1000 iterations, evaluate_kitti.py:64-70) enqueued in pieces, so that launching stops
    once every pair has stopped WITHOUT the host ever waiting inside the C ABI (scream_icp_p2p_range): ``advance(n)`` enqueues the
    next n launches on the current stream and, behind them, a copy of the per-pair stopped flags into pinned memory + an event;
    ``all_stopped()`` waits for THAT event only (the caller calls it when it collects the batch, i.e. after the following batches
    were enqueued) and reads the flags.  ``T`` holds the refined transform of every pair whose flag is set; ``finish()`` enqueues
    what is left of the schedule.  Results are those of icp_p2p whatever the piece sizes (tests/test_gpu_evaluate.py)."""

    def __init__(self, src, ref, src_row0, src_len, ref_row0, ref_len, s, c, T_init, max_src_len: int, max_ref_len: int,
                 max_corr_dist: float, max_iter: int, rel_fitness: float = 1e-6, rel_rmse: float = 1e-6):
        self.lib = _lib.load()
        n = s.shape[0]
        dev = src.device
        self.n, self.max_iter, self.next_it = n, int(max_iter), 0
        self.T = T_init.detach().clone().contiguous().float()
        self.fr = torch.empty(n, 2, device=dev, dtype=torch.float32)
        self.iters = torch.empty(n, device=dev, dtype=torch.int32)
        self.flags = torch.zeros(n, device=dev, dtype=torch.int32)
        self.flags_host = torch.zeros(n, dtype=torch.int32, pin_memory=True)
        need = self.lib.scream_icp_workspace_bytes(src.shape[0], ref.shape[0], n)
        self.ws = torch.empty(need, device=dev, dtype=torch.uint8)
        self._keep = (src, ref, src_row0, src_len, ref_row0, ref_len, s, c)
        self._args = (_p(src), _p(ref), _p(src_row0, torch.int32), _p(src_len, torch.int32), _p(ref_row0, torch.int32),
                      _p(ref_len, torch.int32), _p(s), _p(c), n, int(max_src_len), int(max_ref_len), src.shape[0], ref.shape[0],
                      float(max_corr_dist), int(max_iter), float(rel_fitness), float(rel_rmse), _p(self.T), _p(self.fr),
                      _p(self.iters, torch.int32))
        self.event = None

    @property
    def launches_left(self) -> int:
        return self.max_iter + 2 - self.next_it

    def advance(self, n_launches: int) -> None:
        end = min(self.next_it + int(n_launches), self.max_iter + 2)
        check(self.lib.scream_icp_p2p_range(*self._args, self.next_it, end, _p(self.flags, torch.int32), self.ws.data_ptr(),
                                            self.ws.numel(), _stream()), "scream_icp_p2p_range")
        self.next_it = end
        self.flags_host.copy_(self.flags, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(self.T.device))

    def all_stopped(self) -> bool:
        """Waits for the flags of the last advance() (not for the stream's later work) and reads them."""
        self.event.synchronize()
        return self.launches_left == 0 or bool(self.flags_host.all())
