"""PointTransformer: the reference's model surface (models/pointnet.py:8-99) on the HIP kernels.

The module tree exists to hold parameters under the reference's 190 state_dict names, so
``load_state_dict(torch.load("params/point-generator.pth"))`` works unchanged.  No submodule's
``forward`` is ever used: ``PointTransformer.forward`` packs the pair(s) and makes one
``scream_forward`` C-ABI call (scream_amd/csrc/forward.hip).  Running it on anything but an
MI355X raises: there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib, ops, scales
from .packing import PackedBatch

D_MODEL = 256
NHEAD = 8


class _MHAParams(nn.Module):
    """Parameter holder with the names of models/transformer.py:47-72."""

    def __init__(self, d_model: int, nhead: int = NHEAD):
        super().__init__()
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model, d_model * 4, bias=False), nn.ReLU(True),
                                 nn.Linear(d_model * 4, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _CrossParams(nn.Module):
    """models/transformer.py:110-121: the MHAttention lives under ``.layer`` and is xavier-initialised."""

    def __init__(self, d_model: int, nhead: int = NHEAD):
        super().__init__()
        self.layer = _MHAParams(d_model, nhead)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


def pe_dim_t(d_model: int = D_MODEL) -> torch.Tensor:
    """Frequency table of the sine embedding, computed with the same torch-CPU ops as
    models/transformer.py:168-170 so the table is bit-identical to the reference's."""
    npf = d_model // 3 // 2 * 2
    i = torch.arange(npf, dtype=torch.float32)
    return 10000 ** (2 * torch.trunc(torch.div(i, 2)) / npf)


class _Pack:
    """One backend's kernel-layout image of the weights (PointTransformer._pack_weights)."""

    def __init__(self, backend, sig, mt, layers, keep, fused, n_cross_batched, event, streams):
        self.backend, self.sig, self.mt, self.layers, self.keep = backend, sig, mt, layers, keep
        self.fused, self.n_cross_batched, self.event, self.streams = fused, n_cross_batched, event, streams


class PointTransformer(nn.Module):
    def __init__(self, d_model: int = 256, self_layer_num: int = 6, cross_layer_num: int = 6):
        super().__init__()
        if d_model != D_MODEL:
            raise NotImplementedError("the gfx950 kernels are built for d_model=256 (8 heads x 32), the only "
                                      "configuration the reference evaluates (evaluate_3d_match.py:188)")
        self.embedding = nn.Conv1d(3, d_model, kernel_size=1, stride=1)
        self.pre_norm = nn.LayerNorm(d_model)
        self.self_layer_num = self_layer_num
        self.cross_layer_num = cross_layer_num
        self.stem = nn.ModuleList([_MHAParams(d_model) for _ in range(self_layer_num)])
        self.cross = nn.ModuleList()
        for _ in range(cross_layer_num):
            self.cross.append(_MHAParams(d_model))
            self.cross.append(_CrossParams(d_model))
        self.coor_mlp = nn.Sequential(nn.Conv1d(d_model, d_model, 1), nn.ReLU(), nn.Conv1d(d_model, d_model, 1),
                                      nn.ReLU(), nn.Conv1d(d_model, 3, 1))
        # models/pointnet.py:36 builds a RegistrationRender here; it owns no parameters or buffers and is
        # only used when get_imgs=True (training-time GAN loss) -- out of scope, see DESIGN.md.
        self._packs = {}  # gemm backend -> _Pack: the weights in that backend's kernel layout (built on first use)
        self._ws = None

    # ------------------------------------------------------------------ weights -> kernel layout
    def _layer_modules(self) -> List[_MHAParams]:
        mods = list(self.stem)
        for i, m in enumerate(self.cross):
            mods.append(m if i % 2 == 0 else m.layer)
        return mods

    def _stem_tgt_modules(self) -> Optional[List[_MHAParams]]:
        return None  # PointTransformer: one stem for both clouds

    # GEMM path of the forward.  "h2" (default since round 3): fp16 matrix cores, operands split in two fp16 planes with
    # power-of-two scales derived from the weights (scream_amd/scales.py), three products -- fp32-level accuracy at half the
    # matrix instructions of "x3" (bf16 matrix cores, three planes, six products; scale invariant, rounds 1-2);
    # "f32" = fp32-input MFMA.  All three are held to the same parity tolerances; SCREAM_GEMM overrides.
    # "h1": ONE fp16 plane, one product -- NOT fp32-accurate: the mirror of the reference's `with autocast()` around the KITTI
    # forward (evaluate_kitti.py:37); set only by evaluate_kitti.evaluate(autocast=True) or by hand, never a default.
    gemm_backend = os.environ.get("SCREAM_GEMM", "h2")

    # split backends only: attention apply, merge + LayerNorm1 and the FFN + LayerNorm2 as one launch per block
    # (csrc/tail_split.hip); SCREAM_FUSED_TAIL=0 falls back to attn_apply + three GEMM launches (same arithmetic, the
    # intermediate activations then go through HBM)
    fused_tail = os.environ.get("SCREAM_FUSED_TAIL", "1") != "0"

    # fused tail only: project the frozen target features for ALL cross layers in one launch after the stem (and finalise their
    # K^T V images in one) instead of once per cross layer; SCREAM_BATCHED_CROSS_KV=0 keeps the per-layer launches
    batched_cross_kv = os.environ.get("SCREAM_BATCHED_CROSS_KV", "1") != "0"

    # fused tail on an fp16 split only: the layer tail of every cross-stage SELF layer also projects the queries of the cross
    # layer behind it (eight more ring stages at the end of every tile; csrc/tail_split.hip, NQ) -- six launches fewer per forward;
    # SCREAM_FUSE_NEXT_Q=0 keeps the separate projection launches
    fuse_next_q = os.environ.get("SCREAM_FUSE_NEXT_Q", "1") != "0"

    # fused tail on an fp16 split only: the q/k/v projections (and the batched target-side key/value projection) on the RING kernel
    # (csrc/proj_ring.hip, round 4: 64 rows per wave, epilogues riding under the next chunk's matrix instructions, no partial last
    # round) instead of the 8-wave GEMM; SCREAM_RING_PROJ=0 keeps the GEMM (same arithmetic per product; K^T V partials summed
    # from two halves instead of four quarters)
    ring_proj = os.environ.get("SCREAM_RING_PROJ", "1") != "0"

    # EXPERIMENTAL, off: every layer tail begins with its OWN query projection (Wq in front of the tail image; csrc/tail_split.hip,
    # QF) -- Q' is neither written by a projection launch nor read back, the self layers' projection computes key/value chunks only
    # and the cross layers launch no query projection at all.  Parity-green and +0.8 % on the bench line, but NOT repeatable bit for
    # bit when other kernels run beside it on a second stream (tools/tail_soak.py kind 3: isolated 32-row groups differ by ~1e-5
    # between two launches on the same inputs; root cause open, profiles/r04_qf_*.txt), so it is opt-in for experiments only
    # (SCREAM_Q_FIRST=1) and nothing measured or shipped uses it.
    q_first = os.environ.get("SCREAM_Q_FIRST", "0") == "1"

    def _fused_cfg(self, split) -> bool:
        return bool(split and self.fused_tail)

    def _signature(self):
        return (self.fused_tail, self.batched_cross_kv, self.fuse_next_q, self.ring_proj, self.q_first) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _layer_inputs(self):
        """(in_q, in_kv) per layer of _layer_modules() [+ per layer of _stem_tgt_modules()] and the coordinate MLP's input:
        the (gamma, beta) of the LayerNorm that produced each block input (models/pointnet.py:48-57) -- what bounds it."""
        ln = lambda n: (n.weight, n.bias)
        mods, tgt_mods = self._layer_modules(), self._stem_tgt_modules()
        ns = self.self_layer_num
        pre = ln(self.pre_norm)
        ins, prev = [], pre
        for i in range(ns):  # stem (on the source side for DEMTransformer)
            ins.append((prev, prev))
            prev = ln(mods[i].norm2)
        src_in = prev
        tgt_ins, prev_t = [], pre
        for m in (tgt_mods or []):
            tgt_ins.append((prev_t, prev_t))
            prev_t = ln(m.norm2)
        tgt_in = prev_t if tgt_mods else src_in  # the frozen target features of the cross stage
        for j in range(2 * self.cross_layer_num):
            ins.append((src_in, src_in if j % 2 == 0 else tgt_in))
            src_in = ln(mods[ns + j].norm2)
        return ins, tgt_ins, src_in

    def _pack_weights(self, backend: Optional[str] = None) -> "_Pack":
        """The weights in the kernel layout of `backend` (default: self.gemm_backend), cached PER BACKEND and rebuilt when a
        parameter or a fusion switch changes.  A caller that wants another arithmetic for one call (evaluate_kitti's
        autocast mirror) names it here instead of toggling the module attribute: no repacking on the way in and out,
        and forwards of the other backend that are still queued keep their images.
        "h2" whose weights have no fp16 exponent in range (scales.ScaleRangeError: a LayerNorm gain or weight row so large
        that |x| 2^e <= 2^15 would need e < -24) falls back to "x3" -- the scale-free bf16 x 3 split -- with a warning."""
        backend = self.gemm_backend if backend is None else backend
        if backend not in ("h2", "x3", "f32", "h1"):
            raise ValueError("gemm_backend must be 'h2', 'x3', 'f32' or 'h1', got %r" % (backend,))
        sig = self._signature()
        if self._packs and next(iter(self._packs.values())).sig != sig:
            # the weights changed: every image is stale.  Forwards that read the old ones may still be queued on other streams
            # (lanes, batches in flight) -- let them finish before their memory is released
            torch.cuda.synchronize(self.embedding.weight.device) if self.embedding.weight.is_cuda else None
            self._packs = {}
        pk = self._packs.get(backend)
        if pk is not None:
            return pk
        try:
            pk = self._build_pack(backend, sig)
        except scales.ScaleRangeError as e:
            if backend != "h2":
                raise
            import warnings
            warnings.warn("scream_amd: gemm_backend 'h2' cannot carry these weights (%s); using 'x3' (bf16 x 3 split, scale free, "
                          "same fp32-level accuracy at twice the matrix instructions)" % (e,), RuntimeWarning, stacklevel=3)
            pk = self._pack_weights("x3")
        self._packs[backend] = pk
        return pk

    def _build_pack(self, backend: str, sig) -> "_Pack":
        dev = self.embedding.weight.device
        if dev.type != "cuda":
            raise _lib.ScreamHipError("PointTransformer must be on the MI355X (net.to('cuda:0')) before forward; "
                                      "scream_amd has no CPU path")
        keep = []  # device tensors the ctypes structs point into

        def dev_f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        split = {"h2": _lib.SPLIT_H2, "x3": _lib.SPLIT_BF3, "f32": 0, "h1": _lib.SPLIT_H1}[backend]
        fp16_split = split in (_lib.SPLIT_H2, _lib.SPLIT_H1)  # the splits that carry power-of-two operand exponents

        def dev_mat(t):  # a weight MATRIX: fp32 [N,K], or its operand planes for the split GEMM; returns (pointer, exponent)
            if not split:
                return dev_f32(t), 0
            pw = ops.pack_w(t.detach().to(device=dev, dtype=torch.float32), split)
            keep.append(pw.data)
            return pw.data_ptr(), pw.w_exp

        mods = self._layer_modules()
        tgt_mods = self._stem_tgt_modules() or []
        ins, tgt_ins, coor_in = self._layer_inputs()
        layers = (_lib.LayerT * len(mods))()
        tgt_layers = (_lib.LayerT * max(len(tgt_mods), 1))()
        ns = self.self_layer_num
        # the cross layer behind every cross-stage self layer, when its query projection rides in that layer's tail
        qf = bool(fp16_split and self.fused_tail and self.q_first)  # every tail projects its own queries
        if qf:
            import warnings
            warnings.warn("scream_amd: q_first is experimental -- its layer-tail kernel is not repeatable bit for bit when other kernels run "
                          "beside it (profiles/r04_qf_experiment.txt); do not use it for results")
        next_cross = {id(mods[i]): i + 1 for i in range(ns, len(mods) - 1, 2)} if (fp16_split and self.fused_tail and self.fuse_next_q and not qf) else {}
        for L, m, (in_q, in_kv) in list(zip(layers, mods, ins)) + list(zip(tgt_layers, tgt_mods, tgt_ins)):
            # [q | k heads 0-3 | v heads 0-3 | k heads 4-7 | v heads 4-7]: a 256-wide GEMM tile then holds K and V of
            # four heads for the same tokens, which is what the fused K^T V epilogue needs (include/scream_hip.h)
            k, v = m.k_proj.weight, m.v_proj.weight
            wkv = torch.cat([k[:128], v[:128], k[128:], v[128:]], dim=0)
            L.wqkv, L.e_wqkv = dev_mat(torch.cat([m.q_proj.weight, wkv], dim=0))
            L.proj = None
            if fp16_split and self.fused_tail and self.ring_proj:  # the same matrix as the ring kernel's stage image, same exponent
                pp = ops.pack_proj(torch.cat([m.q_proj.weight, wkv], dim=0).detach().to(device=dev, dtype=torch.float32), 256, split, L.e_wqkv)
                keep.append(pp.data)
                L.proj = pp.data_ptr()
            L.wq, L.e_wq = dev_mat(m.q_proj.weight)
            L.wkv, L.e_wkv = dev_mat(wkv)
            ex = scales.layer_exps(m, in_q, in_kv) if fp16_split else {}
            L.e_xq, L.e_xkv, L.e_k, L.e_v = ex.get("e_xq", 0), ex.get("e_xkv", 0), ex.get("e_k", 0), ex.get("e_v", 0)
            wq_next = None
            if id(m) in next_cross:  # this block's output (a LayerNorm2 output) is the operand of the next layer's q_proj
                nxt = next_cross[id(m)]
                wq_next = mods[nxt].q_proj.weight
                ex.update(e_y=scales.exp_for(scales.ln_bound(*ins[nxt][0])), e_wq=scales.w_exp(wq_next))
            if qf:  # the block input (a LayerNorm output, bounded by in_q) is the operand of this layer's own q_proj
                ex.update(e_x=ex["e_xq"], e_wq=scales.w_exp(m.q_proj.weight))
            L.tail_exps = ops.tail_exps(**ex)
            L.tail = None
            L.tail_next_q = int(wq_next is not None)
            L.tail_q_first, L.proj_kv = int(qf), None
            if qf and self.ring_proj:  # the projection of such a layer: key/value chunks only
                pp = ops.pack_proj(wkv.detach().to(device=dev, dtype=torch.float32), 0, split, L.e_wkv)
                keep.append(pp.data)
                L.proj_kv = pp.data_ptr()
            if split and self.fused_tail:  # one launch for everything behind the projections (scream_layer_tail_f32)
                f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)
                img = ops.pack_tail(f32(m.merge.weight), f32(m.mlp[0].weight), f32(m.mlp[2].weight), split, L.tail_exps,
                                    Wq_next=None if wq_next is None else f32(wq_next), Wq_own=f32(m.q_proj.weight) if qf else None)
                keep.append(img.data)
                L.tail = img.data_ptr()
                L.wm, L.w1, L.w2 = None, None, None
            else:  # separate GEMMs: packed with the same exponents the bounds were derived with
                L.wm, L.e_wm_g = dev_mat(m.merge.weight)
                L.w1, L.e_w1_g = dev_mat(m.mlp[0].weight)
                L.w2, L.e_w2_g = dev_mat(m.mlp[2].weight)
            L.g1, L.b1 = dev_f32(m.norm1.weight), dev_f32(m.norm1.bias)
            L.g2, L.b2 = dev_f32(m.norm2.weight), dev_f32(m.norm2.bias)
        mt = _lib.ModelT()
        mt.n_self, mt.n_cross = self.self_layer_num, self.cross_layer_num
        mt.dim_t = dev_f32(pe_dim_t())
        mt.emb_w = dev_f32(self.embedding.weight[:, :, 0])
        mt.emb_b = dev_f32(self.embedding.bias)
        mt.pre_g, mt.pre_b = dev_f32(self.pre_norm.weight), dev_f32(self.pre_norm.bias)
        mt.layers_host = C.cast(layers, C.POINTER(_lib.LayerT))
        mt.stem_tgt_layers_host = C.cast(tgt_layers, C.POINTER(_lib.LayerT)) if tgt_mods else None
        mt.gemm_split = split
        c0w, c2w = self.coor_mlp[0].weight[:, :, 0], self.coor_mlp[2].weight[:, :, 0]
        (mt.c0_w, mt.e_c0w), mt.c0_b = dev_mat(c0w), dev_f32(self.coor_mlp[0].bias)
        (mt.c2_w, mt.e_c2w), mt.c2_b = dev_mat(c2w), dev_f32(self.coor_mlp[2].bias)
        mt.c4_w, mt.c4_b = dev_f32(self.coor_mlp[4].weight[:, :, 0]), dev_f32(self.coor_mlp[4].bias)
        mt.wkv_cross, mt.proj_cross = None, None
        if self._fused_cfg(split) and self.batched_cross_kv and self.cross_layer_num > 0:
            # the cross layers' key/value projections of the (frozen) target features as ONE GEMM, models/pointnet.py:53-57
            cross = [m for i, m in enumerate(self.cross) if i % 2 == 1]
            stack = torch.cat([torch.cat([c.layer.k_proj.weight[:128], c.layer.v_proj.weight[:128], c.layer.k_proj.weight[128:],
                                          c.layer.v_proj.weight[128:]], dim=0) for c in cross], dim=0)
            mt.wkv_cross, mt.e_wkv_cross = dev_mat(stack)
            if fp16_split and self.ring_proj:
                pp = ops.pack_proj(stack.detach().to(device=dev, dtype=torch.float32), 0, split, mt.e_wkv_cross)
                keep.append(pp.data)
                mt.proj_cross = pp.data_ptr()
            cross_L = [layers[self.self_layer_num + 2 * j + 1] for j in range(self.cross_layer_num)]
            mt.e_k_cross, mt.e_v_cross = min(L.e_k for L in cross_L), min(L.e_v for L in cross_L)  # one launch: the tightest
        if fp16_split:  # coor_mlp (models/pointnet.py:27-33): LayerNorm2 output -> Conv1d + bias, relu -> Conv1d
            mt.e_c0x = scales.exp_for(scales.ln_bound(*coor_in))
            mt.e_c2x = scales.exp_for(scales.lin_bound(c0w, *coor_in, bias=self.coor_mlp[0].bias))
        # the pack kernels and copies above were enqueued on the CURRENT stream; any other stream (a concurrent lane,
        # scream_amd/lanes.py) must order its first use of these buffers behind them (forward_packed waits once per stream)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(dev))
        return _Pack(backend, sig, mt, (layers, tgt_layers), keep, self._fused_cfg(split), self.cross_layer_num if mt.wkv_cross else 0,
                     event, {torch.cuda.current_stream(dev).cuda_stream})

    # ------------------------------------------------------------------ batched entry
    def forward_packed(self, batch: PackedBatch, return_feats: bool = False, trace=None, backend: Optional[str] = None):
        """A1-A6 for every pair of the batch in one C-ABI call; returns src_pred packed [rows_src, 3].
        backend: the arithmetic of THIS call (default: self.gemm_backend); images are cached per backend (_pack_weights)."""
        pk = self._pack_weights(backend)
        mt = pk.mt
        lib = _lib.load()
        dev = batch.xyz.device
        if ops._stream() not in pk.streams:  # first forward of this stream since the weights were packed
            torch.cuda.current_stream(dev).wait_event(pk.event)
            pk.streams.add(ops._stream())
        need = lib.scream_forward_workspace_bytes(batch.rows_src, batch.rows_total, batch.n_pairs, batch.max_chunks, int(pk.fused),
                                                  pk.n_cross_batched)
        # one scratch buffer per stream: concurrent lanes (scream_amd/lanes.py) run forwards of the same model side by side
        if self._ws is None:
            self._ws = {}
        key = (dev, ops._stream())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(need, device=dev, dtype=torch.uint8)
        bt = _lib.BatchT()
        bt.n_pairs, bt.rows_src, bt.rows_total, bt.max_chunks = batch.n_pairs, batch.rows_src, batch.rows_total, batch.max_chunks
        bt.xyz, bt.center = ops._p(batch.xyz), ops._p(batch.center)
        bt.tile_cloud = ops._p(batch.tile_cloud, torch.int32)
        bt.cloud_row0 = ops._p(batch.cloud_row0, torch.int32)
        bt.cloud_len = ops._p(batch.cloud_len, torch.int32)
        src_pred = torch.empty(batch.rows_src, 3, device=dev, dtype=torch.float32)
        feats = torch.empty(batch.rows_src, D_MODEL, device=dev, dtype=torch.float32) if return_feats else None
        _lib.check(lib.scream_forward(C.byref(mt), C.byref(bt), ws.data_ptr(), ws.numel(),
                                      src_pred.data_ptr(), feats.data_ptr() if return_feats else None,
                                      trace, ops._stream()), "scream_forward")
        return (src_pred, feats) if return_feats else src_pred

    def forward_batch(self, srcs: Sequence[torch.Tensor], tgts: Sequence[torch.Tensor],
                      centers: Optional[Sequence[Optional[torch.Tensor]]] = None) -> List[torch.Tensor]:
        """B pairs at once: srcs[i] [N_i,3], tgts[i] [M_i,3]; returns [src_pred_i [N_i,3]] -- each equal to
        what the B == 1 reference forward gives for that pair."""
        batch = PackedBatch.from_pairs(srcs, tgts, centers)
        return [t.clone() for t in batch.unpack_src(self.forward_packed(batch))]

    # ------------------------------------------------------------------ the reference's signature
    @torch.no_grad()
    def forward(self, src, tgt, src_center=None, s=1, get_imgs=False, get_transform=False, filter=None):
        """models/pointnet.py:38-91.  Inference only (the reference wraps evaluation in no_grad)."""
        assert src.shape[0] == 1, "batch size must 1"
        assert tgt.shape[0] == 1, "batch size must 1"
        if get_imgs:
            raise NotImplementedError("get_imgs=True (depth renderer for the training-time GAN loss, "
                                      "models/render.py) is out of scope; every evaluate_* call passes False")
        center = None if src_center is None else src_center.reshape(3)
        batch = PackedBatch.from_pairs([src[0]], [tgt[0]], [center])
        src_ = self.forward_packed(batch)[: src.shape[1]].unsqueeze(0).clone()
        transform = None
        if get_transform:  # pointnet.py:66-74: NN against `filter` at 0.075, Kabsch in the normalised frame
            from .geometry import register_from_prediction
            ref = tgt[0] if filter is None else filter[0]
            transform = register_from_prediction(src[0], src_[0], ref, float(s), 0.075)
        return src_, None, transform

    def loss(self, src_pred, src_pcd, rot_gt, trans_gt):
        """models/pointnet.py:93-99 (L1 point loss; metric bookkeeping, not a hot-path kernel)."""
        reg = (torch.matmul(rot_gt, src_pcd.permute([0, 2, 1])) + trans_gt).permute([0, 2, 1])
        return torch.mean(torch.sum(torch.abs(src_pred - reg), dim=-1), dim=1).mean(dim=0)


class DEMTransformer(PointTransformer):
    """models/pointnet.py:103-167 (ground generation on OpenGF): the same blocks with separate stem weights for the
    DSM cloud (``stem_dsm``) and the coarse DEM cloud (``stem_dem``), raw coordinates into the embedding for both
    clouds, no correspondence/Kabsch stage.  forward(dsm, dem_coarse, get_imgs=False) -> (dem_, imgs)."""

    def __init__(self, d_model: int = 256, self_layer_num: int = 6, cross_layer_num: int = 6):
        nn.Module.__init__(self)
        if d_model != D_MODEL:
            raise NotImplementedError("the gfx950 kernels are built for d_model=256")
        self.embedding = nn.Conv1d(3, d_model, kernel_size=1, stride=1)
        self.pre_norm = nn.LayerNorm(d_model)
        self.self_layer_num = self_layer_num
        self.cross_layer_num = cross_layer_num
        self.stem_dsm = nn.ModuleList([_MHAParams(d_model) for _ in range(self_layer_num)])
        self.stem_dem = nn.ModuleList([_MHAParams(d_model) for _ in range(self_layer_num)])
        self.cross = nn.ModuleList()
        for _ in range(cross_layer_num):
            self.cross.append(_MHAParams(d_model))
            self.cross.append(_CrossParams(d_model))
        self.coor_mlp = nn.Sequential(nn.Conv1d(d_model, d_model, 1), nn.ReLU(), nn.Conv1d(d_model, d_model, 1),
                                      nn.ReLU(), nn.Conv1d(d_model, 3, 1))
        self._packs = {}
        self._ws = None

    def _layer_modules(self) -> List[_MHAParams]:
        mods = list(self.stem_dsm)
        for i, m in enumerate(self.cross):
            mods.append(m if i % 2 == 0 else m.layer)
        return mods

    def _stem_tgt_modules(self) -> Optional[List[_MHAParams]]:
        return list(self.stem_dem)

    @torch.no_grad()
    def forward(self, dsm, dem_coarse, get_imgs=False):
        assert dsm.shape[0] == 1, "batch size must 1"
        assert dem_coarse.shape[0] == 1, "batch size must 1"
        if get_imgs:
            raise NotImplementedError("get_imgs=True (depth renderer, models/render.py) is out of scope")
        zero = torch.zeros(3, device=dsm.device)  # both clouds embed their raw coordinates (pointnet.py:138-139)
        batch = PackedBatch.from_pairs([dsm[0]], [dem_coarse[0]], [zero])
        dem_ = self.forward_packed(batch)[: dsm.shape[1]].unsqueeze(0).clone()
        return dem_, None

    def forward_batch(self, dsms, dems, centers=None):
        zero = torch.zeros(3, device=dsms[0].device)
        return super().forward_batch(dsms, dems, [zero] * len(dsms))

    def loss(self, dem_pred, dem):
        """models/pointnet.py:162-166."""
        return torch.mean(torch.sum(torch.abs(dem_pred - dem), dim=-1), dim=1).mean(dim=0)
