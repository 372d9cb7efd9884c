"""3DMatch-family evaluation harness (evaluate_3d_match.py:31-183) over batched, sharded pairs.

Per batch of B pairs: forward (A1-A6), thresholded 1-NN (A7), fused gather + Kabsch (A8/A9) and RE/TE
(A10) run on the MI355X in a handful of launches; the per-pair metric bookkeeping (A11: RMSE with the
6x6 information matrix, success, per-scene lists) stays on the host like the reference's.

Deliberate differences from evaluate_3d_match.py, all documented in DESIGN.md:
  * pairs are processed B at a time and may be sharded over ranks (the reference: one pair at a time);
  * the ``open3d.registration_icp`` refinement (:106-119) runs on the MI355X: ``evaluate_loader`` and the three
    ``evaluate_3d_*`` wrappers default to ``icp="gpu"`` -- the batched point-to-point ICP of scream_icp_p2p (same loop
    as open3d's RegistrationICP, max_correspondence_distance 0.1, 30 iterations, kept only where it improves both RE
    and TE, :117) -- because the reference ALWAYS refines and reports RR/RRE/RTE of the refined pose; ``icp=None``
    opts out (pre-ICP metrics), ``icp=callable`` plugs in anything else (e.g. a wrapper around open3d).  open3d is
    not in this image, so parity with open3d itself is unpinned and every other parity statement is made on the
    pre-ICP pose (the parity tests pass ``icp=None`` explicitly);
  * ``nibabel.quaternions.mat2quat`` (:46) is replaced by an in-repo wxyz, w >= 0 quaternion;
  * a scene with no counted pair is skipped in the scene mean instead of raising ZeroDivisionError (:160).
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import dist as sdist
from . import lanes as _lanes
from . import ops
from .data import SCENE_NAMES, collate_pairs, unpack_batch
from .geometry import processbar, register_batch
from .packing import PackedBatch, StagingBuffers


ICP_MAX_CORR_DIST = 0.1  # evaluate_3d_match.py:111 (metres; KITTI uses 1 and 1000 iterations, evaluate_kitti.py:64-70)
ICP_MAX_ITER = 30        # open3d's default ICPConvergenceCriteria


def mat2quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> wxyz quaternion with w >= 0 (the convention RMSE needs from
    nibabel.quaternions.mat2quat, evaluate_3d_match.py:46; algorithm as lie/torch/so3_common.py:91-129)."""
    R = np.asarray(R, dtype=np.float64)
    tr = 1.0 + R[0, 0] + R[1, 1] + R[2, 2]
    r = math.sqrt(max(tr, 0.0))
    if r > 1e-8:
        k = 0.5 / r
        return np.array([0.5 * r, (R[2, 1] - R[1, 2]) * k, (R[0, 2] - R[2, 0]) * k, (R[1, 0] - R[0, 1]) * k])
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))  # rotation by ~pi: pivot on the largest diagonal entry
    j, k_ = (i + 1) % 3, (i + 2) % 3
    r = math.sqrt(R[i, i] - R[j, j] - R[k_, k_] + 1.0)
    k = 0.5 / r
    q = np.zeros(4)
    q[0], q[1 + i], q[1 + j], q[1 + k_] = (R[k_, j] - R[j, k_]) * k, 0.5 * r, (R[i, j] + R[j, i]) * k, (R[k_, i] + R[i, k_]) * k
    return q if q[0] >= 0 else -q


def RMSE(trans: np.ndarray, info: np.ndarray) -> float:
    """evaluate_3d_match.py:31-50: er = [t, q_xyz] of the residual pose; er' info er / info[0,0]."""
    er = np.concatenate([trans[:3, 3], mat2quat(trans[:3, :3])[1:]], axis=0)
    return (er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]).item()


def gt_pose_metric(rot: torch.Tensor, trans: torch.Tensor, s: float, c: torch.Tensor) -> torch.Tensor:
    """evaluate_3d_match.py:90 in fp32: [R | t/s + c - R c]."""
    t = trans / s + c.view(3, 1) - torch.matmul(rot, c.view(3, 1))
    return torch.cat([torch.cat([rot, t], dim=1), torch.tensor([[0.0, 0.0, 0.0, 1.0]])], dim=0)


def _strip(item):
    """Accept both raw dataset items and DataLoader(batch_size=1) collations of them."""
    src, tgt, rot, trans, s, idx, cov, c, scene = item
    if torch.is_tensor(src) and src.dim() == 3:
        src, tgt, rot, trans, idx, cov, c = src[0], tgt[0], rot[0], trans[0], idx[0], cov[0], c[0]
        s = s[0] if torch.is_tensor(s) or isinstance(s, (list, tuple)) else s
        scene = scene[0] if torch.is_tensor(scene) or isinstance(scene, (list, tuple)) else scene
    s = float(s.item()) if torch.is_tensor(s) else float(s)
    scene = int(scene.item()) if torch.is_tensor(scene) else int(scene)
    return src.float(), tgt.float(), rot.float(), trans.float(), s, [int(v) for v in idx], np.asarray(cov, dtype=np.float32), c.float(), scene


def aggregate_rows(rows: np.ndarray, re_static_method: str = "median"):
    """evaluate_3d_match.py:147-171 from gathered rows.  Returns (point_trans_loss, rre, rte, rr): the loss
    averaged over all pairs; rre/rte/rr averaged over scenes of the per-scene median (or mean) / recall
    over counted pairs (|idx1 - idx0| > 1), failures contributing 0 to the RE/TE lists (:131-138)."""
    n = rows.shape[0]
    loss = float(rows[:, sdist.COL_LOSS].sum() / max(n, 1))
    stat = np.median if re_static_method == "median" else np.mean
    rre = rte = rr = 0.0
    used = 0
    for sc in range(len(SCENE_NAMES)):
        sel = rows[(rows[:, sdist.COL_SCENE] == sc) & (rows[:, sdist.COL_COUNTED] > 0)]
        if sel.shape[0] == 0:
            continue  # the reference divides by zero here (evaluate_3d_match.py:160); we skip the scene
        ok = sel[:, sdist.COL_SUCCESS] > 0
        rre += float(stat(np.where(ok, sel[:, sdist.COL_RE], 0.0)))
        rte += float(stat(np.where(ok, sel[:, sdist.COL_TE], 0.0)))
        rr += float(ok.sum() / sel.shape[0])
        used += 1
    used = max(used, 1)
    return loss, rre / used, rte / used, rr / used


_staging = {}  # (device, lane stream) -> StagingBuffers
# Batches evaluate_loader keeps enqueued, one stream each.  Measured end to end with the GPU ICP on (tools/eval_e2e.py, 4 096 pairs,
# profiles/r03_eval_e2e.txt): 2 -> 1 366 pairs/s, 3 -> 1 510, 4 -> 1 535, 6 -> 1 564 (without ICP 1 512 / 1 665 / 1 618 / 1 664): with two, the
# device ran one batch alone whenever the host was collecting the other.  Each costs one forward workspace and one staging set.
IN_FLIGHT = int(os.environ.get("SCREAM_IN_FLIGHT", "4"))


def _register_lane(net, its, centers, pair_ids, corr, dis_thresh, icp, icp_dist, icp_iters, device, pred_hook, backend=None):
    """Device part of register_items for one lane (runs on the current stream): pack, A1-A6, A7-A9, A10 (+ GPU ICP).
    Nothing here blocks the host: the batch and every per-pair scalar go up in two asynchronous copies from pinned
    memory (a pageable copy or a torch.tensor(..., device=) would wait for everything already queued on the stream)."""
    B = len(its)
    T_gt = torch.stack([gt_pose_metric(it[2], it[3], it[4], it[5]) for it in its])
    extra = np.concatenate([np.array([it[4] for it in its], dtype=np.float32),                       # s      [B]
                            torch.stack([it[5].reshape(3) for it in its]).numpy().reshape(-1),        # c      [B,3]
                            T_gt.numpy().astype(np.float32).reshape(-1),                              # T_gt   [B,4,4]
                            torch.stack([it[2].reshape(3, 3) for it in its]).numpy().reshape(-1),     # rot    [B,3,3]
                            torch.stack([it[3].reshape(3) for it in its]).numpy().reshape(-1)])       # trans  [B,3]
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    stg = _staging.setdefault(key, StagingBuffers())
    batch, ex = PackedBatch.from_host([it[0] for it in its], [it[1] for it in its], centers, device, extra, stg)
    stg.uploaded()
    s, c = ex[:B], ex[B:4 * B].view(B, 3)
    T_gt_d = ex[4 * B:20 * B].view(B, 4, 4)
    rot_d, trans_d = ex[20 * B:29 * B].view(B, 3, 3), ex[29 * B:32 * B].view(B, 3, 1)
    src_pred = net.forward_packed(batch) if backend is None else net.forward_packed(batch, backend=backend)
    if pred_hook is not None:
        src_pred = pred_hook(batch, src_pred, pair_ids)
    T, n_corr, idx, dmin, valid = register_batch(batch, src_pred, s, c, dis_thresh, corr)
    re, te = ops.transformation_error_batched(T, T_gt_d)
    out = {"T_gt": T_gt, "T_gt_d": T_gt_d, "run": None}
    if isinstance(icp, str):
        if icp != "gpu":
            raise ValueError("icp must be None, 'gpu' or a callable")
        # evaluate_3d_match.py:106-119 on the MI355X for the whole batch: refine from the Kabsch pose, keep the
        # refinement only where it improves both RE and TE against the ground truth (:117)
        T0, re0, te0 = T, re, te

        def accept(T2):
            re2, te2 = ops.transformation_error_batched(T2, T_gt_d)
            better = (re2 <= re0) & (te2 <= te0)
            return torch.where(better[:, None, None], T2, T0), torch.where(better, re2, re0), torch.where(better, te2, te0)

        tgt_row0 = (batch.tgt_row0 - batch.rows_src).contiguous()
        icp_args = (batch.xyz[: batch.rows_src], batch.xyz[batch.rows_src:], batch.src_row0, batch.src_len_dev, tgt_row0,
                    batch.tgt_len_dev, s, c, T, max(batch.src_len), max(batch.tgt_len), icp_dist, icp_iters)
        if icp_iters <= ops.ICP_ASYNC_ITERS:  # the whole schedule in one asynchronous call
            T, re, te = accept(ops.icp_p2p(*icp_args)[0])
        else:
            # A long schedule (KITTI: 1000 iterations) usually stops within tens: its first piece is enqueued here together with
            # everything behind the ICP, computed as if every pair had stopped by then; finish() reads the stopped flags when it
            # collects the batch and, only if a pair was still moving, enqueues more pieces and this tail again.  The host never
            # waits while it enqueues (until round 3 the C call itself synchronised every 32 iterations).
            run = ops.IcpRun(*icp_args)
            run.advance(ops.ICP_ASYNC_ITERS)
            run.accept, run.stream = accept, torch.cuda.current_stream(device)
            out["run"] = run
            T, re, te = accept(run.T)
    # PointTransformer.loss per pair (models/pointnet.py:93-99, evaluate_3d_match.py:86) for the whole batch in one launch
    loss = ops.point_loss(src_pred.contiguous(), batch.xyz[: batch.rows_src], batch.src_row0, batch.src_len_dev, rot_d, trans_d)
    out.update(T=T, re=re, te=te, loss=loss)
    return out


@torch.no_grad()
def register_items_async(net, its: Sequence[tuple], centers: Sequence[torch.Tensor], pair_ids: Sequence[int],
                         corr: str = "tgt", dis_thresh: float = 0.1, icp=None, icp_dist: float = ICP_MAX_CORR_DIST,
                         icp_iters: int = ICP_MAX_ITER, device: Optional[torch.device] = None,
                         pred_hook: Optional[Callable] = None, lanes: Optional[int] = None,
                         stream: Optional[torch.cuda.Stream] = None, backend: Optional[str] = None) -> Callable[[], tuple]:
    """Enqueue A1-A10 (+ optional GPU ICP) for one batch and return ``finish()``, which waits for the device and gives
    the host arrays of register_items.  Everything between the call and ``finish()`` overlaps the GPU work, which is
    how evaluate_loader hides the host side of batch i-1 and the packing of batch i+1 behind batch i.
    ``stream``: run the WHOLE batch -- one packed forward, search, solve, ICP, result copy -- on that stream and leave the
    current stream alone (evaluate_loader alternates two such streams between consecutive batches, below).
    ``backend``: the forward's arithmetic for THIS batch (PointTransformer.forward_packed(backend=); None = the model's own)."""
    device = device or next(net.parameters()).device
    if stream is not None:
        lanes = 1
        stream.wait_stream(torch.cuda.current_stream(device))  # (normally idle: orders the batch behind whatever the caller queued)
    if lanes is None:
        lanes = _lanes.DEFAULT_LANES if len(its) >= 8 else 1
    parts = _lanes.split_weighted([it[0].shape[0] + it[1].shape[0] for it in its], lanes)
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        outs = _lanes.run(device, parts, lambda rg: _register_lane(
            net, [its[i] for i in rg], [centers[i] for i in rg], [pair_ids[i] for i in rg], corr, dis_thresh, icp, icp_dist,
            icp_iters, device, pred_hook, backend))
        pub = {}

        def publish():  # one small device buffer -> one asynchronous D2H copy into pinned memory
            T = torch.cat([o["T"] for o in outs])
            re, te = torch.cat([o["re"] for o in outs]), torch.cat([o["te"] for o in outs])
            loss = torch.cat([o["loss"] for o in outs]).reshape(-1)
            flat = torch.cat([T.reshape(-1), re, te, loss]).float()
            host = torch.empty(flat.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(flat, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(device))
            pub.update(flat=flat, host=host, done=done)

        publish()
        T_gt = torch.cat([o["T_gt"] for o in outs])
        T_gt_d = torch.cat([o["T_gt_d"] for o in outs])
    keep = [outs, pub]  # lane-stream allocations stay referenced until the copy has been consumed

    def continue_long_icp():
        """Only for schedules longer than ops.ICP_ASYNC_ITERS whose first piece left a pair moving: more pieces, then the ICP's
        tail and the result copy again (scream_amd/ops.py: IcpRun).  Runs when the batch is COLLECTED, i.e. behind the enqueueing
        of the following batches -- the device has their work while the host waits here."""
        redo = False
        for o in outs:
            run = o["run"]
            if run is None or run.all_stopped():
                continue
            redo = True
            with torch.cuda.stream(run.stream):
                piece = 2 * ops.ICP_ASYNC_ITERS
                while not run.all_stopped():
                    run.advance(piece)
                    piece *= 2
                o["T"], o["re"], o["te"] = run.accept(run.T)
                ev = torch.cuda.Event()
                ev.record(run.stream)
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                torch.cuda.current_stream(device).wait_event(ev)
        if redo:
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                publish()

    def finish():
        continue_long_icp()
        pub["done"].synchronize()
        B = len(its)
        h = pub["host"].numpy().astype(np.float64)
        T_h = h[:16 * B].reshape(B, 4, 4).astype(np.float32)
        re_h, te_h, loss_h = h[16 * B:17 * B].copy(), h[17 * B:18 * B].copy(), h[18 * B:19 * B].copy()
        if callable(icp):  # e.g. a wrapper around o3d.registration_icp: (item, T_init) -> T
            for i, it in enumerate(its):
                refined = np.asarray(icp(it, T_h[i]), dtype=np.float32)
                r1, t1 = ops.transformation_error_batched(torch.from_numpy(refined[None]).to(device), T_gt_d[i:i + 1].contiguous())
                if r1.item() <= re_h[i] and t1.item() <= te_h[i]:
                    T_h[i], re_h[i], te_h[i] = refined, r1.item(), t1.item()
        keep.clear()
        return T_h, T_gt.numpy(), re_h, te_h, loss_h

    return finish


def register_items(net, its: Sequence[tuple], centers: Sequence[torch.Tensor], pair_ids: Sequence[int],
                   corr: str = "tgt", dis_thresh: float = 0.1, icp=None, icp_dist: float = ICP_MAX_CORR_DIST,
                   icp_iters: int = ICP_MAX_ITER, device: Optional[torch.device] = None,
                   pred_hook: Optional[Callable] = None, lanes: Optional[int] = None):
    """A1-A10 (+ optional ICP) for one batch.  its[i] = (src, tgt, rot, trans, s, c) normalised fp32 CPU tensors;
    centers[i] = the src_center the evaluator passes to the model.  Returns host arrays
    (T [B,4,4], T_gt [B,4,4], re [B], te [B], loss [B]).  The batch runs as `lanes` concurrent sub-batches of pairs
    (scream_amd/lanes.py; default 2 from 8 pairs up) -- per-pair results do not depend on the split."""
    return register_items_async(net, its, centers, pair_ids, corr, dis_thresh, icp, icp_dist, icp_iters, device,
                                pred_hook, lanes)()


def evaluate_items_async(net, items: Sequence[tuple], pair_ids: Sequence[int], corr: str = "tgt", dis_thresh: float = 0.1,
                         icp=None, device: Optional[torch.device] = None, pred_hook: Optional[Callable] = None,
                         stream: Optional[torch.cuda.Stream] = None):
    """Enqueue one 3DMatch-family batch (items are the reference's 9-tuples); returns ``finish() -> rows [B, 8]``."""
    its = [_strip(it) for it in items]
    core = [(it[0], it[1], it[2], it[3], it[4], it[7]) for it in its]
    centers = [it[3] for it in its]  # src_center = trans^T, evaluate_3d_match.py:84
    fin = register_items_async(net, core, centers, pair_ids, corr, dis_thresh, icp, device=device, pred_hook=pred_hook,
                               stream=stream)

    def finish():
        T_h, T_gt, re_h, te_h, loss = fin()
        rows = np.zeros((len(its), sdist.ROW_WIDTH), dtype=np.float64)
        for i, it in enumerate(its):
            rmse = math.sqrt(max(RMSE(np.linalg.inv(T_gt[i]) @ T_h[i], it[6]), 0.0))  # evaluate_3d_match.py:122
            rows[i] = [pair_ids[i], it[8], float(abs(it[5][1] - it[5][0]) > 1), float(rmse < 0.2), re_h[i], te_h[i], rmse, loss[i]]
        return rows

    return finish


def evaluate_items(net, items: Sequence[tuple], pair_ids: Sequence[int], corr: str = "tgt", dis_thresh: float = 0.1,
                   icp=None, device: Optional[torch.device] = None, pred_hook: Optional[Callable] = None) -> np.ndarray:
    """One 3DMatch-family batch: items are the reference's 9-tuples.  Returns metric rows [B, 8] (dist.ROW_WIDTH).
    pred_hook(batch, src_pred, pair_ids) -> src_pred may replace the network's prediction (used for the
    "registered src + noise" throughput/metric variant of SURVEY.md section 8d and by the parity tests)."""
    return evaluate_items_async(net, items, pair_ids, corr, dis_thresh, icp, device, pred_hook)()


def evaluate_loader(net, loader: Iterable, corr: str = "tgt", dis_thresh: float = 0.1,
                    re_static_method: str = "median", batch_pairs: int = 32, icp="gpu",
                    verbose: bool = True, pred_hook: Optional[Callable] = None, num_workers: int = 0,
                    worker_context: Optional[str] = None, in_flight: int = IN_FLIGHT):
    """evaluate_3d_match.py:53-171.  ``loader`` is a dataset or DataLoader of the reference's 9-tuples.
    ``icp="gpu"`` (default) refines every pose like evaluate_3d_match.py:106-119 does; ``icp=None`` skips it.
    With torch.distributed initialised the pairs are sharded round-robin over ranks and the per-pair rows
    are all-gathered once at the end; every rank returns the same (point_trans_loss, rre, rte, rr)."""
    dataset = getattr(loader, "dataset", loader)
    n = len(dataset)
    rank, world = sdist.rank_world()
    mine = sdist.shard_indices(n, rank, world)
    rows: List[np.ndarray] = []
    done = 0
    # var-len batches are lists of items; worker processes overlap file I/O + normalisation (or synthetic generation) with
    # the GPU work of the previous batch.  NOT pin_memory=True: the batch is repacked on the host into this module's own
    # pinned staging buffers anyway (PackedBatch.from_host), and reading a batch back out of the loader's pinned copy --
    # page-locked memory that the CPU reads uncached on this platform -- cost more than the file I/O it overlapped
    # (690 pairs/s with workers against 1 330 without, tools/eval_e2e.py, until round 3).
    # Worker processes come from a FORK SERVER by default (worker_context=None): workers forked from this process -- one
    # that has the GPU runtime initialised and gigabytes mapped -- delivered 650 pairs/s whatever their number, workers from a
    # clean fork server 1 300 (tools/eval_e2e.py); the dataset must then be picklable, as the reference's module-level dataset
    # classes are.  "fork" restores the torch default.
    ctx = None
    if num_workers > 0:
        ctx = worker_context or ("forkserver" if torch.cuda.is_available() and torch.cuda.is_initialized() else None)
    batches = torch.utils.data.DataLoader(torch.utils.data.Subset(dataset, mine), batch_size=batch_pairs, shuffle=False,
                                          collate_fn=collate_pairs, num_workers=num_workers, pin_memory=False,
                                          multiprocessing_context=ctx)
    def collect(finish, n_done):
        r = finish()
        rows.append(r)
        if verbose and rank == 0:
            print("\r%s  re: %.5f  te: %.5f  rmse: %.5f  rr: %.5f" % (
                processbar(n_done, len(mine)), r[-1, sdist.COL_RE], r[-1, sdist.COL_TE], r[-1, sdist.COL_RMSE],
                float(np.concatenate(rows)[:, sdist.COL_SUCCESS].mean())), end="")

    # Batch i is enqueued before the host side of batch i-1 runs: the GPU never waits for the host.  Consecutive batches go to
    # alternating HIP streams, each batch whole (one packed forward of all its pairs): several batches are then in flight
    # out of phase, and the launch-bound end of one -- the ICP loop is ~60 tiny dependent launches -- runs beside the
    # others' forwards instead of beside nothing (two lanes INSIDE a batch finish together and both sit in their ICP loops
    # at the same time: 1 373 vs 1 570 pairs/s with / without ICP; results are the same either way, pairs never interact).
    dev = next(net.parameters()).device if hasattr(net, "parameters") else None
    # in_flight batches are enqueued before the oldest is collected, each on its own stream (a stream, its staging buffers and
    # its workspace are reused only after the batch that last used them has been collected).
    in_flight = max(2, int(in_flight))
    streams = _lanes.lane_streams(dev, in_flight) if dev is not None and dev.type == "cuda" else [None] * in_flight
    pending, k = [], 0
    for items in batches:
        items = unpack_batch(items)  # (worker processes send a batch as three flat tensors, scream_amd/data.py)
        ids = mine[done:done + len(items)]
        fin = evaluate_items_async(net, items, ids, corr, dis_thresh, icp, pred_hook=pred_hook, stream=streams[k % in_flight])
        k += 1
        done += len(ids)
        pending.append((fin, done))
        if len(pending) >= in_flight:
            collect(*pending.pop(0))
    for pd in pending:
        collect(*pd)
    local = np.concatenate(rows) if rows else np.zeros((0, sdist.ROW_WIDTH))
    allrows = sdist.all_gather_rows(local)
    out = aggregate_rows(allrows, re_static_method)
    if verbose and rank == 0:
        print("\nmean: loss: %.5f  rre: %.5f  rte: %.5f  rr: %.5f" % out)
    return out


def evaluate_3d_match(net, dataset, dis_thresh: float = 0.1, **kw):
    """evaluate_3d_match.py:178-179."""
    return evaluate_loader(net, dataset, dis_thresh=dis_thresh, **kw)


def evaluate_3d_lo_match(net, dataset, dis_thresh: float = 0.1, **kw):
    """evaluate_3d_match.py:174-175."""
    return evaluate_loader(net, dataset, dis_thresh=dis_thresh, **kw)


def evaluate_3d_zero_match(net, dataset, dis_thresh: float = 0.1, **kw):
    """evaluate_3d_match.py:182-183: correspondences against src_pred itself, per-scene mean."""
    return evaluate_loader(net, dataset, corr="src_pred", dis_thresh=dis_thresh, re_static_method="mean", **kw)
