#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code on CPU.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  The reference sources are imported from where they lie; nothing of them is
copied.  Three third-party imports the hot path never touches (cv2, open3d, igraph --
utils.py:2,6, models/render.py:5) are absent from this image, so empty placeholder
modules are registered for the import to succeed, and the (eval-unused, CUDA-only)
RegistrationRender is rebound to a parameter-less no-op.

Weights are not stored: they are regenerated from a numpy seed by
scream_amd.synthetic.make_state_dict (PCG64, machine independent) and loaded into the
reference model with load_state_dict, so a fixture is (seed, inputs, expected outputs).

Usage:  python oracle/make_golden.py   (rewrites tests/golden/)
"""
import os
import pickle
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SCREAM_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def import_reference():
    for name in ("cv2", "open3d", "igraph"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import models.pointnet as ref_pointnet  # noqa
    import models.transformer as ref_transformer  # noqa
    import utils as ref_utils  # noqa
    import lie.torch.so3_common as ref_so3  # noqa

    class _NoRender(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    ref_pointnet.RegistrationRender = _NoRender
    sys.path.remove(REF)
    return ref_pointnet, ref_transformer, ref_utils, ref_so3


def cloud(rng, n, spread=0.6):
    return (rng.uniform(-spread, spread, size=(1, n, 3))).astype(np.float32)


def main():
    from scream_amd.synthetic import make_state_dict

    torch.set_num_threads(8)
    ref_pointnet, ref_tf, ref_utils, ref_so3 = import_reference()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)

    # ---- A1: position embedding ------------------------------------------------
    xyz = cloud(rng, 32, 1.0)
    out = {"xyz": xyz}
    for d in (64, 256):
        out["pe_%d" % d] = ref_tf.PositionEmbeddingCoordsSine(3, d)(torch.from_numpy(xyz)).numpy()
    np.savez(os.path.join(OUT, "pe.npz"), **out)

    # ---- A3: linear attention ---------------------------------------------------
    q = rng.normal(size=(1, 40, 8, 32)).astype(np.float32)
    k = rng.normal(size=(1, 56, 8, 32)).astype(np.float32)
    v = rng.normal(size=(1, 56, 8, 32)).astype(np.float32)
    la = ref_tf.LinearAttention()(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v))
    np.savez(os.path.join(OUT, "linattn.npz"), q=q, k=k, v=v, out=la.numpy())

    # ---- A2-A4: one MHAttention, self and cross, every intermediate --------------
    sd = make_state_dict(11, 256, 1, 1)
    mha = ref_tf.MHAttention(256, 8)
    mha.load_state_dict({k_[len("stem.0."):]: v_ for k_, v_ in sd.items() if k_.startswith("stem.0.")})
    mha.eval()
    xq = rng.normal(size=(1, 24, 256)).astype(np.float32)
    xk = rng.normal(size=(1, 40, 256)).astype(np.float32)
    rec = {"xq": xq, "xk": xk, "seed": np.int64(11)}
    with torch.no_grad():
        for tag, (a, b) in {"self": (xq, xq), "cross": (xq, xk)}.items():
            ta, tb = torch.from_numpy(a), torch.from_numpy(b)
            qq = mha.q_proj(ta).view(1, -1, 8, 32)
            kk = mha.k_proj(tb).view(1, -1, 8, 32)
            vv = mha.v_proj(tb).view(1, -1, 8, 32)
            att = mha.attention(qq, kk, vv)
            msg = mha.merge(att.view(1, -1, 256))
            m1 = mha.norm1(msg + ta)
            ffn = mha.mlp(m1)
            y = mha.norm2(ta + ffn)
            assert torch.equal(y, mha(ta, tb, tb))
            for nm, t in dict(q=qq, k=kk, v=vv, att=att, msg=msg, m1=m1, ffn=ffn, out=y).items():
                rec["%s_%s" % (tag, nm)] = t.numpy()
    np.savez(os.path.join(OUT, "mha.npz"), **rec)

    # ---- A5/A6 end-to-end PointTransformer, d_model=256 ---------------------------
    cases = [(21, 1, 1, 96, 128, True), (22, 2, 2, 130, 70, True), (23, 6, 6, 200, 257, True),
             (24, 1, 1, 1, 1, True), (25, 1, 1, 3, 300, True), (26, 2, 1, 150, 129, False)]
    rec = {"cases": np.array([c[:5] + (int(c[5]),) for c in cases], dtype=np.int64)}
    e2e = {}
    for seed, ns, nc, n, m, explicit_center in cases:
        sd = make_state_dict(seed, 256, ns, nc)
        net = ref_pointnet.PointTransformer(d_model=256, self_layer_num=ns, cross_layer_num=nc)
        missing = net.load_state_dict(sd, strict=True)
        net.eval()
        r2 = np.random.default_rng(seed)
        src, tgt = cloud(r2, n), cloud(r2, m)
        center = (r2.uniform(-0.3, 0.3, size=(1, 1, 3))).astype(np.float32)
        with torch.no_grad():
            src_, imgs, tr = net(torch.from_numpy(src), torch.from_numpy(tgt),
                                 torch.from_numpy(center) if explicit_center else None, 1.0, False, False, None)
        assert imgs is None and tr is None
        rec["src_%d" % seed], rec["tgt_%d" % seed], rec["center_%d" % seed] = src, tgt, center
        rec["out_%d" % seed] = src_.numpy()
        e2e[seed] = (src_, torch.from_numpy(tgt))
    np.savez(os.path.join(OUT, "e2e.npz"), **rec)

    # ---- DEMTransformer, d_model=256 ---------------------------------------------
    rec = {}
    dem_cases = [(31, 1, 1, 120, 90), (32, 2, 2, 200, 257)]
    rec["cases"] = np.array(dem_cases, dtype=np.int64)
    for seed, ns, nc, n, m in dem_cases:
        sd = make_state_dict(seed, 256, ns, nc, dem=True)
        net = ref_pointnet.DEMTransformer(d_model=256, self_layer_num=ns, cross_layer_num=nc)
        net.load_state_dict(sd, strict=True)
        net.eval()
        r2 = np.random.default_rng(seed)
        dsm, demc = cloud(r2, n), cloud(r2, m)
        with torch.no_grad():
            out, imgs = net(torch.from_numpy(dsm), torch.from_numpy(demc), False)
        assert imgs is None
        rec["dsm_%d" % seed], rec["dem_%d" % seed], rec["out_%d" % seed] = dsm, demc, out.numpy()
    np.savez(os.path.join(OUT, "dem.npz"), **rec)

    # ---- A7: thresholded 1-NN -----------------------------------------------------
    rec = {}
    src_, tgt = e2e[23]
    svals = np.array([1.0, 0.3137, 2.5], dtype=np.float64)
    rec["e2e_s"] = svals
    for i, s in enumerate(svals):
        dist = ref_utils.square_distance(src_ / float(s), tgt / float(s))[0]
        d, idx = dist.min(dim=1)
        second = dist.clone()
        second[torch.arange(dist.shape[0]), idx] = float("inf")
        rec["e2e_d_%d" % i], rec["e2e_idx_%d" % i] = d.numpy(), idx.numpy()
        rec["e2e_d2_%d" % i] = second.min(dim=1)[0].numpy()
    r2 = np.random.default_rng(77)
    tgt_big = (r2.uniform(-0.8, 0.8, size=(1, 900, 3))).astype(np.float32)
    tgt_big[0, 450:460] = tgt_big[0, 100:110]  # exact duplicates -> ties resolved to the lowest index
    pick = r2.integers(0, 900, size=700)
    src_big = tgt_big[:, pick] + r2.normal(scale=0.004, size=(1, 700, 3)).astype(np.float32)
    src_big[0, :40] = tgt_big[0, 450:490]  # exact hits on duplicated / plain points
    s_big = 0.4211
    dist = ref_utils.square_distance(torch.from_numpy(src_big) / s_big, torch.from_numpy(tgt_big) / s_big)[0]
    d, idx = dist.min(dim=1)
    rec.update(big_src=src_big, big_tgt=tgt_big, big_s=np.float64(s_big), big_d=d.numpy(), big_idx=idx.numpy(),
               big_valid=(d < 0.1).numpy())
    np.savez(os.path.join(OUT, "nn.npz"), **rec)

    # ---- A9: rigid_transform_3d ----------------------------------------------------
    with open(os.path.join(REF, "datasets/3DMatch/indoor/3DMatch.pkl"), "rb") as f:
        info = pickle.load(f)
    with open(os.path.join(REF, "datasets/3DMatch/indoor/3DLoMatch.pkl"), "rb") as f:
        info_lo = pickle.load(f)
    rots = np.stack([info["rot"][i] for i in (0, 7, 300, 901)] + [info_lo["rot"][i] for i in (5, 1200)]).astype(np.float64)
    trs = np.stack([info["trans"][i] for i in (0, 7, 300, 901)] + [info_lo["trans"][i] for i in (5, 1200)]).astype(np.float64)
    rec = {"gt_rot": rots, "gt_trans": trs}
    r2 = np.random.default_rng(5)
    kab = []

    def add(name, A, B, w=None, thr=0):
        tw = None if w is None else torch.from_numpy(w.copy())
        T = ref_utils.rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(B), tw, thr)
        rec[name + "_A"], rec[name + "_B"], rec[name + "_T"] = A, B, T.numpy()
        if w is not None:
            rec[name + "_w"], rec[name + "_thr"] = w, np.float64(thr)
        kab.append(name)

    for i in range(6):
        A = r2.uniform(-2, 2, size=(1, 500, 3)).astype(np.float32)
        B = (A[0].astype(np.float64) @ rots[i].T + trs[i].T)[None].astype(np.float32)
        add("exact%d" % i, A, B)
        add("noisy%d" % i, A, B + r2.normal(scale=0.02, size=B.shape).astype(np.float32))
    # planar (rank-2) sets incl. a mirrored target that triggers the det(V U^T) = -1 branch (utils.py:171-174)
    P = np.concatenate([r2.uniform(-1, 1, size=(1, 300, 2)), np.zeros((1, 300, 1))], axis=2).astype(np.float32)
    add("planar", P, (P[0].astype(np.float64) @ rots[1].T + trs[1].T)[None].astype(np.float32))
    A = r2.uniform(-1, 1, size=(1, 200, 3)).astype(np.float32)
    add("mirror", A, A * np.array([1, 1, -1], dtype=np.float32) + r2.normal(scale=0.01, size=A.shape).astype(np.float32))
    add("k0", np.zeros((1, 0, 3), np.float32), np.zeros((1, 0, 3), np.float32))
    A3 = r2.uniform(-1, 1, size=(1, 3, 3)).astype(np.float32)
    add("k3", A3, (A3[0].astype(np.float64) @ rots[2].T + trs[2].T)[None].astype(np.float32))
    A = r2.uniform(-2, 2, size=(2, 400, 3)).astype(np.float32)  # batched + weights + threshold
    B = np.stack([(A[0].astype(np.float64) @ rots[3].T + trs[3].T), (A[1].astype(np.float64) @ rots[4].T + trs[4].T)]).astype(np.float32)
    B[:, 300:] += r2.normal(scale=0.5, size=(2, 100, 3)).astype(np.float32)  # outliers get low weight
    w = r2.uniform(0.2, 1, size=(2, 400)).astype(np.float32)
    w[:, 300:] = r2.uniform(0, 0.1, size=(2, 100)).astype(np.float32)
    add("weighted", A, B, w, 0.15)
    rec["names"] = np.array(kab)
    np.savez(os.path.join(OUT, "kabsch.npz"), **rec)

    # ---- A10: transformation_error; rotmat2quat (lie/) -------------------------------
    poses = []
    for i in range(6):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3:] = rots[i], trs[i]
        poses.append(T)
    Rpi = np.diag([1.0, -1.0, -1.0]).astype(np.float32)
    Tpi = np.eye(4, dtype=np.float32)
    Tpi[:3, :3] = Rpi
    poses += [np.eye(4, dtype=np.float32), Tpi]
    # a near-pi rotation exercising the branch of rotmat2quat (lie/torch/so3_common.py:113-125)
    ax = np.array([0.6, 0.0, 0.8])
    Rnp = (2 * np.outer(ax, ax) - np.eye(3)).astype(np.float32)
    Tnp = np.eye(4, dtype=np.float32)
    Tnp[:3, :3] = Rnp
    poses.append(Tnp)
    poses = np.stack(poses)
    re = np.zeros((len(poses), len(poses)), np.float32)
    te = np.zeros_like(re)
    for i in range(len(poses)):
        for j in range(len(poses)):
            a, b = ref_utils.transformation_error(torch.from_numpy(poses[i]), torch.from_numpy(poses[j]))
            re[i, j], te[i, j] = a.item(), b.item()
    quat = ref_so3.rotmat2quat(torch.from_numpy(poses[:, :3, :3].copy())).numpy()
    np.savez(os.path.join(OUT, "pose_metrics.npz"), poses=poses, re=re, te=te, quat=quat)

    # real 6x6 information matrices (benchmark metadata files, data not code) for the RMSE metric
    infos = []
    for scene in ("7-scenes-redkitchen", "sun3d-hotel_uc-scan3"):
        lines = [l.strip() for l in open(os.path.join(REF, "datasets/3DMatch/info/3DMatch", scene, "gt.info"))]
        for p in range(3):
            infos.append(np.array([lines[p * 7 + j].split() for j in range(1, 7)], dtype=np.float32))
    np.savez(os.path.join(OUT, "info.npz"), info=np.stack(infos))
    dataset_items(rng)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print("  %-20s %8d B" % (fn, os.path.getsize(os.path.join(OUT, fn))))


def dataset_items(rng):
    """A12: the reference's OWN test-set items.  ThreeDMatchTest / ThreeDLoMatchTest / ThreeDZeroMatchTest
    (datasets/three_d_match.py:219-294) and KITTI_Test (datasets/kitti.py:328-350, norm_pc :268-273) read
    '<split>/src%d.npy' ... relative to the working directory, so small seeded pairs are written in that on-disk layout
    (process_3d_match.py:38-40,199-200; process_kitti.py:72-74) into a temporary directory, the reference classes are
    run there, and raw inputs + the 9-/6-tuples they return are stored.  Poses and 6x6 information matrices are real
    benchmark metadata (datasets/3DMatch/indoor/*.pkl, info/**/gt.info: data files)."""
    import tempfile
    for name in ("cv2", "open3d", "igraph"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    import datasets.kitti as ref_kitti
    import datasets.three_d_match as ref_3dm
    sys.path.remove(REF)
    with open(os.path.join(REF, "datasets/3DMatch/indoor/3DMatch.pkl"), "rb") as f:
        meta = pickle.load(f)
    scenes = sorted(ref_3dm.scene_name_to_idx)
    infos = np.load(os.path.join(OUT, "info.npz"))["info"]
    rec = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            k = 0
            for split, cls in (("3DMatch_test", ref_3dm.ThreeDMatchTest), ("3DLoMatch_test", ref_3dm.ThreeDLoMatchTest),
                               ("3DZeroMatch_test", ref_3dm.ThreeDZeroMatchTest)):
                os.makedirs(os.path.join(split, "info"))
                names = []
                for i in range(3):
                    n, m = int(rng.integers(150, 400)), int(rng.integers(150, 400))
                    T = np.eye(4)
                    T[:3, :3], T[:3, 3:] = meta["rot"][37 * k + 5], meta["trans"][37 * k + 5]
                    tgt = rng.uniform(-1.5, 2.5, size=(m, 3))
                    src = (rng.uniform(-1.5, 2.5, size=(n, 3)) - T[:3, 3]) @ T[:3, :3]  # R src + t lies in the same box
                    idx = np.array([int(rng.integers(0, 30)), int(rng.integers(0, 30))], dtype=np.int64)
                    cov = infos[k % len(infos)]
                    np.save("%s/src%d.npy" % (split, i), src)
                    np.save("%s/tgt%d.npy" % (split, i), tgt)
                    np.save("%s/T%d.npy" % (split, i), T)
                    np.save("%s/info/idx%d.npy" % (split, i), idx)
                    np.save("%s/info/covariance%d.npy" % (split, i), cov)
                    names.append(scenes[(3 * k + 1) % 8])
                    pre = "%s_%d_" % (split, i)
                    rec.update({pre + "src": src, pre + "tgt": tgt, pre + "T": T, pre + "idx": idx, pre + "cov": cov,
                                pre + "scene_name": np.array(names[-1])})
                    k += 1
                with open(os.path.join(split, "info", "scene_names.txt"), "w") as f:
                    f.writelines(nm + "\n" for nm in names)
                ds = cls()
                for i in range(3):
                    item = ds[i]
                    pre = "%s_%d_out_" % (split, i)
                    for key, val in zip(("src", "tgt", "rot", "trans", "s", "idx", "cov", "c", "scene"), item):
                        rec[pre + key] = val.numpy() if torch.is_tensor(val) else np.asarray(val)
            os.makedirs("KITTI_test")
            for i in range(3):
                n, m = int(rng.integers(200, 500)), int(rng.integers(200, 500))
                yaw = float(rng.uniform(-0.2, 0.2))
                T = np.eye(4)
                T[:3, :3] = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
                T[:3, 3] = [rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-0.5, 0.5)]
                box = np.array([60.0, 45.0, 4.0])
                tgt = rng.uniform(-1, 1, size=(m, 3)) * box
                src = (rng.uniform(-1, 1, size=(n, 3)) * box - T[:3, 3]) @ T[:3, :3]
                np.save("KITTI_test/src%d.npy" % i, src)
                np.save("KITTI_test/tgt%d.npy" % i, tgt)
                np.save("KITTI_test/T%d.npy" % i, T)
                pre = "KITTI_test_%d_" % i
                rec.update({pre + "src": src, pre + "tgt": tgt, pre + "T": T})
                item = ref_kitti.KITTI_Test()[i]
                for key, val in zip(("src", "tgt", "rot", "trans", "s", "c"), item):
                    rec[pre + "out_" + key] = val.numpy() if torch.is_tensor(val) else np.asarray(val)
                c_ref, s_ref = ref_kitti.norm_pc(np.concatenate([(T[:3, :3] @ src.T + T[:3, 3:]).T, tgt], axis=0))
                rec[pre + "norm_pc_c"], rec[pre + "norm_pc_s"] = np.asarray(c_ref), np.asarray(s_ref)
        finally:
            os.chdir(cwd)
    np.savez(os.path.join(OUT, "dataset_items.npz"), **rec)


if __name__ == "__main__":
    if "--dataset-items-only" in sys.argv:  # adds tests/golden/dataset_items.npz without regenerating the other fixtures
        dataset_items(np.random.default_rng(20261004))
    else:
        main()
