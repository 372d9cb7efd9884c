"""Host-side logic that needs no GPU: the C-ABI surface, packing, metric aggregation, sharding (gloo,
world size 2), synthetic data, and the rule that the product never falls back to a CPU path."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import scream_ref as O
from scream_amd import _lib
from scream_amd import dist as sdist
from scream_amd.synthetic import make_3dmatch_pair, make_state_dict, make_uniform_pair, state_dict_keys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------ C ABI
def _header_functions():
    text = open(os.path.join(REPO, "include", "scream_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scream_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # builds with hipcc if needed; no GPU required to load
    names = _header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), "libscream_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.scream_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.scream_version()


def test_host_side_argument_checks_need_no_gpu():
    lib = _lib.load()
    unfused, fused = lib.scream_forward_workspace_bytes(128, 256, 1, 1, 0, 0), lib.scream_forward_workspace_bytes(128, 256, 1, 1, 1, 0)
    assert unfused > 256 * (256 * 5 + 1024) * 4  # x0, x1, q, att, m1 + hidden
    assert 256 * 256 * 3 * 4 < fused <= unfused - 256 * (256 * 2 + 1024) * 4  # the fused tail needs x0, x1, q only
    assert lib.scream_forward_workspace_bytes(256, 128, 1, 1, 1, 0) == -1  # rows_total < rows_src
    # six cross layers' target-side K^T V partials (one 128-row target tile x 8 heads x 1056 floats) and images side by side
    assert lib.scream_forward_workspace_bytes(128, 256, 1, 1, 1, 6) - fused == 6 * (8 * 1056 * 4 + lib.scream_kv_image_bytes())
    # the split entry points validate `split` and the fp16 exponents on the host
    assert [lib.scream_tail_image_bytes(k, 0) for k in (1, 2, 3)] == [72 * 16 * 1024, 72 * 32 * 1024, 72 * 48 * 1024]
    assert [lib.scream_tail_image_bytes(k, 1) for k in (1, 2, 3)] == [80 * 16 * 1024, 80 * 32 * 1024, -1]  # the next layer's query stages: fp16 splits
    assert lib.scream_tail_image_bytes(4, 0) == -1
    # NULL pointers / bad shapes are rejected before any launch
    assert lib.scream_gemm_f32(None, 256, None, None, 256, 128, 256, 256, 0, 0, None, None, 0, None, None, None) == -1
    assert lib.scream_nn_search(*([None] * 7), 1, 1, 1, 128, 128, 0.1, *([None] * 6)) == -1


def test_product_code_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "scream_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    for f in ("models/pointnet.py", "utils.py", "evaluate_3d_match.py"):
        src = open(os.path.join(REPO, f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_no_cpu_fallback_product_path_fails_loudly():
    from scream_amd import ops
    from scream_amd.model import PointTransformer
    net = PointTransformer(256, 1, 1).eval()
    with pytest.raises(_lib.ScreamHipError):
        net(torch.zeros(1, 10, 3), torch.zeros(1, 12, 3))
    with pytest.raises(_lib.ScreamHipError):
        ops.gemm_f32(torch.zeros(128, 256), torch.zeros(256, 256))
    with pytest.raises(NotImplementedError):
        PointTransformer(d_model=64)


# ---------------------------------------------------------------------------- model surface
def test_state_dict_layout_matches_reference_names():
    from scream_amd.model import PointTransformer
    net = PointTransformer(256, 6, 6)
    want = state_dict_keys(256, 6, 6)
    got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert got == [(k, tuple(s)) for k, s in want]
    net.load_state_dict(make_state_dict(1, 256, 6, 6), strict=True)


def test_dem_transformer_state_dict_layout():
    from scream_amd.model import DEMTransformer
    from scream_amd.synthetic import dem_state_dict_keys
    net = DEMTransformer(256, 2, 1)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(s)) for k, s in dem_state_dict_keys(256, 2, 1)]
    net.load_state_dict(make_state_dict(1, 256, 2, 1, dem=True), strict=True)
    with pytest.raises(_lib.ScreamHipError):
        net(torch.zeros(1, 5, 3), torch.zeros(1, 5, 3))


def test_loss_matches_oracle():
    from scream_amd.model import PointTransformer
    net = PointTransformer(256, 1, 1)
    g = torch.Generator().manual_seed(0)
    sp, src = torch.randn(1, 50, 3, generator=g), torch.randn(1, 50, 3, generator=g)
    rot, tr = torch.linalg.qr(torch.randn(1, 3, 3, generator=g))[0], torch.randn(1, 3, 1, generator=g)
    torch.testing.assert_close(net.loss(sp, src, rot, tr), O.point_loss(sp, src, rot, tr))


# -------------------------------------------------------------------------------- packing
def test_packed_layout():
    from scream_amd.packing import PackedBatch
    srcs = [torch.rand(n, 3) for n in (200, 1, 129)]
    tgts = [torch.rand(m, 3) for m in (130, 300, 128)]
    b = PackedBatch.from_pairs(srcs, tgts, [None, torch.tensor([1.0, 2.0, 3.0]), None])
    assert b.rows_src == 256 + 128 + 256 and b.rows_total == b.rows_src + 256 + 384 + 128
    assert b.cloud_row0_host.tolist() == [0, 256, 384, 640, 896, 1280]
    assert b.cloud_len_host.tolist() == [200, 1, 129, 130, 300, 128]
    assert b.max_chunks == 2 and b.tile_cloud.tolist() == [0, 0, 1, 2, 2, 3, 3, 4, 4, 4, 5]
    torch.testing.assert_close(b.xyz[256:257], srcs[1])
    assert float(b.xyz[257:384].abs().sum()) == 0.0  # zero padding
    torch.testing.assert_close(b.center[0], srcs[0].mean(dim=0))  # default centre = mean (pointnet.py:43-44)
    torch.testing.assert_close(b.center[1], torch.tensor([1.0, 2.0, 3.0]))
    assert float(b.center[3:].abs().sum()) == 0.0
    parts = b.unpack_src(b.xyz[: b.rows_src])
    assert [p.shape[0] for p in parts] == [200, 1, 129] and torch.equal(parts[2], srcs[2])
    with pytest.raises(ValueError):
        PackedBatch.from_pairs([torch.rand(0, 3)], [torch.rand(5, 3)])


# ------------------------------------------------------------------------ metrics (A11)
def test_mat2quat_and_rmse_vs_reference_vectors(golden):
    from scream_amd.evaluate import RMSE, mat2quat
    g = golden("pose_metrics")
    info = golden("info")["info"]
    for i, P in enumerate(g["poses"]):
        q = mat2quat(P[:3, :3])
        np.testing.assert_allclose(q, g["quat"][i], atol=1e-6)  # lie/torch rotmat2quat, incl. the near-pi case
        assert q[0] >= -1e-12
        for j in (0, 3):
            er = np.linalg.inv(g["poses"][j].astype(np.float64)) @ P.astype(np.float64)
            np.testing.assert_allclose(RMSE(er, info[i % len(info)]), O.rmse_metric(er, info[i % len(info)]), rtol=1e-9, atol=1e-12)


def test_gt_pose_metric_matches_oracle():
    from scream_amd.evaluate import gt_pose_metric
    src, tgt, T, *_ = make_3dmatch_pair(3)
    s_, t_, rot, trans, s, c = O.normalize_pair(src, tgt, T)
    torch.testing.assert_close(gt_pose_metric(rot, trans, s, c), O.gt_pose_metric(rot, trans, s, c), rtol=0, atol=0)
    # the metric-frame GT pose is the original T up to fp32 rounding
    np.testing.assert_allclose(gt_pose_metric(rot, trans, s, c).numpy(), T, atol=2e-5)


def _reference_style_aggregate(rows, method):
    """evaluate_3d_match.py:128-138,152-169 written as the per-scene list bookkeeping it is."""
    metric = {sc: [[], [], 0, 0] for sc in range(8)}
    for r in rows:
        if r[sdist.COL_COUNTED] > 0:
            m = metric[int(r[sdist.COL_SCENE])]
            m[3] += 1
            if r[sdist.COL_SUCCESS] > 0:
                m[2] += 1
                m[0].append(r[sdist.COL_RE]); m[1].append(r[sdist.COL_TE])
            else:
                m[0].append(0); m[1].append(0)
    f = np.median if method == "median" else np.mean
    used = [m for m in metric.values() if m[3] > 0]
    return (sum(f(m[0]) for m in used) / len(used), sum(f(m[1]) for m in used) / len(used),
            sum(m[2] / m[3] for m in used) / len(used))


def test_aggregate_rows():
    from scream_amd.evaluate import aggregate_rows
    rng = np.random.default_rng(0)
    n = 200
    rows = np.zeros((n, sdist.ROW_WIDTH))
    rows[:, sdist.COL_PAIR] = np.arange(n)
    rows[:, sdist.COL_SCENE] = rng.integers(0, 7, size=n)  # scene 7 stays empty: skipped, not a ZeroDivisionError
    rows[:, sdist.COL_COUNTED] = rng.random(n) < 0.8
    rows[:, sdist.COL_SUCCESS] = rng.random(n) < 0.6
    rows[:, sdist.COL_RE], rows[:, sdist.COL_TE] = rng.random(n) * 5, rng.random(n) * 0.2
    rows[:, sdist.COL_LOSS] = rng.random(n)
    for method in ("median", "mean"):
        loss, rre, rte, rr = aggregate_rows(rows, method)
        w = _reference_style_aggregate(rows, method)
        np.testing.assert_allclose([rre, rte, rr], w, rtol=1e-12)
        np.testing.assert_allclose(loss, rows[:, sdist.COL_LOSS].mean())


# ------------------------------------------------------------------------- sharding (gloo)
WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import torch, torch.distributed as tdist
from scream_amd import dist as sdist
from scream_amd import evaluate as ev
rank, world, _ = sdist.init_from_env("gloo")
assert (rank, world) == sdist.rank_world() and world == 2
n = 11  # odd: ranks hold 6 and 5 rows -> exercises the padding path

class DS:
    def __len__(self): return n
    def __getitem__(self, i): return i

def fake_items(net, items, ids, corr, dis_thresh, icp, device=None, pred_hook=None, stream=None):
    rows = np.zeros((len(ids), sdist.ROW_WIDTH))
    for k, i in enumerate(ids):
        rows[k] = [i, i %% 8, 1.0, float(i %% 3 != 0), 1.0 + i, 0.01 * i, 0.1, 0.5 * i]
    return rows
ev.evaluate_items_async = lambda *a, **k: (lambda: fake_items(*a, **k))
out = ev.evaluate_loader(None, DS(), batch_pairs=4, verbose=False)
allrows = fake_items(None, None, list(range(n)), None, None, None)
want = ev.aggregate_rows(allrows, "median")
assert np.allclose(out, want), (out, want)
mine = sdist.shard_indices(n, rank, world)
assert mine == list(range(rank, n, world))
g = sdist.all_gather_rows(fake_items(None, None, mine, None, None, None))
assert g.shape == (n, sdist.ROW_WIDTH) and (g[:, 0] == np.arange(n)).all()
tdist.barrier(); tdist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_evaluation_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_single_process_gather_is_identity_sorted():
    rows = np.zeros((3, sdist.ROW_WIDTH))
    rows[:, 0] = [2, 0, 1]
    assert sdist.all_gather_rows(rows)[:, 0].tolist() == [0, 1, 2]
    assert sdist.rank_world() == (0, 1)


# --------------------------------------------------------------------------- synthetic data
def test_synthetic_is_deterministic_and_sized():
    a, b = make_state_dict(7, 256, 1, 1), make_state_dict(7, 256, 1, 1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert abs(float(a["stem.0.q_proj.weight"][0, 0]) - float(make_state_dict(8, 256, 1, 1)["stem.0.q_proj.weight"][0, 0])) > 0
    src, tgt, T, idx, cov, scene = make_3dmatch_pair(5)
    src2 = make_3dmatch_pair(5)[0]
    assert np.array_equal(src, src2) and 3000 < len(src) < 8000 and 3000 < len(tgt) < 8000
    assert cov.shape == (6, 6) and np.all(np.linalg.eigvalsh(cov.astype(np.float64)) > 0) and idx.tolist() == [0, 2]
    np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
    s, t, _ = make_uniform_pair(0, 1000)
    assert s.shape == (1000, 3) and np.linalg.norm(t, axis=1).max() <= 1.0


def test_normalize_pair_matches_oracle_both_modes():
    from scream_amd.data import SyntheticPairs, normalize_pair
    src, tgt, T, *_ = make_3dmatch_pair(2)
    for a, b in zip(normalize_pair(src, tgt, T), O.normalize_pair(src, tgt, T)):
        assert (a == b) if isinstance(a, float) else torch.equal(a, b)
    sn, tn, rot, tr, s, c = normalize_pair(src, tgt, T, "bbox")  # datasets/kitti.py norm_pc
    both = torch.cat([(rot @ sn.T + tr).T, tn])
    ext = both.max(dim=0)[0] - both.min(dim=0)[0]
    assert abs(float(ext.max()) - 2.0) < 1e-4
    item = SyntheticPairs("3dmatch", 4, 10)[1]
    assert len(item) == 9 and item[5].tolist() == [0, 2] and item[8] == 11 % 8


def test_weighted_lane_split_balances_point_counts():
    from scream_amd.lanes import split_weighted
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 9, 32):
        w = rng.integers(2000, 20000, size=n).tolist()
        for lanes in (1, 2, 3, 40):
            parts = split_weighted(w, lanes)
            assert len(parts) == min(lanes, n) and all(len(p) > 0 for p in parts)
            assert [i for p in parts for i in p] == list(range(n))
    w = [10000, 5000, 1000, 1000, 1000, 1000, 1000]  # half the weight sits in the first item
    assert [list(p) for p in split_weighted(w, 2)] == [[0], [1, 2, 3, 4, 5, 6]]
    assert [list(p) for p in split_weighted([5] * 8, 2)] == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_lane_split_is_a_contiguous_partition():
    from scream_amd.lanes import split
    for n in (1, 2, 7, 8, 32, 33):
        for lanes in (1, 2, 3, 4, 40):
            parts = split(n, lanes)
            assert len(parts) == min(lanes, n) and all(len(p) > 0 for p in parts)
            assert [i for p in parts for i in p] == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_open_gf_voxel_down_sample_and_sample_layout():
    from scream_amd.evaluate_open_gf import SyntheticDEM, voxel_down_sample
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 100, size=(5000, 3))
    out = voxel_down_sample(pts, 20.0)
    origin = pts.min(axis=0) - 10.0
    keys = np.floor((pts - origin) / 20.0).astype(np.int64)
    assert out.shape[0] == len({tuple(k) for k in keys})
    np.testing.assert_allclose(out.mean(axis=0) * 0 + np.sort(np.floor((out - origin) / 20.0).astype(np.int64), axis=0),
                               np.sort(np.unique(keys, axis=0), axis=0))  # every centroid lies in its own voxel
    k0 = keys[0]
    np.testing.assert_allclose(out[(np.floor((out - origin) / 20.0).astype(np.int64) == k0).all(axis=1)][0],
                               pts[(keys == k0).all(axis=1)].mean(axis=0))
    dsm, coarse, dem, c = SyntheticDEM(1, seed0=3, points=500)[0]
    assert dsm.dtype == torch.float32 and dsm.shape == dem.shape == (500, 3) and coarse.shape[1] == 3
    assert torch.equal(dsm[:, :2], dem[:, :2]) and (dsm[:, 2] >= dem[:, 2] - 1e-6).all()


def test_staging_buffers_alternate_and_grow():
    from scream_amd.packing import StagingBuffers
    st = StagingBuffers()
    f0, i0 = st.take(100, 10)
    f1, i1 = st.take(50, 5)
    assert f0.data_ptr() != f1.data_ptr() and f0.numel() >= 100 and i1.numel() >= 5
    f2, _ = st.take(80, 8)
    assert f2.data_ptr() == f0.data_ptr()      # slot 0 again, large enough: reused
    f3, _ = st.take(5000, 8)
    assert f3.numel() >= 5000                   # slot 1 regrown


def test_fp16_exponents_raise_outside_their_range_instead_of_clamping():
    """scream_amd/scales.py: e = floor(log2(2^15 / bound)).  Clamping it at E_MAX only gives headroom away; clamping at E_MIN
    would break |x| 2^e <= 2^15 silently (fp16 inf / NaN in the products), so the low side raises ScaleRangeError -- which
    PointTransformer turns into the scale-free bf16 x 3 path.  Also: the contract holds at every power of two, layer_exps and
    tail_exps stay inside what scream_pack_tail accepts or raise with a message, never a bare SCREAM_EINVAL."""
    from scream_amd import scales
    from scream_amd.model import PointTransformer
    from scream_amd.synthetic import make_trained_like_state_dict
    for k in range(-60, 38):
        for b in (2.0 ** k, 2.0 ** k * 1.0000001, 2.0 ** k * 0.9999999, 3.0 * 2.0 ** k):
            e = scales.exp_for(b)
            assert scales.E_MIN <= e <= scales.E_MAX and b * 2.0 ** e <= scales.TOP
            assert e == scales.E_MAX or b * 2.0 ** (e + 1) > scales.TOP  # the largest such exponent
    assert scales.exp_for(0.0) == scales.E_MAX and scales.exp_for(2.0 ** 39) == scales.E_MIN
    for bad in (2.0 ** 39 * 1.001, 1e30, float("inf"), float("nan")):
        with pytest.raises(scales.ScaleRangeError):
            scales.exp_for(bad)
    assert issubclass(scales.ScaleRangeError, ValueError)
    # exponents of whole models: default-init and trained-like weights at three overall scales stay inside the kernels' range
    for sd in [make_state_dict(0, 256, 2, 2)] + [make_trained_like_state_dict(5, 256, 2, 2, model_scale_log2=k) for k in (0, 6, -6)]:
        net = PointTransformer(256, 2, 2)
        net.load_state_dict(sd)
        ins = net._layer_inputs()[0]
        for m, (in_q, in_kv) in zip(net._layer_modules(), ins):
            ex = scales.layer_exps(m, in_q, in_kv)
            assert all(scales.E_MIN <= v <= scales.E_MAX for v in ex.values()), ex
            assert abs(ex["e_wm"] + ex["e_att"]) <= 44 and abs(ex["e_w2"] + ex["e_h"]) <= 44 and abs(ex["e_h"] - ex["e_w1"] - ex["e_m1"]) <= 100
    # a gain that no exponent >= E_MIN can hold: layer_exps raises (with the remedy in the message)
    net = PointTransformer(256, 1, 1)
    net.load_state_dict(make_state_dict(1, 256, 1, 1))
    with torch.no_grad():
        net.stem[0].norm1.weight.mul_(2.0 ** 37)
    with pytest.raises(scales.ScaleRangeError, match="x3"):
        scales.layer_exps(net.stem[0], (net.pre_norm.weight, net.pre_norm.bias), (net.pre_norm.weight, net.pre_norm.bias))
    # accumulator units below what the tail's LayerNorm arithmetic was checked for: tail_exps raises, scream_pack_tail is never reached
    g, b = torch.ones(256), torch.zeros(256)
    with pytest.raises(scales.ScaleRangeError, match="merge"):
        scales.tail_exps(torch.full((256, 256), 2.0 ** 38), torch.ones(1024, 256), torch.ones(256, 1024), g, b, 2.0 ** 38, 1.0)


def test_split_kernel_k_loop_has_no_register_spills(tmp_path):
    """gemm_split.hip loads its A operands with inline asm, so the compiler does not know those registers are in flight:
    a spill (scratch store) of one of them inside the k-loop would save stale bytes.  Compile to assembly and require
    the k-loop of every instantiation to be free of scratch traffic (spills outside it only touch loop invariants)."""
    import re, shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scream_amd", "csrc", "gemm_split.hip")
    out = tmp_path / "x3.s"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    lines = out.read_text().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*gemm_split_kernelINS_\d+Split\w+ELi\d+ELb[01]E.*:", l)]
    assert len(starts) == 36  # three operand splits x six epilogues x two layouts of the A operand
    checked = 0
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        body = lines[a:b]
        hdr = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l]
        assert hdr, "k-loop not found"
        i = hdr[0]
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}

        def backward(k):  # a branch to a label at or above the loop header: the latch of the (rotated) k-loop
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)\b", body[k])
            return m is not None and labels.get(m.group(1), len(body)) <= i

        j = next(k for k in range(i, len(body)) if backward(k))
        loop = body[i:j]
        n_prod = 1 if "SplitH1" in body[0] else 3 if "SplitH2" in body[0] else 6
        assert sum(("v_mfma_f32_32x32x16_bf16" if n_prod == 6 else "v_mfma_f32_32x32x16_f16") in l for l in loop) == 48 * n_prod  # three k-tiles of 16 groups
        assert not any("scratch_" in l for l in loop), "register spill inside the x3 k-loop"
        # the first operand registers of the NEXT tile are in flight during the epilogue as well: spill stores are only
        # tolerated in the kernel prologue, where they save loop invariants
        outer = next(k for k, l in enumerate(body) if "Loop Header: Depth=1" in l)
        assert not any("scratch_store" in l for l in body[outer:]), "spill store inside the persistent tile loop"
        checked += 1
    assert checked == 36


def test_no_instruction_touches_a_register_whose_asm_load_is_pending(tmp_path):
    """gemm_split.hip and tail_split.hip load operands with inline asm and wait for them with hand-counted s_waitcnt vmcnt(N):
    hipcc believes such a register holds its value from the asm statement on and may copy, spill or reuse it before the
    wait (the round-1 advisor's finding; round 2 saw it happen -- a scratch_store of the destination right behind the
    load -- until the tail kernel stopped keeping requests pending across its LayerNorm blocks).
    tools/asm_inflight_check.py walks the control-flow graph of the generated code with the in-order model of the
    vector-memory queue and reports every instruction that touches a VGPR whose load is still outstanding: none allowed."""
    import shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import asm_inflight_check as chk
    for src, extra, want in (("gemm_split.hip", [], "gemm_split_kernel"), ("tail_split.hip", ["-ffp-contract=off"], "11tail_kernel")):  # (the mangled name: not pack_tail_kernel)
        out = tmp_path / (src + ".s")
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", *extra, "-o", str(out),
                        os.path.join(REPO, "scream_amd", "csrc", src)], check=True, capture_output=True, timeout=900)
        n_kernels = 0
        for name, body in chk.kernels(str(out)):
            if want not in name:
                continue
            n_kernels += 1
            n_loads = sum(1 for l in body if re.match(r"\s*global_load_dwordx4 [va]", l))
            assert n_loads >= 16, (name, n_loads)  # the asm loads are there (the check is not vacuous)
            bad = chk.check_kernel(name, body)
            assert not bad, (name, bad[:5])
            # ... and no asm memory instruction reads a scalar base that a VALU instruction (the restore of a spilled scalar)
            # wrote fewer than five wait states earlier: hipcc does not pad asm statements (a real memory fault in round 2)
            assert not chk.check_scalar_base_hazard(name, body), name
            # ... nor does a VALU instruction overwrite the data registers of a wide asm store right behind it (wrong dwords in
            # a quarter of the lanes, also round 2)
            assert not chk.check_store_data_hazard(name, body), name
            # ... nor does an asm vector instruction read a matrix-instruction result inside the window hipcc would have padded
            assert not chk.check_mfma_asm_read_hazard(name, body), name
            if src == "tail_split.hip":
                # The layer tail keeps NO scratch: a spilled value that is reloaded inside a stage puts a vmcnt(0) in front
                # of its use (hipcc cannot order a scratch reload against the LDS-DMA in flight) and drains the weight ring
                # once per stage -- it did, for a 64-bit per-lane pointer, until every address became scalar base + lane offset.
                # Full drains are allowed only at the tile boundaries (five of them at the time of writing).
                assert not any(re.match(r"\s*scratch_", l) for l in body), "scratch traffic in " + want
                drains = sum(1 for l in body if re.match(r"\s*s_waitcnt vmcnt\(0\)", l))
                assert drains <= 10, drains  # (QF: the x rows of a tile are loaded and split in the open at its start -- four more, all in front of the tile's first barrier)
        assert n_kernels == {"gemm_split.hip": 36, "tail_split.hip": 7}[src]  # every operand split of each (the tail: + the two fp16 splits with the next layer's query stages, + the same two with the layer's own query stages in front)


def test_the_static_checker_detects_what_it_is_there_for():
    """tools/asm_inflight_check.py guards the build; these are its three detectors on hand-written snippets (the patterns that
    were really seen in round 2), each with its fixed counterpart."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import asm_inflight_check as chk
    lines = lambda text: [l for l in text.strip().splitlines()]
    # (1) a pending asm-load destination copied before the counted wait
    bad = lines("""
        global_load_dwordx4 v[4:7], v0, s[2:3]
        global_load_lds_dwordx4 v[8:9], off
        v_accvgpr_write_b32 a1, v5
        s_waitcnt vmcnt(1)
        v_add_f32_e32 v9, v4, v5
    """)
    assert [b[1] for b in chk.check_kernel("k", bad)] == ["v_accvgpr_write_b32 a1, v5"]
    good = lines("""
        global_load_dwordx4 v[4:7], v0, s[2:3]
        global_load_lds_dwordx4 v[8:9], off
        s_waitcnt vmcnt(1)
        v_accvgpr_write_b32 a1, v5
    """)
    assert not chk.check_kernel("k", good)
    acc = lines("""
        global_load_dwordx4 a[0:3], v0, s[2:3]
        v_accvgpr_read_b32 v9, a2
        s_waitcnt vmcnt(0)
    """)
    assert len(chk.check_kernel("k", acc)) == 1   # loads into accumulation registers are tracked as well
    # (1b) the round-2 aperture violation (gpurun_out/r2i): an ablation build WITHOUT the weight DMA kept
    # the ring's counted wait -- vmcnt(12) leaves "the twelve pieces of the next stage" in flight, but no piece had been issued,
    # so the row operand requested just before was still pending when the MFMA consumed it (and later landed in a register the
    # allocator had recycled as a 64-bit address).  With the pieces in the queue the same wait is sound.
    r2i = lines("""
        global_load_dwordx4 v[4:7], v0, s[2:3]
        s_waitcnt vmcnt(12)
        v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]
    """)
    assert [b[2] for b in chk.check_kernel("k", r2i)] == [[4, 5, 6, 7]]
    sound = lines("global_load_dwordx4 v[4:7], v0, s[2:3]\n" + "global_load_lds_dwordx4 v1, s[4:5]\n" * 12 +
                  "s_waitcnt vmcnt(12)\nv_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]")
    assert not chk.check_kernel("k", sound)
    # (2) a scalar base restored by a VALU instruction right in front of the memory instruction that uses it
    hz = lines("""
        v_readlane_b32 s1, v255, 19
        global_store_dwordx4 v240, v[60:63], s[0:1]
    """)
    assert len(chk.check_scalar_base_hazard("k", hz)) == 1
    ok = lines("""
        v_readlane_b32 s1, v255, 19
        s_nop 4
        global_store_dwordx4 v240, v[60:63], s[0:1]
    """)
    assert not chk.check_scalar_base_hazard("k", ok)
    assert not chk.check_scalar_base_hazard("k", lines("s_add_u32 s0, s4, 16\nglobal_load_dwordx4 v[4:7], v0, s[0:1]"))  # SALU: no hazard
    # (3) the data of a wide store overwritten in the next cycle
    sd = lines("""
        global_store_dwordx4 v240, v[0:3], s[0:1]
        v_accvgpr_read_b32 v1, a156
    """)
    assert len(chk.check_store_data_hazard("k", sd)) == 1
    assert not chk.check_store_data_hazard("k", lines("global_store_dwordx4 v240, v[0:3], s[0:1]\ns_nop 1\nv_mov_b32_e32 v1, 0"))
    assert not chk.check_store_data_hazard("k", lines("global_store_dword v240, v1, s[0:1]\nv_mov_b32_e32 v1, 0"))  # 32-bit data: none
    # (4) an asm vector instruction reading a matrix-instruction result (VGPR form) too close behind it
    mh = lines("""
        v_mfma_f32_32x32x16_f16 v[0:15], v[40:43], a[120:123], v[0:15]
        s_nop 7
        v_fma_mixlo_f16 v48, v3, s50, 0
    """)
    assert len(chk.check_mfma_asm_read_hazard("k", mh)) == 1
    assert not chk.check_mfma_asm_read_hazard("k", lines("v_mfma_f32_32x32x16_f16 v[0:15], v[40:43], a[120:123], v[0:15]\ns_nop 7\ns_nop 7\ns_nop 7\nv_fma_mixlo_f16 v48, v3, s50, 0"))
    # a compiler-known instruction in between that rewrites the register (it is padded by hipcc) clears it; AGPR results are read through v_accvgpr_read
    assert not chk.check_mfma_asm_read_hazard("k", lines("v_mfma_f32_32x32x16_f16 v[0:15], v[40:43], a[120:123], v[0:15]\nv_mul_f32_e32 v3, v3, v3\nv_fma_mixlo_f16 v48, v3, s50, 0"))
    assert not chk.check_mfma_asm_read_hazard("k", lines("v_mfma_f32_32x32x16_f16 a[0:15], v[40:43], v[44:47], a[0:15]\nv_fma_mixlo_f16 v48, v3, s50, 0"))


def test_tuning_builds_are_verified_before_they_can_be_launched(tmp_path):
    """tools/tail_ablate.py / tail_stamps.py / gemm_ablate.py / gemm_stamps.py compile -D variants of the two hand-scheduled kernels;
    every variant goes through asm_inflight_check.verify_source first (round 2 launched an unverified one and faulted the GPU).
    The "no weight DMA" variant of the fp16 layer tail -- the variant that faulted, now with draining waits -- passes; a
    variant in which hipcc moves a pending register raises instead of producing a library."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import asm_inflight_check as chk
    src = os.path.join(REPO, "scream_amd", "csrc", "tail_split.hip")
    assert chk.verify_source(src, ["-ffp-contract=off", "-DT_ABLATE=1"], str(tmp_path / "t1.s"), "11tail_kernelINS_7SplitH2ELb0ELb0") == 1  # (the instance the tuning tools launch)
    bad = tmp_path / "bad.hip"  # a register load consumed behind a wait that does not cover it
    bad.write_text("""#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o) {
    f32x4 d;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p + threadIdx.x * 4));
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    o[threadIdx.x] = d[0] + d[1];
}
""")
    with pytest.raises(RuntimeError, match="still in flight"):
        chk.verify_source(str(bad), [], str(tmp_path / "bad.s"))


def test_f32_kernel_k_loop_has_no_register_spills(tmp_path):
    """gemm_f32.hip (the non-default SCREAM_GEMM=f32 path) is built at 2 blocks per CU = 128 VGPRs + 128 accumulators, and
    hipcc parks a handful of LOOP INVARIANTS (tile bookkeeping, epilogue pointers) in scratch: 6-18 dwords per
    instantiation, stored in the kernel prologue / at the top of the persistent tile loop and reloaded in the epilogue.
    That is harmless -- its A loads are ordinary compiler-tracked loads, so a spill can never capture an in-flight
    register (unlike gemm_split.hip's inline-asm loads) -- as long as none of it sits inside the k-loop, where it would cost
    a scratch round trip per 32-deep k-tile.  Pin exactly that."""
    import re, shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(REPO, "scream_amd", "csrc", "gemm_f32.hip")
    out = tmp_path / "f32.s"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    lines = out.read_text().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*gemm_f32_kernelILi\d+ELi32EE.*:", l)]
    assert len(starts) == 6
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        body = lines[a:b]
        i = next(k for k, l in enumerate(body) if "Inner Loop Header: Depth=2" in l)
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}

        def backward(k):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)\b", body[k])
            return m is not None and labels.get(m.group(1), len(body)) <= i

        j = next(k for k in range(i, len(body)) if backward(k))
        loop = body[i:j]
        assert sum("v_mfma_f32_32x32x2_f32" in l for l in loop) >= 128  # at least one 32-deep k-tile of 8 x 16 MFMAs
        assert not any("scratch_" in l for l in loop), "register spill inside the fp32 k-loop"
    assert max(int(l.split(":")[1]) for l in lines if ".vgpr_spill_count" in l) <= 32


def test_empty_dataset_evaluates_to_zeros():
    from scream_amd import dist as sdist
    from scream_amd import evaluate as ev

    class Empty:
        def __len__(self):
            return 0

        def __getitem__(self, i):
            raise IndexError(i)

    assert ev.aggregate_rows(np.zeros((0, sdist.ROW_WIDTH)), "median") == (0.0, 0.0, 0.0, 0.0)
    assert ev.evaluate_loader(None, Empty(), verbose=False) == (0.0, 0.0, 0.0, 0.0)


def test_evaluate_loader_keeps_in_flight_batches_enqueued_and_collects_in_order(monkeypatch):
    """evaluate_loader(in_flight=n): n batches are enqueued before the oldest is collected, a slot (stream, staging set,
    workspace) is reused only after the batch that last used it has been collected, rows come back in dataset order."""
    from scream_amd import dist as sdist
    from scream_amd import evaluate as ev
    log = []

    def fake_async(net, items, ids, corr, dis_thresh, icp, pred_hook=None, stream=None):
        k = len([e for e in log if e[0] == "enq"])
        log.append(("enq", k))

        def fin():
            log.append(("fin", k))
            rows = np.zeros((len(ids), sdist.ROW_WIDTH))
            rows[:, 0] = ids
            rows[:, sdist.COL_RE] = ids
            return rows
        return fin
    monkeypatch.setattr(ev, "evaluate_items_async", fake_async)

    class DS:
        def __len__(self):
            return 22

        def __getitem__(self, i):
            z = torch.zeros(4, 3)
            return (z, z, torch.eye(3), torch.zeros(3, 1), 1.0, torch.tensor([0, 2]), torch.eye(6), torch.zeros(3), 0)
    for n in (2, 3, 4):
        log.clear()
        ev.evaluate_loader(None, DS(), batch_pairs=4, icp=None, verbose=False, in_flight=n)
        n_batches = 6
        assert [k for t, k in log if t == "fin"] == list(range(n_batches))  # collected oldest first
        for k in range(n_batches):
            if k >= n:  # slot k % n is reused by batch k only after batch k - n was collected
                assert log.index(("fin", k - n)) < log.index(("enq", k))
            if k >= n - 1 and k + 1 < n_batches:  # ... and not before n batches were enqueued
                assert log.index(("enq", k)) < log.index(("fin", k - (n - 1)))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it spawns the N ranks itself (CPU-only parent), forwards rank
    0's single JSON line and propagates a failing rank's exit code (round-1 verdict: the driver's command form)."""
    import bench
    env = bench.rank_env({"PATH": "x", "WORLD_SIZE": "9"}, 2, 4, 1234)
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["LOCAL_WORLD_SIZE"]) == ("2", "2", "4", "4")
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "1234" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--dry-run"]
    r = subprocess.run(cmd, env=base, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == [0, 1] and out["steps"] == 4
    r = subprocess.run(cmd, env=dict(base, SCREAM_BENCH_DRY_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and not r.stdout.strip()


def test_ring_projection_kernel_keeps_its_registers(tmp_path):
    """csrc/proj_ring.hip holds the operand planes of a wave's 64 rows in the WHOLE accumulation register file (256 AGPRs, pinned
    with "+a" constraints) and its accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1, scream_amd/build.py): the generated code
    must have no scratch, and the only moves between the two register files are the 256 writes per tile that park the freshly split
    planes -- left to itself hipcc keeps the planes VGPR-class and moves each one back in front of every matrix instruction
    (1 141 moves, a quarter of the kernel's vector instructions, when the kernel was first built)."""
    import re, shutil, subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    from scream_amd import build as B
    src = os.path.join(REPO, "scream_amd", "csrc", "proj_ring.hip")
    out = tmp_path / "ring.s"
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", *B.SOURCES["proj_ring.hip"], "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    m = re.search(r"^(\S*proj_ring_kernelINS_7SplitH2\S*):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M)
    assert m, "proj_ring_kernel<SplitH2> not found"
    body = m.group(2)
    assert "scratch_" not in body
    assert len(re.findall(r"v_accvgpr_write", body)) == 256 and len(re.findall(r"v_accvgpr_read", body)) == 0
    assert len(re.findall(r"v_mfma_f32_32x32x16_f16 v\[", body)) == len(re.findall(r"v_mfma_f32_32x32x16_f16", body))  # accumulators in VGPRs
    meta = text[text.index("amdhsa.kernels"):]
    k = meta[meta.index("proj_ring_kernelINS_7SplitH2") - 400: meta.index("proj_ring_kernelINS_7SplitH2") + 400]
    assert ".private_segment_fixed_size: 0" in k and ".vgpr_spill_count: 0" in k
