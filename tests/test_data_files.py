"""the reference's on-disk format and test-item normalisation.

tests/golden/dataset_items.npz was produced by oracle/make_golden.py running the REFERENCE's own dataset classes
(ThreeDMatchTest / ThreeDLoMatchTest / ThreeDZeroMatchTest, datasets/three_d_match.py:219-294; KITTI_Test and norm_pc,
datasets/kitti.py:268-273,328-350) on small seeded pairs written in the layout of process_3d_match.py:38-40,199-200 and
process_kitti.py:72-74.  Here the same raw arrays are written to a temporary directory in that layout, read back through
the product's loaders, and must give the reference's items BIT FOR BIT."""
import os

import numpy as np
import pytest
import torch

from oracle import scream_ref as O
from scream_amd.data import PairFileDataset, normalize_pair

SPLITS = ("3DMatch_test", "3DLoMatch_test", "3DZeroMatch_test")


def write_split(g, split, root):
    os.makedirs(os.path.join(root, "info"), exist_ok=True)
    names = []
    for i in range(3):
        pre = "%s_%d_" % (split, i)
        for k in ("src", "tgt", "T"):
            np.save(os.path.join(root, "%s%d.npy" % (k, i)), g[pre + k])
        np.save(os.path.join(root, "info", "idx%d.npy" % i), g[pre + "idx"])
        np.save(os.path.join(root, "info", "covariance%d.npy" % i), g[pre + "cov"])
        names.append(str(g[pre + "scene_name"]))
    with open(os.path.join(root, "info", "scene_names.txt"), "w") as f:
        f.writelines(n + "\n" for n in names)


def write_kitti(g, root):
    os.makedirs(root, exist_ok=True)
    for i in range(3):
        for k in ("src", "tgt", "T"):
            np.save(os.path.join(root, "%s%d.npy" % (k, i)), g["KITTI_test_%d_%s" % (i, k)])


def same(a, b):
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_array_equal(a, np.asarray(b))
    assert a.dtype == np.asarray(b).dtype


@pytest.mark.parametrize("split", SPLITS)
def test_pair_files_give_the_reference_items_bitwise(golden, tmp_path, split):
    g = golden("dataset_items")
    write_split(g, split, str(tmp_path / split))
    ds = PairFileDataset(str(tmp_path / split))
    assert len(ds) == 3
    for i in range(3):
        item = ds[i]
        pre = "%s_%d_out_" % (split, i)
        for key, val in zip(("src", "tgt", "rot", "trans"), item[:4]):
            same(val, g[pre + key])
        assert item[4] == float(g[pre + "s"])                      # the scale stays a python float (fp64), as in the reference
        same(item[5], g[pre + "idx"])
        same(item[6], g[pre + "cov"])
        same(item[7], g[pre + "c"])
        assert item[8] == int(g[pre + "scene"])


def test_normalize_pair_both_modes_vs_reference_items(golden):
    """normalize_pair (product) and oracle.normalize_pair against the reference's __getitem__ arithmetic and norm_pc."""
    g = golden("dataset_items")
    for i in range(3):
        pre = "3DMatch_test_%d_" % i
        for fn in (normalize_pair, O.normalize_pair):
            out = fn(g[pre + "src"], g[pre + "tgt"], g[pre + "T"])
            for key, val in zip(("src", "tgt", "rot", "trans"), out[:4]):
                same(val, g[pre + "out_" + key])
            assert out[4] == float(g[pre + "out_s"])
            same(out[5], g[pre + "out_c"])
        pre = "KITTI_test_%d_" % i
        out = normalize_pair(g[pre + "src"], g[pre + "tgt"], g[pre + "T"], "bbox")
        for key, val in zip(("src", "tgt", "rot", "trans"), out[:4]):
            same(val, g[pre + "out_" + key])
        assert out[4] == float(g[pre + "out_s"]) == float(g[pre + "norm_pc_s"])
        same(out[5], g[pre + "out_c"])
        np.testing.assert_array_equal(g[pre + "norm_pc_c"].astype(np.float32), g[pre + "out_c"])


def test_kitti_pair_files_give_the_reference_items_bitwise(golden, tmp_path):
    from scream_amd.evaluate_kitti import KittiPairFiles
    g = golden("dataset_items")
    write_kitti(g, str(tmp_path / "KITTI_test"))
    ds = KittiPairFiles(str(tmp_path / "KITTI_test"))
    assert len(ds) == 3
    for i in range(3):
        item = ds[i]
        pre = "KITTI_test_%d_out_" % i
        for key, val in zip(("src", "tgt", "rot", "trans"), item[:4]):
            same(val, g[pre + key])
        assert item[4] == float(g[pre + "s"])
        same(item[5], g[pre + "c"])


def test_open_gf_files_equal_in_memory_samples(tmp_path):
    """datasets/open_gf.py:54-69: <root>/<i>.npy float [N,6] (dsm | dem), i = 1..count, and <root>/centers/<i>.npy."""
    from scream_amd.evaluate_open_gf import OpenGFFiles, SyntheticDEM, make_sample
    root = tmp_path / "OpenGF_test"
    os.makedirs(root / "centers")
    rng = np.random.default_rng(3)
    raws = []
    for i in range(1, 4):
        n = int(rng.integers(300, 600))
        xy = rng.uniform(0, 500, size=(n, 2))
        ground = 5 * np.sin(xy[:, 0] / 60) + 0.01 * xy[:, 1]
        arr = np.concatenate([xy, (ground + rng.uniform(0, 8, size=n))[:, None], xy, ground[:, None]], axis=1)
        center = rng.uniform(0, 1000, size=3)
        np.save(root / ("%d.npy" % i), arr)
        np.save(root / "centers" / ("%d.npy" % i), center)
        raws.append((arr, center))
    ds = OpenGFFiles(str(root), count=3)
    assert len(ds) == 3
    for i, (arr, center) in enumerate(raws):
        got, want = ds[i], make_sample(arr, center)
        for a, b in zip(got[:3], want[:3]):
            assert torch.equal(a, b) and a.dtype == torch.float32
        np.testing.assert_array_equal(got[3], center)
        assert got[0].shape == (arr.shape[0], 3) and got[1].shape[0] < arr.shape[0]  # the 20 m voxel grid thins the DEM
    assert len(SyntheticDEM(2)[0]) == 4


def test_prefetching_loader_yields_the_same_items_in_order(golden, tmp_path):
    """evaluate_loader's DataLoader (worker processes, var-len collate) over the file dataset: same items, same order.
    A worker ships a batch as three flat tensors (collate_pairs in a worker process); unpack_batch restores the 9-tuples
    bit for bit -- dtypes, shapes and values -- and passes an in-process batch (a plain list) through."""
    from scream_amd.data import collate_pairs, unpack_batch
    g = golden("dataset_items")
    write_split(g, "3DMatch_test", str(tmp_path / "3DMatch_test"))
    ds = PairFileDataset(str(tmp_path / "3DMatch_test"))
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=collate_pairs, num_workers=2)
    batches = list(dl)
    assert all(isinstance(b, (tuple, list)) and isinstance(b[0], str) and len(b) == 4 for b in batches)  # the packed form
    flat = [it for batch in batches for it in unpack_batch(batch)]
    assert len(flat) == 3
    assert unpack_batch([ds[0], ds[1]])[1][0] is not None and len(unpack_batch(collate_pairs([ds[0]]))) == 1  # main-process path: a list
    for i, it in enumerate(flat):
        ref = ds[i]
        for a, b in zip(it, ref):
            if torch.is_tensor(a):
                assert torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape
            else:
                assert a == b and type(a) == type(b)


@pytest.mark.gpu
def test_evaluate_loader_from_files_with_workers_equals_in_memory(golden, tmp_path):
    """SURVEY.md 8f row 3 end to end: evaluate_loader(num_workers=2) on a split directory in the reference's layout
    returns exactly what the in-memory run of the same items returns (pre-ICP and with the default GPU ICP)."""
    from scream_amd.evaluate import evaluate_loader
    from scream_amd.model import PointTransformer
    from scream_amd.synthetic import make_3dmatch_pair, make_state_dict
    assert torch.cuda.is_available()
    root = tmp_path / "3DMatch_test"
    os.makedirs(root / "info")
    scenes = ["7-scenes-redkitchen", "sun3d-hotel_uc-scan3", "sun3d-home_at-home_at_scan1_2013_jan_1", "sun3d-hotel_uc-scan3"]
    for i in range(4):
        src, tgt, T, idx, cov, _ = make_3dmatch_pair(200 + i, n_samples=30000)
        np.save(root / ("src%d.npy" % i), src)
        np.save(root / ("tgt%d.npy" % i), tgt)
        np.save(root / ("T%d.npy" % i), T)
        np.save(root / "info" / ("idx%d.npy" % i), idx)
        np.save(root / "info" / ("covariance%d.npy" % i), cov)
    with open(root / "info" / "scene_names.txt", "w") as f:
        f.writelines(s + "\n" for s in scenes)
    ds = PairFileDataset(str(root))
    items = [ds[i] for i in range(4)]
    net = PointTransformer(256, 1, 1)
    net.load_state_dict(make_state_dict(2, 256, 1, 1))
    net = net.to("cuda:0").eval()

    def hook(batch, src_pred, ids):  # registered src + 1 cm noise: the threshold keeps real correspondences
        out = src_pred.clone()
        for k, i in enumerate(ids):
            it = items[i]
            rng = np.random.default_rng(1000 + i)
            reg = (it[2] @ it[0].T + it[3]).T + torch.from_numpy(rng.normal(scale=0.01 * it[4], size=it[0].shape).astype(np.float32))
            r0 = int(batch.cloud_row0_host[k])
            out[r0:r0 + it[0].shape[0]] = reg.to("cuda:0")
        return out

    class Mem(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return items[i]

    for icp in (None, "gpu"):
        want = evaluate_loader(net, Mem(), batch_pairs=3, verbose=False, pred_hook=hook, icp=icp)
        got = evaluate_loader(net, ds, batch_pairs=3, verbose=False, pred_hook=hook, icp=icp, num_workers=2)
        assert got == want, (icp, got, want)
        assert got[3] > 0.5  # pairs register
