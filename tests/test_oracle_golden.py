"""The CPU oracle (oracle/scream_ref.py) against the golden vectors produced by the
reference's own code (oracle/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import scream_ref as O
from scream_amd.synthetic import make_state_dict, state_dict_keys


def test_state_dict_layout_190_keys():
    keys = state_dict_keys(256, 6, 6)
    assert len(keys) == 190
    assert sum(int(np.prod(s)) for _, s in keys) == 14308099  # SURVEY.md 8a parameter inventory


def test_pe(golden):
    g = golden("pe")
    for d in (64, 256):
        out = O.pe_sine(torch.from_numpy(g["xyz"]), d).numpy()
        np.testing.assert_array_equal(out, g["pe_%d" % d])


def test_linear_attention(golden):
    g = golden("linattn")
    out = O.linear_attention(*(torch.from_numpy(g[k]) for k in "qkv")).numpy()
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-6)


def test_mha_intermediates(golden):
    g = golden("mha")
    sd = make_state_dict(int(g["seed"]), 256, 1, 1)
    xq, xk = torch.from_numpy(g["xq"]), torch.from_numpy(g["xk"])
    for tag, kv in (("self", xq), ("cross", xk)):
        want = {}
        out = O.mh_attention(xq, kv, kv, sd, "stem.0.", want)
        for nm in ("q", "k", "v", "att", "msg", "m1", "ffn", "out"):
            np.testing.assert_allclose(want[nm].numpy(), g["%s_%s" % (tag, nm)], rtol=2e-5, atol=2e-6, err_msg=nm)
        np.testing.assert_allclose(out.numpy(), g[tag + "_out"], rtol=2e-5, atol=2e-6)


def test_point_transformer_e2e(golden):
    g = golden("e2e")
    for seed, ns, nc, n, m, explicit in g["cases"]:
        sd = make_state_dict(int(seed), 256, int(ns), int(nc))
        center = torch.from_numpy(g["center_%d" % seed]) if explicit else None
        out = O.point_transformer_forward(torch.from_numpy(g["src_%d" % seed]), torch.from_numpy(g["tgt_%d" % seed]), sd, center)
        assert out.shape == (1, n, 3)
        np.testing.assert_allclose(out.numpy(), g["out_%d" % seed], rtol=1e-4, atol=2e-5)


def test_nn_torch_path(golden):
    g = golden("nn")
    e = golden("e2e")
    src_, tgt = torch.from_numpy(e["out_23"]), torch.from_numpy(e["tgt_23"])
    for i, s in enumerate(g["e2e_s"]):
        d, idx, _ = O.nn_search(src_, tgt, float(s), 0.1)
        np.testing.assert_array_equal(idx.numpy(), g["e2e_idx_%d" % i])
        np.testing.assert_array_equal(d.numpy(), g["e2e_d_%d" % i])


def test_nn_exact_matches_reference_bitwise(golden):
    """The numpy fp32 model used to check the HIP kernel at sizes where N x M does not fit:
    bit-identical distances and indices to the reference (incl. duplicate-point ties)."""
    g = golden("nn")
    e = golden("e2e")
    for i, s in enumerate(g["e2e_s"]):
        d, idx, d2 = O.nn_search_exact(e["out_23"][0], e["tgt_23"][0], float(s))
        np.testing.assert_array_equal(idx, g["e2e_idx_%d" % i])
        np.testing.assert_array_equal(d, g["e2e_d_%d" % i])
        np.testing.assert_array_equal(d2, g["e2e_d2_%d" % i])
    d, idx, _ = O.nn_search_exact(g["big_src"][0], g["big_tgt"][0], float(g["big_s"]))
    np.testing.assert_array_equal(idx, g["big_idx"])
    np.testing.assert_array_equal(d, g["big_d"])
    np.testing.assert_array_equal(d < np.float32(0.1), g["big_valid"])
    # duplicated target rows 450..459 == 100..109: the lowest index must win
    assert set(g["big_idx"][:10].tolist()) == set(range(100, 110))


def test_nn_exact_matches_torch_other_shapes():
    rng = np.random.default_rng(3)
    for n, m, s in ((257, 1031, 0.77), (1000, 63, 1.9), (5, 4097, 0.25)):
        a = rng.uniform(-1, 1, size=(1, n, 3)).astype(np.float32)
        b = rng.uniform(-1, 1, size=(1, m, 3)).astype(np.float32)
        d, idx, _ = O.nn_search(torch.from_numpy(a), torch.from_numpy(b), s, 0.1)
        de, ie, _ = O.nn_search_exact(a[0], b[0], s)
        np.testing.assert_array_equal(ie, idx.numpy())
        np.testing.assert_array_equal(de, d.numpy())


def test_rigid_transform(golden):
    g = golden("kabsch")
    for name in g["names"]:
        name = str(name)
        w = torch.from_numpy(g[name + "_w"]) if name + "_w" in g else None
        thr = float(g[name + "_thr"]) if name + "_thr" in g else 0
        T = O.rigid_transform_3d(torch.from_numpy(g[name + "_A"]), torch.from_numpy(g[name + "_B"]), w, thr).numpy()
        if name == "k3":  # three points: rank-2 H, still unique
            np.testing.assert_allclose(T, g[name + "_T"], atol=2e-4)
        else:
            np.testing.assert_allclose(T, g[name + "_T"], atol=1e-5, err_msg=name)
    np.testing.assert_array_equal(g["k0_T"][0], np.eye(4, dtype=np.float32))


def test_pose_metrics(golden):
    g = golden("pose_metrics")
    P = g["poses"]
    for i in range(len(P)):
        for j in range(len(P)):
            re, te = O.transformation_error(torch.from_numpy(P[i]), torch.from_numpy(P[j]))
            np.testing.assert_allclose(re.item(), g["re"][i, j], atol=1e-4)
            np.testing.assert_allclose(te.item(), g["te"][i, j], rtol=1e-6)
        q = O.rotmat2quat(P[i, :3, :3])
        np.testing.assert_allclose(q, g["quat"][i], atol=1e-6)
        assert q[0] >= 0


def test_rmse_metric_real_info(golden):
    info = golden("info")["info"]
    P = golden("pose_metrics")["poses"]
    assert info.shape == (6, 6, 6)
    assert O.rmse_metric(np.eye(4), info[0]) == 0.0
    er = np.linalg.inv(P[0].astype(np.float64)) @ P[1].astype(np.float64)
    assert O.rmse_metric(er, info[1]) > 0


def test_dem_transformer(golden):
    from scream_amd.synthetic import dem_state_dict_keys
    g = golden("dem")
    for seed, ns, nc, n, m in g["cases"]:
        sd = make_state_dict(int(seed), 256, int(ns), int(nc), dem=True)
        assert list(sd) == [k for k, _ in dem_state_dict_keys(256, int(ns), int(nc))]
        out = O.dem_transformer_forward(torch.from_numpy(g["dsm_%d" % seed]), torch.from_numpy(g["dem_%d" % seed]), sd)
        np.testing.assert_allclose(out.numpy(), g["out_%d" % seed], rtol=1e-4, atol=2e-5)
