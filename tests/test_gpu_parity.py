"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden vectors.  Needs an MI355X: run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from oracle import scream_ref as O
from scream_amd import ops, scales
from scream_amd.synthetic import make_3dmatch_pair, make_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    from scream_amd import _lib
    _lib.load()  # fails loudly if libscream_hip.so is missing: there is no fallback path to test


def dev(x):
    return torch.as_tensor(x).to(DEV)


BACKENDS = ["h2", "x3", "f32"]  # every GEMM path of the forward is held to the same tolerances
SPLITS = {"h2": ops.SPLIT_H2, "x3": ops.SPLIT_BF3}  # 2 x fp16 / 3 products (default), 3 x bf16 / 6 products


def build_net(seed, n_self, n_cross, backend=None):
    from scream_amd.model import PointTransformer
    net = PointTransformer(256, n_self, n_cross)
    if backend is not None:
        net.gemm_backend = backend
    net.load_state_dict(make_state_dict(seed, 256, n_self, n_cross), strict=True)
    return net.to(DEV).eval()


# ------------------------------------------------------------------------------------- GEMM
def _gemm(kind, A, W, *a, **k):
    """The same GEMM contract on the three matrix-core paths: fp32-input MFMA, 3 x bf16 split, 2 x fp16 split (whose
    operand exponents default to the largest that max|A|, max|W| allow -- what a caller with a bound would pass)."""
    return ops.gemm_f32(A, W, *a, **k) if kind == "f32" else ops.gemm_split(A, ops.pack_w(W, SPLITS[kind]), *a, **k)


def tail_exps_for(sd, pre, v_absmax, q_absmax):
    """scream_tail_exps_t of block `pre` for inputs whose value rows stay below v_absmax and whose query projections stay below
    q_absmax (tests feed raw data, not LayerNorm outputs, into single blocks: the forward derives the same exponents from the
    weights alone, scream_amd/scales.py)."""
    return ops.tail_exps(**scales.tail_exps(sd[pre + "merge.weight"], sd[pre + "mlp.0.weight"], sd[pre + "mlp.2.weight"],
                                            sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], v_absmax, q_absmax))


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("M,N,K", [(128, 256, 256), (384, 768, 256), (256, 1024, 256), (256, 256, 1024), (128, 512, 64), (66048, 256, 256)])
def test_gemm_plain_and_activations(M, N, K, kind):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = (A.double() @ W.double().t())
    out = _gemm(kind, dev(A), dev(W)).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=2e-5)
    out = _gemm(kind, dev(A), dev(W), ops.EPI_RELU).cpu()
    torch.testing.assert_close(out.double(), ref.clamp_min(0), rtol=1e-5, atol=2e-5)
    if N >= 512:
        out = _gemm(kind, dev(A), dev(W), ops.EPI_ELU1, n_act=N - 256).cpu()
        want = ref.clone()
        want[:, : N - 256] = torch.nn.functional.elu(ref[:, : N - 256]) + 1
        torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=2e-5)
    bias = torch.randn(N, generator=g)
    out = _gemm(kind, dev(A), dev(W), ops.EPI_BIAS_RELU, bias=dev(bias)).cpu()
    torch.testing.assert_close(out.double(), (ref + bias.double()).clamp_min(0), rtol=1e-5, atol=2e-5)


def test_pack_w_bf3_is_an_exact_split():
    """scream_pack_w_split(SCREAM_SPLIT_BF3): the three bf16 planes sum back to W bit for bit (in fp32 and in fp64), and the image is the
    documented k-tile-major layout: logical chunk c = 2 half + s of a 32-deep k-slice holds the contraction indices
    8 (2 s + (j >> 2)) + 4 half + (j & 3), j = 0 .. 7 (what lane-half `half` supplies in step s), stored at chunk
    c ^ ((n >> 2) & 3) (gemm_split.hip)."""
    g = torch.Generator().manual_seed(3)
    N, K = 512, 160
    W = torch.randn(N, K, generator=g) * torch.logspace(-6, 3, N).unsqueeze(1)  # nine decades of magnitudes
    img = ops.pack_w(dev(W), ops.SPLIT_BF3).data.cpu()  # [3, K/32, N, 32]
    assert img.shape == (3, K // 32, N, 32) and img.dtype == torch.bfloat16
    n = torch.arange(N)
    chunk = (torch.arange(4).view(1, 4) ^ ((n >> 2) & 3).view(N, 1))  # stored chunk cs of row n holds logical chunk cs ^ swz
    planes = torch.empty(3, N, K)
    for kt in range(K // 32):
        rows = img[:, kt].float().view(3, N, 4, 8)
        for cs in range(4):
            c = chunk[:, cs]
            for cc in range(4):
                sel = c == cc
                hf, st = cc >> 1, cc & 1
                ks = [kt * 32 + 8 * (2 * st + (j >> 2)) + 4 * hf + (j & 3) for j in range(8)]
                planes[:, sel][:, :, ks] = rows[:, sel, cs]
                tmp = planes[:, sel]
                tmp[:, :, ks] = rows[:, sel, cs]
                planes[:, sel] = tmp
    assert torch.equal((planes[0] + planes[1]) + planes[2], W)
    assert torch.equal(planes.double().sum(0), W.double())
    assert torch.equal(planes[0], W.to(torch.bfloat16).float())


def test_pack_w_h2_planes_and_layout():
    """scream_pack_w_split(SCREAM_SPLIT_H2): same k-tile-major layout with two fp16 planes of W * 2^w_exp: the leading plane
    is the round-to-nearest fp16 of the scaled value, the second the fp16 of the exact residual, so the pair carries 22
    significant bits; w_exp is the largest exponent that keeps max|W| 2^w_exp <= 2^15; elements more than 2^18 below the
    largest one keep an ABSOLUTE accuracy of 2^-25 / 2^w_exp (fp16 subnormals are kept by the conversion and by the MFMA)."""
    g = torch.Generator().manual_seed(3)
    N, K = 512, 160
    W = torch.randn(N, K, generator=g) * torch.logspace(-6, 3, N).unsqueeze(1)  # nine decades of magnitudes
    pw = ops.pack_w(dev(W), ops.SPLIT_H2)
    img = pw.data.cpu()
    assert img.shape == (2, K // 32, N, 32) and img.dtype == torch.float16
    s = 2.0 ** pw.w_exp
    assert 2.0 ** 14 < float(W.abs().max()) * s <= 2.0 ** 15
    n = torch.arange(N)
    chunk = (torch.arange(4).view(1, 4) ^ ((n >> 2) & 3).view(N, 1))
    planes = torch.empty(2, N, K, dtype=torch.float64)
    for kt in range(K // 32):
        rows = img[:, kt].double().view(2, N, 4, 8)
        for cs in range(4):
            c = chunk[:, cs]
            for cc in range(4):
                sel = c == cc
                hf, st = cc >> 1, cc & 1
                ks = [kt * 32 + 8 * (2 * st + (j >> 2)) + 4 * hf + (j & 3) for j in range(8)]
                tmp = planes[:, sel]
                tmp[:, :, ks] = rows[:, sel, cs]
                planes[:, sel] = tmp
    Ws = W.double() * s
    assert torch.equal(planes[0], (W * s).to(torch.float16).double())
    assert torch.equal(planes[1], ((W * s) - (W * s).to(torch.float16).float()).to(torch.float16).double())
    err = (planes.sum(0) - Ws).abs()
    assert bool((err <= Ws.abs() * 2.0 ** -22 + 2.0 ** -25).all())


def test_gemm_split_rejects_unsupported_k_and_exponents():
    from scream_amd._lib import ScreamHipError
    A = torch.zeros(128, 128, device=DEV)
    for split, dt in ((ops.SPLIT_BF3, torch.bfloat16), (ops.SPLIT_H2, torch.float16)):
        with pytest.raises(ScreamHipError):
            ops.pack_w(torch.zeros(256, 48, device=DEV), split, 0)  # K % 32
        Wp = ops.PackedW(torch.zeros(split, 4, 256, 32, device=DEV, dtype=dt), split, 0, 256, 128)  # K = 128 is not 64 + 192 j
        with pytest.raises(ScreamHipError):
            ops.gemm_split(A, Wp, a_exp=0)
        # K = 160: five k-tiles. 2 mod 3 but ODD -- the two-stage ring would read a stale stage (round-1 advisor finding)
        with pytest.raises(ScreamHipError):
            ops.gemm_split(torch.zeros(128, 160, device=DEV), ops.PackedW(torch.zeros(split, 5, 256, 32, device=DEV, dtype=dt), split, 0, 256, 160), a_exp=0)
    with pytest.raises(ScreamHipError):
        ops.pack_w(torch.zeros(256, 256, device=DEV), 4, 0)  # no such split
    W = ops.pack_w(torch.zeros(256, 256, device=DEV), ops.SPLIT_H2, 0)
    with pytest.raises(ScreamHipError):
        ops.gemm_split(torch.zeros(128, 256, device=DEV), W, a_exp=200)  # exponent outside the checked range


@pytest.mark.parametrize("kind", BACKENDS)
def test_gemm_asymmetric_identity(kind):
    """A = I with an asymmetric W catches a transposed fragment/C map (cdna guide, section 3); exact on the fp32 and bf16 x 3
    paths (the 3-way bf16 split of W is exact and 1.0 x w needs no rounding), to the 22 bits of its two planes on fp16 x 2."""
    K = N = 256
    A = torch.zeros(128, K)
    A[torch.arange(128), torch.arange(128)] = 1.0
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0
    out = _gemm(kind, dev(A), dev(W)).cpu()
    torch.testing.assert_close(out, W.t()[:128].contiguous(), rtol=2.0 ** -22 if kind == "h2" else 0, atol=2.0 ** -25 if kind == "h2" else 0)


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("K", [256, 1024])
def test_gemm_residual_layernorm(K, kind):
    g = torch.Generator().manual_seed(K)
    M, N = 256, 256
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    res = torch.randn(M, N, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    want = torch.nn.functional.layer_norm((A.double() @ W.double().t()) + res.double(), (N,), gamma.double(), beta.double(), 1e-5)
    out = _gemm(kind, dev(A), dev(W), ops.EPI_RES_LN, residual=dev(res), gamma=dev(gamma), beta=dev(beta)).cpu()
    torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=2e-5)


def test_gemm_rejects_bad_shapes():
    from scream_amd._lib import ScreamHipError
    with pytest.raises(ScreamHipError):
        ops.gemm_f32(dev(torch.zeros(100, 256)), dev(torch.zeros(256, 256)))
    with pytest.raises(ScreamHipError):
        ops.gemm_f32(torch.zeros(128, 256), torch.zeros(256, 256))  # CPU tensors: no fallback


def test_elu_feature_map_is_within_two_ulp_of_float64():
    """elu(x) + 1 of every epilogue (csrc/common.h:elu1: exp2 on the hardware with a first-order correction of the argument's
    rounding) against float64, through the fp32-input GEMM with an identity weight (its products and sums are exact there)."""
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.rand(128, 256, generator=g) * 100 - 95, torch.randn(128, 256, generator=g) * 3,
                   torch.tensor([0.0, -0.0, -1e-30, -87.0, -103.0, -200.0, 1e-30, 88.0, 3e38]).repeat(256 * 128 // 9 + 1)[:256 * 128].reshape(128, 256)])
    got = ops.gemm_f32(dev(x), dev(torch.eye(256)), ops.EPI_ELU1, n_act=256).cpu().double()
    xd = x.double()
    want = torch.where(xd > 0, xd + 1.0, torch.exp(xd))
    # (v_exp_f32 flushes results below the smallest normal number to zero, the reference's exp returns them as denormals: 1e-38 of
    # a feature whose useful range starts nine orders of magnitude higher)
    ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64)) * 2.0 ** -23
    assert torch.isfinite(got).all()
    err = ((got - want).abs() - 2.0 ** -126).clamp(min=0) / ulp
    assert float(err.max()) <= 2.0, float(err.max())


# ------------------------------------------------------------------------------ A1 embedding
def test_pe_sine_embedding_against_float64():
    """The sine embedding (models/transformer.py:157-179) by itself -- zero 1x1-conv weights, unit pre_norm, so the kernel's output is
    LayerNorm(PE(xyz)) -- against float64, on normalised coordinates (|p| <= 2 pi), on coordinates of hundreds of units (the kernel's
    own argument reduction, csrc/embed.hip:sincos_pe) and beyond its range (the libm path)."""
    from scream_amd.model import pe_dim_t
    from scream_amd.packing import PackedBatch
    g = torch.Generator().manual_seed(5)
    # (the fp32 LayerNorm behind the embedding -- outputs up to ~2.5, ulp 2.4e-7, a few roundings -- accounts for ~5e-7 by itself)
    for scale, tol in ((1.0, 1.2e-6), (300.0, 1.2e-6), (5000.0, 1.2e-6)):
        src = (torch.rand(300, 3, generator=g) * 2 - 1) * scale
        tgt = (torch.rand(140, 3, generator=g) * 2 - 1) * scale
        b = PackedBatch.from_pairs([dev(src)], [dev(tgt)], [dev(torch.zeros(3))])
        z = lambda *sh: dev(torch.zeros(*sh))
        feats = ops.pe_embed_ln(b.xyz, b.tile_cloud, b.center, dev(pe_dim_t()), z(256, 3), z(256), dev(torch.ones(256)), z(256)).cpu().double()
        dim_t = pe_dim_t()  # fp32, as the reference holds it
        for x32, r0 in ((src, 0), (tgt, b.rows_src)):
            p = (x32 * np.float32(2 * np.pi))[:, :, None] / dim_t[None, None, :]          # the reference's fp32 argument ...
            p = p.double()                                                                # ... evaluated exactly from there on
            pe = torch.stack([p[:, :, 0::2].sin(), p[:, :, 1::2].cos()], dim=3).flatten(2).flatten(1)
            pe = torch.cat([pe, torch.zeros(pe.shape[0], 4, dtype=torch.float64)], dim=1)
            want = (pe - pe.mean(1, keepdim=True)) / (pe.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
            err = float((feats[r0:r0 + x32.shape[0]] - want).abs().max())
            assert err <= tol, (scale, err)


def test_pe_embed_prenorm_vs_oracle(golden):
    from scream_amd.model import pe_dim_t
    from scream_amd.packing import PackedBatch
    sd = make_state_dict(3, 256, 1, 1)
    rng = np.random.default_rng(0)
    src = torch.from_numpy(rng.uniform(-1, 1, size=(150, 3)).astype(np.float32))
    tgt = torch.from_numpy(rng.uniform(-1, 1, size=(70, 3)).astype(np.float32))
    center = torch.tensor([0.1, -0.2, 0.05])
    b = PackedBatch.from_pairs([dev(src)], [dev(tgt)], [dev(center)])
    feats = ops.pe_embed_ln(b.xyz, b.tile_cloud, b.center, dev(pe_dim_t()), dev(sd["embedding.weight"][:, :, 0].contiguous()),
                            dev(sd["embedding.bias"]), dev(sd["pre_norm.weight"]), dev(sd["pre_norm.bias"])).cpu()
    want_s = O.embed_prenorm(src, src - center, sd)
    want_t = O.embed_prenorm(tgt, tgt, sd)
    torch.testing.assert_close(feats[:150], want_s, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(feats[b.rows_src: b.rows_src + 70], want_t, rtol=1e-4, atol=2e-5)
    # the fused forward's variant writes the same values fragment-major, without the separate layout pass
    frag = ops.pe_embed_ln(b.xyz, b.tile_cloud, b.center, dev(pe_dim_t()), dev(sd["embedding.weight"][:, :, 0].contiguous()),
                           dev(sd["embedding.bias"]), dev(sd["pre_norm.weight"]), dev(sd["pre_norm.bias"]), frag=True)
    assert torch.equal(ops.act_layout(frag, False).cpu(), feats)
    assert torch.equal(frag.cpu(), ops.act_layout(dev(feats), True).cpu())


# ---------------------------------------------------------------------- A3 linear attention
def test_linear_attention_vs_golden(golden):
    g = golden("linattn")
    q, k, v = (torch.from_numpy(g[n])[0] for n in "qkv")  # [L,8,32], [S,8,32]
    L, S = q.shape[0], k.shape[0]
    Qf = torch.zeros(128, 256)
    Qf[:L] = torch.nn.functional.elu(q.reshape(L, 256)) + 1
    KV = torch.zeros(128, 512)
    KV[:S, :256] = torch.nn.functional.elu(k.reshape(S, 256)) + 1
    KV[:S, 256:] = v.reshape(S, 256)
    KV[S:, :256] = 7.0  # padding rows must not leak into the reduction
    KVd = dev(KV)
    row0, clen = dev(torch.tensor([0, 0], dtype=torch.int32)), dev(torch.tensor([L, S], dtype=torch.int32))
    kv = ops.kv_reduce(KVd, KVd[:, 256:], 512, 0, row0, clen, 1, 1, 1, 2)
    out = ops.attn_apply(dev(Qf), 256, kv, dev(torch.tensor([0], dtype=torch.int32)), 1, clen, 128).cpu()
    torch.testing.assert_close(out[:L].reshape(L, 8, 32), torch.from_numpy(g["out"])[0], rtol=2e-5, atol=2e-6)


def test_kv_reduce_multichunk_vs_oracle():
    rng = np.random.default_rng(1)
    S = 1000  # 4 chunks, ragged tail
    k = torch.from_numpy(rng.normal(size=(1, S, 8, 32)).astype(np.float32))
    v = torch.from_numpy(rng.normal(size=(1, S, 8, 32)).astype(np.float32))
    want = {}
    O.linear_attention(k[:, :5], k, v, want)
    buf = torch.zeros(1024, 512)
    buf[:S, :256] = want["K"].reshape(S, 256)
    buf[:S, 256:] = v.reshape(S, 256)
    b = dev(buf)
    kv = ops.kv_reduce(b, b[:, 256:], 512, 0, dev(torch.tensor([0], dtype=torch.int32)),
                       dev(torch.tensor([S], dtype=torch.int32)), 0, 1, 4, 1).cpu()
    kvt = kv[0, :, : 32 * 32].reshape(8, 32, 32)  # [h][v][d]
    torch.testing.assert_close(kvt.permute(0, 2, 1), want["KV"][0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(kv[0, :, 32 * 32:], want["Ksum"][0], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("kind", BACKENDS)
def test_fused_qkv_projection_and_kv_reduce_vs_oracle(kind):
    """scream_gemm_qkv_f32 + scream_kv_finalize: Q' = elu(x Wq^T)+1 stored, K^T (V/S) and Ksum reduced from the
    accumulators per cloud -- against the oracle's intermediates, two ragged clouds incl. padding rows."""
    sd = make_state_dict(9, 256, 1, 1)
    rng = np.random.default_rng(2)
    lens = [300, 129]
    row0 = [0, 384]
    rows = 640
    x = torch.zeros(rows, 256)
    xs = [torch.from_numpy(rng.normal(size=(n, 256)).astype(np.float32)) for n in lens]
    for r0, xc in zip(row0, xs):
        x[r0:r0 + xc.shape[0]] = xc
    x[300:384] = 3.0  # garbage in padding rows must not reach the reduction
    q, k, v = (sd["stem.0.%s_proj.weight" % n] for n in "qkv")
    W = torch.cat([q, k[:128], v[:128], k[128:], v[128:]], dim=0)
    tile_cloud = dev(torch.tensor([0, 0, 0, 1, 1], dtype=torch.int32))
    crow0, clen = dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    pack = (lambda w: dev(w)) if kind == "f32" else (lambda w: ops.pack_w(dev(w), SPLITS[kind]))
    Q, part = ops.gemm_qkv(dev(x), pack(W), 256, tile_cloud, crow0, clen, 0)
    kv = ops.kv_finalize(part, crow0, clen, 0, 0, 2, 2).cpu()
    for ci, (r0, xc) in enumerate(zip(row0, xs)):
        want = {}
        O.mh_attention(xc[None], xc[None], xc[None], sd, "stem.0.", want)
        n = xc.shape[0]
        torch.testing.assert_close(Q[r0:r0 + n].cpu().reshape(n, 8, 32), want["Q"][0], rtol=1e-5, atol=1e-5)
        kvt = kv[ci, :, :1024].reshape(8, 32, 32)  # [h][v][d]
        torch.testing.assert_close(kvt.permute(0, 2, 1), want["KV"][0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(kv[ci, :, 1024:], want["Ksum"][0], rtol=1e-5, atol=1e-4)
    # key/value-only form used by the cross layers (n_q = 0, rows offset by row_base)
    Wkv = W[256:].contiguous()
    _, part2 = ops.gemm_qkv(dev(x[384:]), pack(Wkv), 0, tile_cloud, crow0, clen, 384)
    kv2 = ops.kv_finalize(part2, crow0, clen, 384, 1, 1, 2).cpu()
    torch.testing.assert_close(kv2[1], kv[1], rtol=0, atol=0)


@pytest.mark.parametrize("split", ["h2", "h1"])
def test_ring_projection_vs_oracle_and_vs_the_eight_wave_gemm(split):
    """scream_proj_qkv_f32 (csrc/proj_ring.hip: 64 rows per wave, weights through the LDS ring, epilogues riding under the next
    chunk's matrix instructions) against the oracle's intermediates (models/transformer.py:79-81,28-29,38-41) and against
    scream_gemm_qkv_split_f32 on the same fragment-major input and exponents: Q' of real rows, K^T (V/S), Ksum per cloud; ragged
    clouds with garbage in the padding rows, an odd number of 128-row tiles (two idle waves in the last 256-row tile), the
    key/value-only form of two layers with a row base, and another cut of the rows into 256-row tiles."""
    SPL = {"h2": ops.SPLIT_H2, "h1": ops.SPLIT_H1}[split]
    sd = make_state_dict(9, 256, 1, 1)
    rng = np.random.default_rng(2)
    lens, row0, rows = [300, 129, 700], [0, 384, 640], 1408  # 11 tiles of 128 rows
    x = torch.zeros(rows, 256)
    xs = [torch.from_numpy(rng.normal(size=(n, 256)).astype(np.float32)) for n in lens]
    for r0, xc in zip(row0, xs):
        x[r0:r0 + xc.shape[0]] = xc
    x[300:384] = 3.0
    x[640 + 700:] = -2.5
    q, k, v = (sd["stem.0.%s_proj.weight" % n] for n in "qkv")
    W = torch.cat([q, k[:128], v[:128], k[128:], v[128:]], dim=0)
    tile_cloud = dev(torch.tensor([0] * 3 + [1] * 2 + [2] * 6, dtype=torch.int32))
    crow0, clen = dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    xf = ops.act_layout(dev(x), True)
    w_exp, a_exp = scales.w_exp(W), scales.exp_for(float(x.abs().max()))
    P = ops.pack_proj(dev(W), 256, SPL, w_exp)
    rl = P.row_l1[256:].view(-1, 2, 128)
    k_exp, v_exp = scales.exp_for(1.0 + 6.0 * float(rl[:, 0].max())), scales.exp_for(6.0 * float(rl[:, 1].max()))
    Qf, part = ops.proj_qkv(xf, P, tile_cloud, crow0, clen, 0, a_exp=a_exp, k_exp=k_exp, v_exp=v_exp)
    Q = ops.act_layout(Qf, False).cpu()
    kv = ops.kv_finalize(part, crow0, clen, 0, 0, 3, 3).cpu()
    # the 8-wave GEMM on the same operands
    Qg, partg = ops.gemm_qkv(xf, ops.pack_w(dev(W), SPL, w_exp), 256, tile_cloud, crow0, clen, 0, layout=ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG,
                             a_exp=a_exp, k_exp=k_exp, v_exp=v_exp)
    Qg = ops.act_layout(Qg, False).cpu()
    kvg = ops.kv_finalize(partg, crow0, clen, 0, 0, 3, 3).cpu()
    tol = dict(rtol=1e-5, atol=1e-5) if split == "h2" else dict(rtol=3e-3, atol=3e-3)
    for ci, (r0, xc) in enumerate(zip(row0, xs)):
        n = xc.shape[0]
        assert torch.equal(Q[r0:r0 + n], Qg[r0:r0 + n]) or split == "h2" and (Q[r0:r0 + n] - Qg[r0:r0 + n]).abs().max() < 2e-6  # same products, another internal order
        torch.testing.assert_close(kv[ci], kvg[ci], rtol=2e-5, atol=2e-5)
        if split != "h2":
            continue
        want = {}
        O.mh_attention(xc[None], xc[None], xc[None], sd, "stem.0.", want)
        torch.testing.assert_close(Q[r0:r0 + n].reshape(n, 8, 32), want["Q"][0], **tol)
        kvt = kv[ci, :, :1024].reshape(8, 32, 32)  # [h][v][d]
        torch.testing.assert_close(kvt.permute(0, 2, 1), want["KV"][0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(kv[ci, :, 1024:], want["Ksum"][0], rtol=1e-5, atol=1e-4)
    # deterministic, and the same bits wherever a block's share of the stage sequence begins: rows [384, 1408) as their own launch
    # (other tile boundaries: cloud 1 now starts a 256-row tile) reproduce the partials of those tiles
    Qf2, part2 = ops.proj_qkv(xf, P, tile_cloud, crow0, clen, 0, a_exp=a_exp, k_exp=k_exp, v_exp=v_exp)
    assert torch.equal(Qf2, Qf) and torch.equal(part2, part)
    Qf3, part3 = ops.proj_qkv(xf[384:], P, tile_cloud, crow0, clen, 384, a_exp=a_exp, k_exp=k_exp, v_exp=v_exp)
    assert torch.equal(Qf3, Qf[384:]) and torch.equal(part3, part[3:])
    # key/value-only form of several layers (the cross stage's target side), with a row base
    Wkv = torch.cat([W[256:], 2.0 * W[256:].flip(0)])
    wk_exp = scales.w_exp(Wkv)
    Pk = ops.pack_proj(dev(Wkv), 0, SPL, wk_exp)
    _, partk = ops.proj_qkv(xf[384:], Pk, tile_cloud, crow0, clen, 384, a_exp=a_exp, k_exp=k_exp - 1, v_exp=v_exp - 1)
    _, partkg = ops.gemm_qkv(xf[384:], ops.pack_w(dev(Wkv), SPL, wk_exp), 0, tile_cloud, crow0, clen, 384, layout=ops.LAYOUT_A_FRAG,
                             a_exp=a_exp, k_exp=k_exp - 1, v_exp=v_exp - 1)
    assert partk.shape == partkg.shape == (2, 8, 8, 1056)
    for l in range(2):
        a = ops.kv_finalize(partk[l], crow0, clen, 384, 1, 2, 3).cpu()
        b = ops.kv_finalize(partkg[l], crow0, clen, 384, 1, 2, 3).cpu()
        torch.testing.assert_close(a[1:], b[1:], rtol=2e-5 if split == "h2" else 1e-3, atol=2e-5 if split == "h2" else 1e-3)


def test_ring_projection_smallest_launches():
    """One 128-row tile (a single 256-row unit range with two idle waves and fewer units than blocks), with and without queries, and
    a six-layer key/value-only call on it: against the 8-wave GEMM (Q' bit for bit, K^T V per cloud to fp32 rounding)."""
    g = torch.Generator().manual_seed(23)
    x = torch.randn(128, 256, generator=g).clamp_(-6, 6)
    tc, cr, cl = dev(torch.tensor([0], dtype=torch.int32)), dev(torch.tensor([0], dtype=torch.int32)), dev(torch.tensor([77], dtype=torch.int32))
    xf = ops.act_layout(dev(x), True)
    a_exp = scales.exp_for(6.0)
    for N, n_q in ((768, 256), (512, 0), (3072, 0)):
        W = torch.randn(N, 256, generator=g) / 16
        w_exp = scales.w_exp(W)
        rl = W.abs().sum(dim=1)[n_q:].view(-1, 2, 128)
        kw = dict(a_exp=a_exp, k_exp=scales.exp_for(1.0 + 6.0 * float(rl[:, 0].max())), v_exp=scales.exp_for(6.0 * float(rl[:, 1].max())))
        Q, part = ops.proj_qkv(xf, ops.pack_proj(dev(W), n_q, ops.SPLIT_H2, w_exp), tc, cr, cl, 0, **kw)
        Qg, partg = ops.gemm_qkv(xf, ops.pack_w(dev(W), ops.SPLIT_H2, w_exp), n_q, tc, cr, cl, 0,
                                 layout=ops.LAYOUT_A_FRAG | (ops.LAYOUT_C_FRAG if n_q else 0), **kw)
        assert (Q is None and Qg is None) or torch.equal(Q[:77], Qg[:77]) or (Q[:77] - Qg[:77]).abs().max() < 2e-6
        assert part.shape == partg.shape
        for a, b in ([(part, partg)] if part.dim() == 3 else list(zip(part, partg))):
            torch.testing.assert_close(ops.kv_finalize(a, cr, cl, 0, 0, 1, 1), ops.kv_finalize(b, cr, cl, 0, 0, 1, 1), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("split", ["h2", "x3"])
def test_batched_key_value_projection_of_several_layers_equals_one_launch_per_layer(split):
    """scream_gemm_qkv_split_f32 with N = 512 L, n_q == 0 (the cross stage's target side: the target features are frozen after
    the stem, models/pointnet.py:53-57, so the L cross layers' key/value projections read the SAME rows): the partials of
    layer l, at kv_partial + l (M/128) 8 1056, are bit for bit those of that layer's own N = 512 launch (same exponents), and
    one scream_kv_finalize_image launch over all layers writes the same images as L launches."""
    g = torch.Generator().manual_seed(17)
    L, lens, row0, rows, base = 3, [300, 129, 700], [0, 384, 640], 1408, 256
    x = torch.randn(base + rows, 256, generator=g)
    Ws = [torch.randn(512, 256, generator=g) / 16 * (1 + l) for l in range(L)]
    tiles = dev(torch.tensor([9] * 2 + [0] * 3 + [1] * 2 + [2] * 6, dtype=torch.int32))  # packed rows 256 .. are the three clouds
    crow0, clen = dev(torch.tensor([base + r for r in row0], dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    SPL = SPLITS[split]
    w_exp = scales.w_exp(torch.cat(Ws)) if split == "h2" else 0  # one exponent for the stacked matrix
    a_exp = scales.exp_for(8.0)
    xd = dev(x[base:].clamp(-8, 8))
    _, part_all = ops.gemm_qkv(xd, ops.pack_w(dev(torch.cat(Ws)), SPL, w_exp), 0, tiles, crow0, clen, base, a_exp=a_exp)
    assert part_all.shape == (L, rows // 128, 8, 1056)
    img_all = ops.kv_finalize_image(part_all, crow0, clen, base, 0, 3, 3, split=SPL)
    for l in range(L):
        _, part = ops.gemm_qkv(xd, ops.pack_w(dev(Ws[l]), SPL, w_exp), 0, tiles, crow0, clen, base, a_exp=a_exp)
        assert torch.equal(part, part_all[l])
        assert torch.equal(ops.kv_finalize_image(part, crow0, clen, base, 0, 3, 3, split=SPL), img_all[l])


@pytest.mark.parametrize("backend", ["h2", "x3"])
def test_forward_batched_cross_key_values_equals_per_layer_launches(golden, backend):
    """The forward projects the target features for all cross layers in one launch after the stem (net.batched_cross_kv,
    default) instead of once per cross layer: same arithmetic.  bf16 x 3: bit for bit.  fp16 x 2: the stacked matrix carries ONE
    power-of-two exponent (its largest element's) where each layer's own matrix carries its own, so the weights' second planes
    round differently -- fp32-rounding-level agreement, and both reproduce the reference's outputs."""
    g = golden("e2e")
    for seed, ns, nc, n, m, explicit in g["cases"]:
        if int(nc) == 0:
            continue
        center = dev(g["center_%d" % seed]) if explicit else None
        outs = {}
        for batched in (True, False):
            net = build_net(int(seed), int(ns), int(nc), backend)
            net.batched_cross_kv = batched
            outs[batched] = net(dev(g["src_%d" % seed]), dev(g["tgt_%d" % seed]), center, 1.0, False, False, None)[0]
            np.testing.assert_allclose(outs[batched].cpu().numpy(), g["out_%d" % seed], rtol=2e-4, atol=5e-5, err_msg="case seed=%d" % seed)
        if backend == "x3":
            assert torch.equal(outs[True], outs[False])
        else:
            torch.testing.assert_close(outs[True], outs[False], rtol=2e-5, atol=2e-5)


def test_forward_on_the_ring_projection_equals_the_gemm_projection(golden):
    """The forward projects q/k/v on the ring kernel (net.ring_proj, default on the fp16 splits with the fused tail; csrc/proj_ring.hip)
    or on the 8-wave GEMM (SCREAM_RING_PROJ=0 / net.ring_proj = False): the same products per element -- Q' is bit-identical -- with the
    K^T V partial of a 128-row tile added up from two 64-row halves instead of four 32-row quarters: fp32-rounding-level agreement of
    the outputs, and both reproduce the reference's."""
    g = golden("e2e")
    for seed, ns, nc, n, m, explicit in g["cases"]:
        center = dev(g["center_%d" % seed]) if explicit else None
        outs = {}
        for ring in (True, False):
            net = build_net(int(seed), int(ns), int(nc), "h2")
            net.ring_proj = ring
            outs[ring] = net(dev(g["src_%d" % seed]), dev(g["tgt_%d" % seed]), center, 1.0, False, False, None)[0]
            np.testing.assert_allclose(outs[ring].cpu().numpy(), g["out_%d" % seed], rtol=2e-4, atol=5e-5, err_msg="case seed=%d" % seed)
        torch.testing.assert_close(outs[True], outs[False], rtol=2e-5, atol=2e-5)


# ----------------------------------------------------------------- A1-A6 whole forward pass
@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_vs_reference_golden(golden, backend):
    g = golden("e2e")
    for seed, ns, nc, n, m, explicit in g["cases"]:
        net = build_net(int(seed), int(ns), int(nc), backend)
        center = dev(g["center_%d" % seed]) if explicit else None
        src_, imgs, tr = net(dev(g["src_%d" % seed]), dev(g["tgt_%d" % seed]), center, 1.0, False, False, None)
        assert imgs is None and tr is None and src_.shape == (1, n, 3)
        np.testing.assert_allclose(src_.cpu().numpy(), g["out_%d" % seed], rtol=2e-4, atol=5e-5,
                                   err_msg="case seed=%d" % seed)


@pytest.mark.parametrize("backend", ["h2", "x3"])
def test_forward_unfused_tail_vs_reference_golden(golden, backend):
    """Each split forward has one default path (projection GEMM with fused K^T V + one-launch layer tail) and one fallback:
    the unfused tail (SCREAM_FUSED_TAIL=0 / net.fused_tail = False: attention apply + merge GEMM + FFN-up + FFN-down launches,
    row-major activations, the 9 KB-per-row workspace).  Same arithmetic and operand exponents, different kernels and
    summation orders: it reproduces the reference's outputs to the default path's tolerance and agrees with the default
    path to fp32 rounding."""
    g = golden("e2e")
    for seed, ns, nc, n, m, explicit in g["cases"]:
        center = dev(g["center_%d" % seed]) if explicit else None
        outs = {}
        for name in ("default", "no_fused_tail"):
            net = build_net(int(seed), int(ns), int(nc), backend)
            net.fused_tail = name == "default"
            outs[name] = net(dev(g["src_%d" % seed]), dev(g["tgt_%d" % seed]), center, 1.0, False, False, None)[0]
        np.testing.assert_allclose(outs["no_fused_tail"].cpu().numpy(), g["out_%d" % seed], rtol=2e-4, atol=5e-5, err_msg="case seed=%d" % seed)
        torch.testing.assert_close(outs["no_fused_tail"], outs["default"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_batched_equals_single_pair(backend):
    net = build_net(5, 2, 2, backend)
    rng = np.random.default_rng(9)
    sizes = [(200, 130), (1, 300), (129, 128), (640, 5)]
    srcs = [dev(rng.uniform(-0.7, 0.7, size=(n, 3)).astype(np.float32)) for n, _ in sizes]
    tgts = [dev(rng.uniform(-0.7, 0.7, size=(m, 3)).astype(np.float32)) for _, m in sizes]
    cents = [dev(rng.uniform(-0.2, 0.2, size=3).astype(np.float32)) for _ in sizes]
    batched = net.forward_batch(srcs, tgts, cents)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    for i in range(len(sizes)):
        single = net(srcs[i][None], tgts[i][None], cents[i].view(1, 1, 3))[0][0]
        torch.testing.assert_close(batched[i], single, rtol=0, atol=0)  # same kernels, same order: bitwise
        want = O.point_transformer_forward(srcs[i][None].cpu(), tgts[i][None].cpu(), sd, cents[i].view(1, 1, 3).cpu())[0]
        torch.testing.assert_close(batched[i].cpu(), want, rtol=2e-4, atol=5e-5)


# ------------------------------------------------------------------------------ A7 1-NN
def test_nn_bit_exact_vs_reference_golden(golden):
    from scream_amd.geometry import nn_search_pair
    g, e = golden("nn"), golden("e2e")
    for i, s in enumerate(g["e2e_s"]):
        d, idx, valid = nn_search_pair(dev(e["out_23"][0]), dev(e["tgt_23"][0]), float(s), 0.1)
        np.testing.assert_array_equal(idx.cpu().numpy(), g["e2e_idx_%d" % i])
        np.testing.assert_array_equal(d.cpu().numpy(), g["e2e_d_%d" % i])
    d, idx, valid = nn_search_pair(dev(g["big_src"][0]), dev(g["big_tgt"][0]), float(g["big_s"]), 0.1)
    np.testing.assert_array_equal(idx.cpu().numpy(), g["big_idx"])  # incl. duplicate targets: lowest index wins
    np.testing.assert_array_equal(d.cpu().numpy(), g["big_d"])
    np.testing.assert_array_equal(valid.cpu().numpy(), g["big_valid"])


def test_nn_bit_exact_full_size_and_packed():
    """N = M ~ 5k (BASELINE config 2 size) and a packed multi-pair call with target-range splits."""
    rng = np.random.default_rng(4)
    pairs = [(5000, 5003, 0.31), (777, 4100, 1.7), (1, 1, 0.5), (2049, 1025, 0.9)]
    q_list = [rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) for n, _, _ in pairs]
    r_list = [rng.uniform(-1, 1, size=(m, 3)).astype(np.float32) for _, m, _ in pairs]
    q_row0 = np.cumsum([0] + [((n + 127) // 128) * 128 for n, _, _ in pairs])
    r_row0 = np.cumsum([0] + [((m + 127) // 128) * 128 for _, m, _ in pairs])
    q = np.zeros((q_row0[-1], 3), np.float32)
    r = np.zeros((r_row0[-1], 3), np.float32)
    for i in range(len(pairs)):
        q[q_row0[i]: q_row0[i] + pairs[i][0]] = q_list[i]
        r[r_row0[i]: r_row0[i] + pairs[i][1]] = r_list[i]
    i32 = lambda a: dev(torch.tensor(np.asarray(a), dtype=torch.int32))
    idx, dmin, valid = ops.nn_search(dev(q), dev(r), i32(q_row0[:-1]), i32([p[0] for p in pairs]), i32(r_row0[:-1]),
                                     i32([p[1] for p in pairs]), dev(torch.tensor([p[2] for p in pairs])),
                                     max(p[0] for p in pairs), max(p[1] for p in pairs), 0.01)
    idx, dmin, valid = idx.cpu().numpy(), dmin.cpu().numpy(), valid.cpu().numpy()
    for i, (n, m, s) in enumerate(pairs):
        de, ie, _ = O.nn_search_exact(q_list[i], r_list[i], s)
        sl = slice(q_row0[i], q_row0[i] + n)
        np.testing.assert_array_equal(idx[sl], ie)
        np.testing.assert_array_equal(dmin[sl], de)
        np.testing.assert_array_equal(valid[sl].astype(bool), de < np.float32(0.01))
        assert (idx[q_row0[i] + n: q_row0[i + 1]] == -1).all() and not valid[q_row0[i] + n: q_row0[i + 1]].any()


def test_square_distance_dense_bit_exact():
    from scream_amd.geometry import square_distance
    rng = np.random.default_rng(6)
    a = torch.from_numpy(rng.uniform(-1, 1, size=(2, 33, 3)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(-1, 1, size=(2, 300, 3)).astype(np.float32))
    torch.testing.assert_close(square_distance(dev(a), dev(b)).cpu(), O.square_distance(a, b), rtol=0, atol=0)


# --------------------------------------------------------------------------- A9/A10 Kabsch
def test_rigid_transform_vs_reference_golden(golden):
    from scream_amd.geometry import rigid_transform_3d
    g = golden("kabsch")
    for name in g["names"]:
        name = str(name)
        w = dev(g[name + "_w"].copy()) if name + "_w" in g else None
        thr = float(g[name + "_thr"]) if name + "_thr" in g else 0
        T = rigid_transform_3d(dev(g[name + "_A"]), dev(g[name + "_B"]), w, thr).cpu().numpy()
        # BASELINE.json north_star: R|t within 1e-4 Frobenius of the reference CPU path
        assert np.linalg.norm(T - g[name + "_T"]) < (1e-4 if name != "k3" else 5e-4), name
        R = T[:, :3, :3].astype(np.float64)
        np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.broadcast_to(np.eye(3), R.shape), atol=1e-6)
        np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-6)
        if w is not None:  # sub-threshold weights are zeroed in the caller's tensor (utils.py:151)
            assert (w.cpu().numpy()[g[name + "_w"] < thr] == 0).all()
    np.testing.assert_array_equal(rigid_transform_3d(dev(g["k0_A"]), dev(g["k0_B"])).cpu().numpy()[0], np.eye(4, dtype=np.float32))


def test_kabsch_degenerate_inputs_stay_proper_rotations():
    from scream_amd.geometry import rigid_transform_3d
    rng = np.random.default_rng(8)
    for K in (1, 2):  # rank-deficient H: rotation not unique in the reference either; must stay finite and proper
        A = torch.from_numpy(rng.uniform(-1, 1, size=(1, K, 3)).astype(np.float32))
        B = torch.from_numpy(rng.uniform(-1, 1, size=(1, K, 3)).astype(np.float32))
        T = rigid_transform_3d(dev(A), dev(B)).cpu().numpy()[0].astype(np.float64)
        assert np.isfinite(T).all()
        np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-6)
        np.testing.assert_allclose(np.linalg.det(T[:3, :3]), 1.0, atol=1e-6)


def test_transformation_error_vs_reference_golden(golden):
    from scream_amd.geometry import transformation_error
    g = golden("pose_metrics")
    P = g["poses"]
    n = len(P)
    Tp = dev(np.repeat(P, n, axis=0))
    Tg = dev(np.tile(P, (n, 1, 1)))
    re, te = ops.transformation_error_batched(Tp, Tg)
    np.testing.assert_allclose(re.cpu().numpy().reshape(n, n), g["re"], atol=0.03)  # acos near 1 is ill conditioned in fp32
    np.testing.assert_allclose(te.cpu().numpy().reshape(n, n), g["te"], rtol=1e-6, atol=1e-7)
    r0, t0 = transformation_error(dev(P[0]), dev(P[3]))
    assert r0.dim() == 0 and abs(r0.item() - g["re"][0, 3]) < 1e-3 and abs(t0.item() - g["te"][0, 3]) < 1e-6


# --------------------------------------------------- whole pair A1-A10 at BASELINE config-2 size
def test_register_pairs_full_size_vs_oracle():
    """Two synthetic 3DMatch-like pairs (~5k points, voxel 0.0625) through forward -> 1-NN -> Kabsch -> RE/TE,
    batched, against the oracle per pair.  src_pred is checked to tolerance; the discrete stage is checked
    bit-exactly GIVEN the device src_pred (arg-min of near ties is only defined for
    identical inputs), and R|t to 1e-4 Frobenius on the device correspondences."""
    from scream_amd.data import normalize_pair
    from scream_amd.geometry import register_batch
    from scream_amd.packing import PackedBatch
    net = build_net(0, 6, 6)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    items = [normalize_pair(*make_3dmatch_pair(seed)[:3]) for seed in (1, 2)]
    srcs = [dev(it[0]) for it in items]
    tgts = [dev(it[1]) for it in items]
    cents = [dev(it[3].reshape(3)) for it in items]  # src_center = trans^T (evaluate_3d_match.py:84)
    batch = PackedBatch.from_pairs(srcs, tgts, cents)
    src_pred = net.forward_packed(batch)
    # GT-like prediction so that the threshold keeps a realistic number of correspondences
    s = dev(torch.tensor([it[4] for it in items], dtype=torch.float32))
    c = dev(torch.stack([it[5] for it in items]))
    for i, it in enumerate(items):
        n = it[0].shape[0]
        want = O.point_transformer_forward(it[0][None], it[1][None], sd, it[3].reshape(1, 1, 3))[0]
        got = batch.unpack_src(src_pred)[i].cpu()
        assert got.shape == (n, 3)
        torch.testing.assert_close(got, want, rtol=5e-4, atol=1e-4)
    # replace the (random-weight) prediction by registered src + 1 cm noise: realistic K for gather/Kabsch
    rng = np.random.default_rng(0)
    pred = torch.zeros_like(src_pred)
    for i, it in enumerate(items):
        reg = (it[2] @ it[0].T + it[3]).T + torch.from_numpy(rng.normal(scale=0.01 * it[4], size=it[0].shape).astype(np.float32))
        pred[int(batch.cloud_row0_host[i]): int(batch.cloud_row0_host[i]) + it[0].shape[0]] = dev(reg)
    T, n_corr, idx, dmin, valid = register_batch(batch, pred, s, c, 0.1, "tgt")
    for i, it in enumerate(items):
        n = it[0].shape[0]
        sl = slice(int(batch.cloud_row0_host[i]), int(batch.cloud_row0_host[i]) + n)
        p_i = pred[sl].cpu()
        d_o, idx_o, valid_o = O.nn_search(p_i[None], it[1][None], it[4], 0.1)
        np.testing.assert_array_equal(idx[sl].cpu().numpy(), idx_o.numpy())
        np.testing.assert_array_equal(dmin[sl].cpu().numpy(), d_o.numpy())
        np.testing.assert_array_equal(valid[sl].cpu().numpy().astype(bool), valid_o.numpy())
        assert int(n_corr[i]) == int(valid_o.sum()) > n // 10
        A, Bc = O.gather_correspondences(it[0][None], it[1][None], p_i[None], idx_o, valid_o, it[4], it[5], "tgt")
        T_o = O.rigid_transform_3d(A, Bc)[0]
        assert torch.linalg.norm(T[i].cpu() - T_o).item() < 1e-4
        Tgt = O.gt_pose_metric(it[2], it[3], it[4], it[5])
        re_o, te_o = O.transformation_error(T_o, Tgt)
        re, te = ops.transformation_error_batched(T[i:i + 1].contiguous(), dev(Tgt)[None].contiguous())
        assert abs(re.item() - re_o.item()) < 0.05 and abs(te.item() - te_o.item()) < 1e-4


def test_large_cloud_40k_points_forward_and_search():
    """Robustness at sizes beyond KITTI (BASELINE configs[4] direction): one pair of 40k-point clouds through a
    1+1-layer model; forward against the oracle, the search bit-exact against the chunked numpy model."""
    from scream_amd.geometry import nn_search_pair
    from scream_amd.synthetic import make_uniform_pair
    from scream_amd.data import normalize_pair
    net = build_net(8, 1, 1)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    src, tgt, rot, trans, s, c = normalize_pair(*make_uniform_pair(1, 40000, 39000))
    out = net(dev(src)[None], dev(tgt)[None], dev(trans.reshape(1, 1, 3)), s)[0][0].cpu()
    want = O.point_transformer_forward(src[None], tgt[None], sd, trans.reshape(1, 1, 3))[0]
    torch.testing.assert_close(out, want, rtol=5e-4, atol=1e-4)
    reg = (rot @ src.T + trans).T
    d, idx, valid = nn_search_pair(dev(reg), dev(tgt), s, 0.1)
    de, ie, _ = O.nn_search_exact(reg.numpy(), tgt.numpy(), s, chunk=512)
    np.testing.assert_array_equal(idx.cpu().numpy(), ie)
    np.testing.assert_array_equal(d.cpu().numpy(), de)


@pytest.mark.parametrize("backend", BACKENDS)
def test_dem_transformer_vs_reference_golden(golden, backend):
    """DEMTransformer (models/pointnet.py:103-167) -- separate stem weights per cloud, raw coordinates
    embedded -- on the same kernels; also the Chamfer term of evaluate_open_gf.py:25-41 through the fused search."""
    from models.pointnet import DEMTransformer
    from scream_amd.geometry import chamfer_distance
    g = golden("dem")
    for seed, ns, nc, n, m in g["cases"]:
        net = DEMTransformer(256, int(ns), int(nc))
        net.gemm_backend = backend
        net.load_state_dict(make_state_dict(int(seed), 256, int(ns), int(nc), dem=True), strict=True)
        net = net.to(DEV).eval()
        dem_, imgs = net(dev(g["dsm_%d" % seed]), dev(g["dem_%d" % seed]), False)
        assert imgs is None and dem_.shape == (1, n, 3)
        np.testing.assert_allclose(dem_.cpu().numpy(), g["out_%d" % seed], rtol=2e-4, atol=5e-5)
        gt = torch.from_numpy(g["dem_%d" % seed])
        dist = O.square_distance(dem_.cpu(), gt)
        want = dist.min(dim=2)[0].mean() + dist.min(dim=1)[0].mean()
        torch.testing.assert_close(chamfer_distance(dem_, dev(gt)).cpu(), want, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("split", ["h2", "x3"])
@pytest.mark.parametrize("epi", ["relu", "res_ln", "qkv"])
def test_gemm_split_persistent_blocks_many_tiles(epi, split):
    """Every block of the persistent split-GEMM grid walks >= 3 output tiles (plus a ragged last round and a trailing 128-row
    half tile): the counted waits across tile boundaries (next tile requested before the epilogue, its first barrier
    leaving the epilogue's stores in flight) must hold for every epilogue kind.  Checked against the fp32-MFMA kernel."""
    M = 256 * 256 * 3 + 37 * 256 + 128
    g = torch.Generator(device=DEV).manual_seed(11)
    A = torch.randn(M, 256, device=DEV, generator=g)
    if epi == "relu":
        W = torch.randn(256, 256, device=DEV, generator=g) / 16
        a = ops.gemm_split(A, ops.pack_w(W, SPLITS[split]), ops.EPI_RELU)
        b = ops.gemm_f32(A, W, ops.EPI_RELU)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-5)
    elif epi == "res_ln":
        W = torch.randn(256, 256, device=DEV, generator=g) / 16
        res = torch.randn(M, 256, device=DEV, generator=g)
        gam, bet = torch.randn(256, device=DEV, generator=g), torch.randn(256, device=DEV, generator=g)
        a = ops.gemm_split(A, ops.pack_w(W, SPLITS[split]), ops.EPI_RES_LN, residual=res, gamma=gam, beta=bet)
        b = ops.gemm_f32(A, W, ops.EPI_RES_LN, residual=res, gamma=gam, beta=bet)
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    else:
        W = torch.randn(768, 256, device=DEV, generator=g) / 16
        n_clouds = M // 128 // 8 + 1  # clouds of up to 8 tiles (1024 rows), the last one shorter
        tiles = M // 128
        tile_cloud = torch.arange(tiles, dtype=torch.int32, device=DEV) // 8
        crow0 = (torch.arange(n_clouds, dtype=torch.int32, device=DEV) * 1024)
        clen = torch.full((n_clouds,), 1000, dtype=torch.int32, device=DEV)
        clen[-1] = min(1000, M - int(crow0[-1]))
        qa, pa = ops.gemm_qkv(A, ops.pack_w(W, SPLITS[split]), 256, tile_cloud, crow0, clen, 0)
        qb, pb = ops.gemm_qkv(A, W, 256, tile_cloud, crow0, clen, 0)
        torch.testing.assert_close(qa, qb, rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(pa, pb, rtol=1e-4, atol=2e-3)  # sums of 128 products of O(1) terms


def test_gemm_split_two_stream_soak_short():
    """Six seconds of tools/gemm_soak.py: random shapes and epilogues of the split GEMM (both splits) against the fp32-MFMA GEMM on two
    HIP streams at once (the lanes configuration).  The long form of this test is what caught a counted wait that was
    unsound across the epilogue's stores (once in ~10^5 launches); the short form guards against coarser mistakes."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_soak.py"), "6", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shift", [40, -40, 100, -100])
def test_gemm_bf3_is_exactly_scale_invariant_at_full_size(shift):
    """The bf16 x 3 split GEMM has no range caveat: bf16 keeps fp32's exponent, the three planes of 2^k * x are 2^k times the
    planes of x, products and fp32 accumulation scale with them -- so gemm(2^k A) == 2^k gemm(A) BIT FOR BIT (as long
    as nothing leaves the normal fp32 range), here at the row count of the headline workload and |k| up to 100."""
    g = torch.Generator(device=DEV).manual_seed(5)
    M = 333184
    A = torch.randn(M, 256, device=DEV, generator=g)
    W = torch.randn(256, 256, device=DEV, generator=g) / 16
    base = ops.gemm_split(A, ops.pack_w(W, ops.SPLIT_BF3))
    half = shift // 2
    scaled = ops.gemm_split(A * (2.0 ** half), ops.pack_w(W * (2.0 ** (shift - half)), ops.SPLIT_BF3))
    assert torch.equal(scaled * (2.0 ** -half) * (2.0 ** -(shift - half)), base)
    assert torch.isfinite(scaled).all()


def test_gemm_h2_stated_operand_range():
    """The fp16 x 2 split replaces scale INVARIANCE by a stated RANGE (split.h, include/scream_hip.h): with the exponent fixed
    by a bound B on the operand (|a| 2^a_exp <= 2^15),
      (1) a power of two moved between the data and its exponent changes nothing: gemm(2^k A; a_exp - k) == 2^k gemm(A; a_exp)
          bit for bit, |k| <= 40, at the row count of the headline workload;
      (2) data anywhere down to 2^-12 of the bound keep the full fp32-level accuracy of the split (<= 4e-7 normwise against
          float64, the threshold of test_split_gemm_is_fp32_accurate_against_float64);
      (3) below that the error is bounded in ABSOLUTE terms by the second plane's fp16 subnormal spacing: per product term
          2^-25 / 2^a_exp = B 2^-40 -- rows 2^-20 and 2^-26 below the bound lose relative accuracy, never more than that."""
    g = torch.Generator(device=DEV).manual_seed(5)
    M = 333184
    A = torch.randn(M, 256, device=DEV, generator=g)
    W = torch.randn(256, 256, device=DEV, generator=g) / 16
    Wp = ops.pack_w(W, ops.SPLIT_H2)
    e = scales.exp_for(float(A.abs().max()))
    base = ops.gemm_split(A, Wp, a_exp=e)
    for k in (40, -40, 7):
        moved = ops.gemm_split(A * (2.0 ** k), Wp, a_exp=e - k)
        assert torch.equal(moved * (2.0 ** -k), base) and torch.isfinite(moved).all()
    bound = 8.0  # as if a LayerNorm bound: a_exp = 12
    a_exp = scales.exp_for(bound)
    A1 = A[:4096].cpu()
    W64, Wabs = W.cpu().double(), W.cpu().double().abs()
    for j in (0, 4, 8, 12, 20, 26):
        Aj = A1 * (bound / 8.0) * 2.0 ** -j  # largest values ~ B 2^-j / 2
        out = ops.gemm_split(dev(Aj), Wp, a_exp=a_exp).cpu().double()
        C64 = Aj.double() @ W64.t()
        err = (out - C64).abs()
        if j <= 12:
            assert float((err / (Aj.double().abs() @ Wabs.t())).max()) <= 4e-7, j
        # absolute: 256 terms, each off by at most |w| (2^-25 / 2^a_exp) from the subnormal plane, plus the fp32-level part
        assert bool((err <= 256 * Wabs.max() * 2.0 ** -25 / 2.0 ** a_exp + 4e-7 * (Aj.double().abs() @ Wabs.t())).all()), j


def test_nn_search_at_65536_points_finds_a_permuted_copy():
    """BASELINE config 5 size (65 536 points per cloud): the target cloud is a seeded permutation of the query cloud
    plus a far-away decoy half, so the exact nearest neighbour of query i is known without an N x M matrix: the index
    where the permutation put point i -- for every one of the 65 536 queries, all within the threshold."""
    g = torch.Generator().manual_seed(9)
    n = 65536
    cells = torch.stack(torch.meshgrid(*[torch.arange(41)] * 3, indexing="ij"), dim=-1).reshape(-1, 3)[torch.randperm(41 ** 3, generator=g)[:n]]
    q = (cells.float() + 0.5 + (torch.rand(n, 3, generator=g) - 0.5) * 0.4) / 20.5 - 1  # jittered grid: neighbours >= 0.029 apart
    perm = torch.randperm(n, generator=g)
    decoy = torch.rand(n // 2, 3, generator=g) * 2 + 5.0
    r = torch.cat([q[perm], decoy])[torch.randperm(n + n // 2, generator=g)]
    # where did query i go?  (r is a shuffle of [q[perm] | decoy]; match by exact coordinates through a hash of the bits)
    key = lambda t: (t.view(torch.int32).to(torch.int64) * torch.tensor([1, 1 << 21, 1 << 42])).sum(dim=1)
    order = torch.argsort(key(r))
    pos = order[torch.searchsorted(key(r)[order], key(q))]
    assert torch.equal(r[pos], q)
    i32 = lambda a: dev(torch.tensor(a, dtype=torch.int32))
    idx, dmin, valid = ops.nn_search(dev(q), dev(r), i32([0]), i32([n]), i32([0]), i32([r.shape[0]]), dev(torch.tensor([0.37])),
                                     n, r.shape[0], 1e-4)
    assert torch.equal(idx.cpu().long(), pos)
    assert valid.cpu().bool().all() and float(dmin.abs().max()) < 2e-5  # |a|^2 <= 22 in units of s: cancellation noise only


def test_kabsch_round_trip_at_65536_correspondences():
    """Encode -> decode at the largest size: B = R A + t for a batch of known rigid motions and 65 536 correspondences
    each; the solve must give R | t back within the north-star tolerance (1e-4 Frobenius), and composing it with the
    inverse motion must give the identity."""
    g = torch.Generator().manual_seed(21)
    bs, K = 4, 65536
    A = torch.rand(bs, K, 3, generator=g) * 2 - 1
    Q, _ = torch.linalg.qr(torch.randn(bs, 3, 3, generator=g))
    R = Q * torch.sign(torch.linalg.det(Q)).view(bs, 1, 1)   # proper rotations
    t = torch.randn(bs, 3, 1, generator=g)
    B = (R @ A.transpose(1, 2) + t).transpose(1, 2).contiguous()
    T = ops.rigid_transform_3d_dense(dev(A), dev(B), None, 0.0).cpu()
    want = torch.eye(4).repeat(bs, 1, 1)
    want[:, :3, :3], want[:, :3, 3:] = R, t
    assert float((T - want).flatten(1).norm(dim=1).max()) < 1e-4
    back = torch.linalg.inv(want.double()) @ T.double()
    assert float((back - torch.eye(4, dtype=torch.float64)).flatten(1).norm(dim=1).max()) < 1e-4


# ------------------------------------------------------------------------------------- fused layer tail (tail_split.hip)
@pytest.mark.parametrize("split", ["h2", "x3"])
@pytest.mark.parametrize("cross", [False, True])
def test_fused_layer_tail_vs_oracle_block(cross, split):
    """One whole MHAttention block (models/transformer.py:74-90) on the fused path: q/k/v projection GEMM (K^T V in its
    epilogue) -> scream_kv_finalize_image -> scream_layer_tail_f32 (apply, merge + norm1, FFN + norm2 in ONE launch; att,
    m1 and the hidden activations never reach memory) against the oracle block, on ragged clouds with padding rows, for
    a self block (three clouds, several row tiles per block) and a cross block (queries and keys from different clouds)."""
    sd = make_state_dict(21, 256, 1, 1)
    pre = "cross.1.layer." if cross else "stem.0."
    rng = np.random.default_rng(5)
    lens = [300, 129, 700]
    row0 = [0, 384, 640]
    rows = 640 + 768
    x = torch.zeros(rows, 256)
    xs = [torch.from_numpy(rng.normal(size=(n, 256)).astype(np.float32)) for n in lens]
    for r0, xc in zip(row0, xs):
        x[r0:r0 + xc.shape[0]] = xc
    x[300:384] = 3.0  # garbage in padding rows must not reach the K^T V reduction
    tiles = torch.tensor([0] * 3 + [1] * 2 + [2] * 6, dtype=torch.int32)
    crow0, clen = dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    q, k, v = (sd[pre + "%s_proj.weight" % n] for n in "qkv")
    Wkv = torch.cat([k[:128], v[:128], k[128:], v[128:]], dim=0)
    SPL = SPLITS[split]
    pk = lambda w: ops.pack_w(dev(w), SPL)
    v_absmax = float((x @ v.t()).abs().max()) * 1.01  # what bounds the attention output (scales.py derives it from the weights)
    img = ops.pack_tail(dev(sd[pre + "merge.weight"]), dev(sd[pre + "mlp.0.weight"]), dev(sd[pre + "mlp.2.weight"]), SPL,
                        tail_exps_for(sd, pre, v_absmax, float((x @ q.t()).abs().max()) * 1.01))
    g1, b1, g2, b2 = (dev(sd[pre + n]) for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    xd = dev(x)
    xf = ops.act_layout(xd, True)  # the fused path passes activations FRAGMENT-major between its kernels
    FR = ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG
    assert torch.equal(ops.act_layout(xf, False), xd)
    if not cross:
        Q, part = ops.gemm_qkv(xf, pk(torch.cat([q, Wkv], dim=0)), 256, dev(tiles), crow0, clen, 0, FR)
        Qr, part_r = ops.gemm_qkv(xd, pk(torch.cat([q, Wkv], dim=0)), 256, dev(tiles), crow0, clen, 0)
        # same numbers, two layouts.  bf16 x 3: bit for bit.  fp16 x 2 computes a fragment-major query tile TRANSPOSED (weights as
        # the MFMA's first operand, so the accumulators ARE the fragment-major layout, gemm_split.hip): the same products, summed
        # by the matrix unit in another internal order -- fp32-rounding-level agreement; the key/value tiles are untouched.
        if split == "x3":
            assert torch.equal(ops.act_layout(Q, False), Qr)
        else:
            torch.testing.assert_close(ops.act_layout(Q, False), Qr, rtol=2e-6, atol=2e-6)
        assert torch.equal(part, part_r)
        kvi = ops.kv_finalize_image(part, crow0, clen, 0, 0, 3, 3, split=SPL)
        y = ops.act_layout(ops.layer_tail(Q, kvi, dev(tiles), 0, clen, xf, img, g1, b1, g2, b2), False).cpu()
        for r0, xc in zip(row0, xs):
            want = O.mh_attention(xc[None], xc[None], xc[None], sd, pre)[0]
            torch.testing.assert_close(y[r0:r0 + xc.shape[0]], want, rtol=2e-4, atol=5e-5)
    else:
        # queries: cloud 0 (rows 0..383); keys/values: cloud 2 (rows 640..) -- "source attends to target", kv_cloud_offset 2
        Q = ops.gemm_split(xf[:384], pk(q), ops.EPI_ELU1, n_act=256, layout=FR)
        _, part = ops.gemm_qkv(xf[640:], pk(Wkv), 0, dev(tiles), crow0, clen, 640, ops.LAYOUT_A_FRAG)
        kvi = ops.kv_finalize_image(part, crow0, clen, 640, 2, 1, 3, split=SPL)
        y = ops.act_layout(ops.layer_tail(Q, kvi, dev(tiles[:3].contiguous()), 2, clen, xf[:384], img, g1, b1, g2, b2), False).cpu()
        want = O.mh_attention(xs[0][None], xs[2][None], xs[2][None], sd, pre)[0]
        torch.testing.assert_close(y[:300], want, rtol=2e-4, atol=5e-5)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("split", ["h2", "x3"])
def test_fused_layer_tail_many_tiles_equals_unfused_path(split):
    """33 024 rows = 258 row tiles on 256 persistent blocks (some blocks walk two tiles: the next tile's first two heads
    are applied under the current tile's last stages) in 40 ragged clouds: the fused tail against the kernels it
    replaces (attn_apply + merge GEMM + FFN-up + FFN-down), same split arithmetic and exponents, different summation order."""
    g_ = torch.Generator().manual_seed(3)
    n_clouds, rows = 40, 33024
    bounds = torch.linspace(0, rows // 128, n_clouds + 1).round().int()
    lens, row0, tiles = [], [], []
    for c in range(n_clouds):
        t0, t1 = int(bounds[c]), int(bounds[c + 1])
        row0.append(t0 * 128)
        lens.append((t1 - t0) * 128 - int(torch.randint(0, 127, (1,), generator=g_)))
        tiles += [c] * (t1 - t0)
    x = torch.randn(rows, 256, generator=g_)
    sd = make_state_dict(4, 256, 1, 1)
    pre = "stem.0."
    q, k, v = (sd[pre + "%s_proj.weight" % n] for n in "qkv")
    W = torch.cat([q, k[:128], v[:128], k[128:], v[128:]], dim=0)
    tc, crow0, clen = dev(torch.tensor(tiles, dtype=torch.int32)), dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    g1, b1, g2, b2 = (dev(sd[pre + n]) for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    xd = dev(x)
    xf = ops.act_layout(xd, True)
    SPL = SPLITS[split]
    pk = lambda w: ops.pack_w(dev(w), SPL)
    Qf, part = ops.gemm_qkv(xf, pk(W), 256, tc, crow0, clen, 0, ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG)
    Q = ops.act_layout(Qf, False)
    ex = tail_exps_for(sd, pre, float((x @ v.t()).abs().max()) * 1.01, float((x @ q.t()).abs().max()) * 1.01)
    img = ops.pack_tail(dev(sd[pre + "merge.weight"]), dev(sd[pre + "mlp.0.weight"]), dev(sd[pre + "mlp.2.weight"]), SPL, ex)
    kvi = ops.kv_finalize_image(part, crow0, clen, 0, 0, n_clouds, n_clouds, split=SPL)
    yf = ops.layer_tail(Qf, kvi, tc, 0, clen, xf, img, g1, b1, g2, b2)
    y = ops.act_layout(yf, False)
    kv = ops.kv_finalize(part, crow0, clen, 0, 0, n_clouds, n_clouds)
    att = ops.attn_apply(Q, 256, kv, tc, 0, clen, rows)
    m1 = ops.gemm_split(att, pk(sd[pre + "merge.weight"]), ops.EPI_RES_LN, residual=xd, gamma=g1, beta=b1, a_exp=ex.e_att)
    hid = ops.gemm_split(m1, pk(sd[pre + "mlp.0.weight"]), ops.EPI_RELU, a_exp=ex.e_m1)
    want = ops.gemm_split(hid, pk(sd[pre + "mlp.2.weight"]), ops.EPI_RES_LN, residual=xd, gamma=g2, beta=b2, a_exp=ex.e_h)
    valid = torch.zeros(rows, dtype=torch.bool)
    for r0, n in zip(row0, lens):
        valid[r0:r0 + n] = True
    torch.testing.assert_close(y.cpu()[valid], want.cpu()[valid], rtol=1e-4, atol=2e-5)
    assert torch.equal(ops.layer_tail(Qf, kvi, tc, 0, clen, xf, img, g1, b1, g2, b2), yf)  # deterministic


@pytest.mark.parametrize("cross", [False, True])
def test_layer_tail_with_its_own_query_projection(cross):
    """tail_kernel<SplitH2, QF>: the image carries THIS layer's Wq in front and every tile begins with Q' = elu(x Wq^T) + 1 of its own
    rows, kept in registers for the applies (no Q' in memory; scream_layer_tail_f32 with Q == NULL) -- against the kernel that reads the
    projection's Q' (same products, another summation order inside the matrix unit: fp32-rounding-level agreement of y) and against the
    oracle block, 258 row tiles on 256 persistent blocks, ragged clouds; a cross block takes its keys from other clouds."""
    g_ = torch.Generator().manual_seed(9)
    n_clouds, rows = 24, 33024
    bounds = torch.linspace(0, rows // 128, n_clouds + 1).round().int()
    lens, row0, tiles = [], [], []
    for c in range(n_clouds):
        t0, t1 = int(bounds[c]), int(bounds[c + 1])
        row0.append(t0 * 128)
        lens.append((t1 - t0) * 128 - int(torch.randint(0, 127, (1,), generator=g_)))
        tiles += [c] * (t1 - t0)
    x = torch.randn(rows, 256, generator=g_)
    sd = make_state_dict(8, 256, 1, 1)
    pre = "stem.0."
    q, k, v = (sd[pre + "%s_proj.weight" % n] for n in "qkv")
    W = torch.cat([q, k[:128], v[:128], k[128:], v[128:]], dim=0)
    tc, crow0, clen = dev(torch.tensor(tiles, dtype=torch.int32)), dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    g1, b1, g2, b2 = (dev(sd[pre + n]) for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    xf = ops.act_layout(dev(x), True)
    SPL = ops.SPLIT_H2
    a_exp = scales.exp_for(float(x.abs().max()))
    Qf, part = ops.gemm_qkv(xf, ops.pack_w(dev(W), SPL), 256, tc, crow0, clen, 0, ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG, a_exp=a_exp)
    kvi = ops.kv_finalize_image(part, crow0, clen, 0, 0, n_clouds, n_clouds, split=SPL)
    off = 0
    if cross:  # every tile attends to the NEXT cloud's keys (kv_cloud_offset 1; the last cloud wraps onto an extra copy of image 0)
        kvi = torch.cat([kvi, kvi[:1]])
        clen = torch.cat([clen, clen[:1]])
        off = 1
    exd = scales.tail_exps(sd[pre + "merge.weight"], sd[pre + "mlp.0.weight"], sd[pre + "mlp.2.weight"], sd[pre + "norm1.weight"],
                           sd[pre + "norm1.bias"], float((x @ v.t()).abs().max()) * 1.01, float((x @ q.t()).abs().max()) * 1.01)
    mats = [dev(sd[pre + n]) for n in ("merge.weight", "mlp.0.weight", "mlp.2.weight")]
    plain = ops.pack_tail(*mats, SPL, ops.tail_exps(**exd))
    y_plain = ops.layer_tail(Qf, kvi, tc, off, clen, xf, plain, g1, b1, g2, b2)
    own = ops.pack_tail(*mats, SPL, ops.tail_exps(e_x=a_exp, e_wq=scales.w_exp(q), **exd), Wq_own=dev(q))
    assert own.q_first and own.data.numel() == plain.data.numel() + 8 * 2 * 16 * 1024
    y_own = ops.layer_tail(None, kvi, tc, off, clen, xf, own, g1, b1, g2, b2)
    valid = torch.zeros(rows, dtype=torch.bool)
    for r0, n in zip(row0, lens):
        valid[r0:r0 + n] = True
    a, b = ops.act_layout(y_own, False).cpu(), ops.act_layout(y_plain, False).cpu()
    torch.testing.assert_close(a[valid], b[valid], rtol=2e-5, atol=2e-5)
    assert torch.equal(ops.layer_tail(None, kvi, tc, off, clen, xf, own, g1, b1, g2, b2), y_own)  # deterministic
    if not cross:  # and the oracle block on one cloud
        r0, n = row0[3], lens[3]
        want = O.mh_attention(x[r0:r0 + n][None], x[r0:r0 + n][None], x[r0:r0 + n][None], sd, pre)[0]
        torch.testing.assert_close(a[r0:r0 + n], want, rtol=2e-4, atol=5e-5)
    from scream_amd._lib import ScreamHipError
    with pytest.raises(ScreamHipError):  # the bf16 split has no such image
        ops.pack_tail(*mats, ops.SPLIT_BF3, None, Wq_own=dev(q))


def test_layer_tail_with_the_next_layers_query_projection():
    """tail_kernel<SplitH2, NQ>: the image carries Wq of the NEXT layer and every tile ends with Q'_next = elu(y Wq^T) + 1 written
    over its own rows of Q (258 row tiles on 256 persistent blocks, ragged clouds).  y is bit-identical to the plain kernel's;
    Q'_next equals the projection GEMM of that y (same products, another summation order inside the matrix unit); the bf16 split
    refuses the image."""
    g_ = torch.Generator().manual_seed(5)
    n_clouds, rows = 24, 33024
    bounds = torch.linspace(0, rows // 128, n_clouds + 1).round().int()
    lens, row0, tiles = [], [], []
    for c in range(n_clouds):
        t0, t1 = int(bounds[c]), int(bounds[c + 1])
        row0.append(t0 * 128)
        lens.append((t1 - t0) * 128 - int(torch.randint(0, 127, (1,), generator=g_)))
        tiles += [c] * (t1 - t0)
    x = torch.randn(rows, 256, generator=g_)
    sd = make_state_dict(6, 256, 1, 1)
    pre = "stem.0."
    q, k, v = (sd[pre + "%s_proj.weight" % n] for n in "qkv")
    W = torch.cat([q, k[:128], v[:128], k[128:], v[128:]], dim=0)
    Wq_next = sd["cross.1.layer.q_proj.weight"]
    tc, crow0, clen = dev(torch.tensor(tiles, dtype=torch.int32)), dev(torch.tensor(row0, dtype=torch.int32)), dev(torch.tensor(lens, dtype=torch.int32))
    g1, b1, g2, b2 = (dev(sd[pre + n]) for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    xf = ops.act_layout(dev(x), True)
    SPL = ops.SPLIT_H2
    Qf, part = ops.gemm_qkv(xf, ops.pack_w(dev(W), SPL), 256, tc, crow0, clen, 0, ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG)
    kvi = ops.kv_finalize_image(part, crow0, clen, 0, 0, n_clouds, n_clouds, split=SPL)
    exd = scales.tail_exps(sd[pre + "merge.weight"], sd[pre + "mlp.0.weight"], sd[pre + "mlp.2.weight"], sd[pre + "norm1.weight"],
                           sd[pre + "norm1.bias"], float((x @ v.t()).abs().max()) * 1.01, float((x @ q.t()).abs().max()) * 1.01)
    plain = ops.pack_tail(dev(sd[pre + "merge.weight"]), dev(sd[pre + "mlp.0.weight"]), dev(sd[pre + "mlp.2.weight"]), SPL, ops.tail_exps(**exd))
    y_plain = ops.layer_tail(Qf, kvi, tc, 0, clen, xf, plain, g1, b1, g2, b2)
    # y is a LayerNorm2 output: bounded by its gamma / beta, which is where the forward takes e_y from
    e_y, e_wq = scales.exp_for(scales.ln_bound(sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])), scales.w_exp(Wq_next)
    img = ops.pack_tail(dev(sd[pre + "merge.weight"]), dev(sd[pre + "mlp.0.weight"]), dev(sd[pre + "mlp.2.weight"]), SPL,
                        ops.tail_exps(e_y=e_y, e_wq=e_wq, **exd), Wq_next=dev(Wq_next))
    assert img.next_q and img.data.numel() == 80 * 32 * 1024
    from scream_amd._lib import ScreamHipError
    valid = torch.zeros(rows, dtype=torch.bool)
    for r0, n in zip(row0, lens):
        valid[r0:r0 + n] = True
    q_sep = torch.full_like(Qf, float("nan"))  # a buffer of its own
    y = ops.layer_tail(Qf, kvi, tc, 0, clen, xf, img, g1, b1, g2, b2, q_next=q_sep)
    assert torch.equal(y, y_plain)
    want = ops.gemm_split(y, ops.pack_w(dev(Wq_next), SPL, e_wq), ops.EPI_ELU1, n_act=256, layout=ops.LAYOUT_A_FRAG | ops.LAYOUT_C_FRAG, a_exp=e_y)
    torch.testing.assert_close(ops.act_layout(q_sep, False).cpu()[valid], ops.act_layout(want, False).cpu()[valid], rtol=2e-6, atol=2e-6)
    q_inplace = Qf.clone()  # in place over Q: a tile reads its rows of Q long before it writes them
    assert torch.equal(ops.layer_tail(q_inplace, kvi, tc, 0, clen, xf, img, g1, b1, g2, b2, q_next=q_inplace), y_plain)
    assert torch.equal(ops.act_layout(q_inplace, False).cpu()[valid], ops.act_layout(q_sep, False).cpu()[valid])
    with pytest.raises(AssertionError):
        ops.layer_tail(Qf, kvi, tc, 0, clen, xf, plain, g1, b1, g2, b2, q_next=torch.empty_like(Qf))  # image without the stages
    with pytest.raises(ScreamHipError):
        ops.pack_tail(dev(sd[pre + "merge.weight"]), dev(sd[pre + "mlp.0.weight"]), dev(sd[pre + "mlp.2.weight"]), ops.SPLIT_BF3, Wq_next=dev(Wq_next))


def test_forward_with_and_without_the_fused_query_projection(golden):
    """PointTransformer.fuse_next_q: the cross-stage self layers' tails also project the next layer's queries (six launches fewer).
    Same products either way; the two forwards agree to fp32 rounding and both stay on the reference golden."""
    from scream_amd.model import PointTransformer
    sd = make_state_dict(0, 256, 2, 2)
    g_ = torch.Generator().manual_seed(2)
    src, tgt = torch.rand(700, 3, generator=g_) - 0.5, torch.rand(900, 3, generator=g_) - 0.5
    outs = {}
    for on in (True, False):
        net = PointTransformer(256, 2, 2)
        net.load_state_dict(sd)
        net = net.to("cuda:0").eval()
        net.fuse_next_q, net.q_first = on, False  # (the experimental q_first would replace the next-layer query stages)
        outs[on] = net.forward_batch([dev(src)], [dev(tgt)])[0].cpu()
        layers = net._pack_weights().layers[0]
        assert [int(L.tail_next_q) for L in layers] == ([0, 0, 1, 0, 1, 0] if on else [0] * 6)
    torch.testing.assert_close(outs[True], outs[False], rtol=2e-5, atol=2e-6)


def test_forward_with_and_without_the_tails_own_query_projection(golden):
    """PointTransformer.q_first (experimental, off by default: scream_amd/model.py): every layer tail begins with its own query projection, the self layers'
    projection launches compute key/value chunks only, the cross layers launch none on the query side.  Same products either way; the two
    forwards agree to fp32 rounding (both are held to the reference golden by test_forward_vs_reference_golden)."""
    from scream_amd.model import PointTransformer
    sd = make_state_dict(0, 256, 2, 2)
    g_ = torch.Generator().manual_seed(2)
    src, tgt = torch.rand(700, 3, generator=g_) - 0.5, torch.rand(900, 3, generator=g_) - 0.5
    outs = {}
    for on in (True, False):
        net = PointTransformer(256, 2, 2)
        net.load_state_dict(sd)
        net = net.to("cuda:0").eval()
        net.q_first = on
        outs[on] = net.forward_batch([dev(src)], [dev(tgt)])[0].cpu()
        layers = net._pack_weights().layers[0]
        assert [int(L.tail_q_first) for L in layers] == [int(on)] * 6
        assert all((L.proj_kv is not None) == on for L in layers)
    torch.testing.assert_close(outs[True], outs[False], rtol=2e-5, atol=2e-6)


def test_tail_two_stream_soak_short():
    """tools/tail_soak.py for ~15 s: random row counts / cloud partitions of the fused layer tail on two streams at once
    (the lanes configuration), each draw against the unfused chain and twice for bitwise repeatability (the long runs are
    recorded in DESIGN.md: counted waits and a DMA ring are exactly what a once-in-10^5 race hides in)."""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tail_soak.py"), "15"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tail soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
