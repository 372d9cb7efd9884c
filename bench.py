#!/usr/bin/env python3
"""Headline benchmark: registration pairs/sec on synthetic 3DMatch_test-like clouds (BASELINE.json
configs[1]: voxel 0.0625 m, ~5k points per cloud, batch-of-pairs = 32 per GPU).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment)
this process IS a rank; started bare, it stays a CPU-only parent that spawns the N ranks as child processes before
anything touches a GPU (launch_ranks), forwards rank 0's JSON line and exits non-zero if any rank failed.

One *step* = the hot path A1-A10 over one batch of 32 pairs per GPU, inputs resident
in HBM: PointTransformer forward -> thresholded 1-NN -> fused gather + Kabsch -> RE/TE (+ the all-gather
of per-pair metric rows when N > 1, the path's only collective).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the GEMM kernel (all epilogue instantiations pooled; per-instantiation rows in "by_kernel" so
                  they can be matched against profiles/*kernel_stats*): algorithmic fp32 flops (true, unpadded
                  token counts) / the time during which a GEMM launch was running, measured live with HIP events
                  on the launch streams inside the timed region (scream_trace_*).  Consecutive steps run WHOLE on two
                  alternating HIP streams (config.steps_in_flight = 2; round 3 -- rounds 1-2 split every step into two
                  concurrent lanes, --lanes 2), so launches of two steps overlap: that time is the union of the launch intervals
                  ("busy_ms_per_step"; "summed_launch_ms_per_step" and the per-launch avg_ms rows count overlapped
                  time twice and are what rocprofv3's per-kernel durations add up to; --lanes 1 makes them equal).  Peak
                  (MI355X_MICROARCH.md: 2500 TFLOP/s dense for bf16 and fp16 alike): SCREAM_GEMM=h2 (default) runs the split
                  kernels on the fp16 matrix cores with THREE MFMAs per fp32 product -> 2500 / 3 = 833.3 TFLOP/s of
                  fp32-equivalent work; SCREAM_GEMM=x3 on the bf16 matrix cores with six -> 2500 / 6 = 416.7;
                  SCREAM_GEMM=f32 runs gemm_f32_kernel against the 157.3 TFLOP/s fp32 matrix peak.
  sustained_value -- pairs/s over >= 3 s of back-to-back steps right after the timed region (the power governor needs a few
                  hundred ms to settle; `value` is the K steps the driver asked for).
  variant_registered_pred -- the same step re-timed (outside the headline region) with src_pred replaced, after the
                  forward, by GT-registered src + 1 cm noise, so that the search/gather/Kabsch stages see realistic
                  correspondence counts (random weights leave almost none): pairs/s, mean K, fraction registered.
  power        -- socket power and shader clock (rocm-smi) sampled by rank 0 beside those variant steps (>= 1 s): the
                  split GEMM runs into the 1400 W cap, which is what bounds it.
  cpu_baseline -- the CPU oracle (oracle/scream_ref.py, a PyTorch-CPU restatement validated against the
                  reference) on a bounded sample of the same pairs, on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PAIRS_PER_GPU = 32
PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_16BIT_DENSE_TFLOPS = 2500.0  # same table, bf16 and fp16; the split kernels spend 3 (fp16 x 2) or 6 (bf16 x 3) MFMAs per fp32 product
PEAK_HBM_GBPS = 8000.0  # same table: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
PEAK_FP32_VALU_TFLOPS = 78.6  # unpacked fp32 fma: 64 lanes x 2 flop per 4 cycles per SIMD, 1024 SIMDs, 2.4 GHz (the vector peak of 157.3 counts v_pk_fma_f32, which issues at half rate here)
TR_EMBED, TR_KV_REDUCE = 100, 101  # scream_trace record kinds (include/scream_hip.h)
MFMAS_PER_PRODUCT = {"h2": 3, "x3": 6, "f32": 1, "h1": 1}  # "h1": the labelled fp16 autocast mirror (SCREAM_GEMM=h1), never the headline
# context only, never `peak`: what a pure stream of v_mfma_f32_32x32x16_{bf16,f16} on random register operands sustains at
# the 1400 W socket cap (tools/ubench/mfma_energy.py: profiles/r02_ubench_mfma_energy.txt 1.84 PFLOP/s bf16 at 1.80 GHz;
# profiles/r03_ubench_mfma_energy.txt 1.63 PFLOP/s fp16 at 1.63 GHz on a box whose bf16 row read 1.76)
MEASURED_MFMA_AT_POWER_CAP_TFLOPS = {"x3": 1840.0, "h2": 1627.0, "h1": 1627.0}
GEMM_NAMES = {0: "gemm<EPI_NONE>", 1: "gemm<EPI_ELU1> (cross-layer q projection)",
              5: "q/k/v projection + fused K^T V reduce (proj_ring_kernel; gemm<EPI_QKV> with SCREAM_RING_PROJ=0 or off the fp16 splits)",
              7: "tail_kernel (attention apply, merge + LayerNorm1, FFN + LayerNorm2 in one launch; N = 256 + 2 x 1024 columns)",
              2: "gemm<EPI_RELU> (FFN 256->1024)", 3: "gemm<EPI_BIAS_RELU> (coor_mlp)",
              4: "gemm<EPI_RES_LN> (merge, FFN 1024->256 + residual + LayerNorm)",
              100: "pe_embed_ln_kernel", 101: "kv_finalize_image_kernel (kv_finalize_tiles_kernel when the layer tail is unfused)", 102: "attn_apply_kernel",
              103: "coor_head_kernel"}


WORKLOAD = "3dmatch"


def _gen(seed):
    from scream_amd.data import normalize_pair
    from scream_amd.synthetic import make_3dmatch_pair, make_kitti_pair, make_uniform_pair
    if WORKLOAD == "kitti":      # BASELINE configs[3]: voxel 0.7, ~13-16k points per cloud, bbox normalisation
        return normalize_pair(*make_kitti_pair(seed), "bbox")
    if WORKLOAD == "uniform64k":  # BASELINE configs[4]: 65 536 uniform points per cloud
        return normalize_pair(*make_uniform_pair(seed, 65536))
    src, tgt, T, idx, cov, scene = make_3dmatch_pair(seed)
    return normalize_pair(src, tgt, T)


def make_items(seeds, n_proc=0):
    """Seeded synthetic pairs; generated in worker processes BEFORE this process touches the GPU
    (--gen-procs 1 keeps it in-process, e.g. under rocprofv3 whose preload initialises the GPU first)."""
    import multiprocessing as mp
    if n_proc <= 0:
        n_proc = max(1, min(8, len(os.sched_getaffinity(0)) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    if n_proc == 1:
        return [_gen(s) for s in seeds]
    with mp.get_context("fork").Pool(n_proc) as pool:
        return pool.map(_gen, seeds)


def gemm_flops_per_pair(n, m, d=256):
    """stem 6 x 24 d^2 (N+M); src-self 6 x 24 d^2 N; cross 6 x (20 N + 4 M) d^2; coor_mlp 4 N d^2 + 6 d N."""
    return d * d * (412 * n + 168 * m) + 6 * d * n


def activation_bytes_per_pair(n, m, fused=True):
    """HBM bytes of 256-wide fp32 activations per pair.  'algorithmic': SURVEY.md 8d's lower bound, read x + write y = 2 KB
    per token and MHAttention application (6 stem applications on N + M tokens, 12 more on the N source tokens).  'design':
    what this build's kernels move -- fused path: the projection GEMM reads x and writes Q' (2 KB), the layer-tail kernel
    reads Q', x twice, writes y (4 KB), a cross layer also reads the target rows once (K/V projection); unfused x3 path
    (round 1): 18 KB per token and application."""
    algorithmic = 2048.0 * (6 * (n + m) + 12 * n)
    if fused:
        design = 1024.0 * (6 * 6 * (n + m) + 6 * 6 * n + 6 * (6 * n + m))
    else:
        design = 18432.0 * (6 * (n + m) + 12 * n)
    return algorithmic, design


def load_traffic_record():
    """profiles/r*_traffic.json written by tools/pmc_traffic.py from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this
    same command (the PMC passes cannot run inside bench.py); SCREAM_TRAFFIC_JSON overrides the newest one."""
    import glob
    path = os.environ.get("SCREAM_TRAFFIC_JSON")
    if not path:
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))
        path = cands[-1] if cands else None
    if not path or not os.path.exists(path):
        return None, None
    try:
        return json.load(open(path)), os.path.relpath(path, REPO)
    except Exception:
        return None, None


def cpu_baseline(items, sd, max_pairs, budget_s=25.0):
    import torch
    from oracle import scream_ref as O  # checker, timed here as the reported CPU baseline only
    # the GPU box exposes every host core in the affinity mask but one GPU's share of the host is 16 cores
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SCREAM_CPU_BASELINE_THREADS", "16")))
    torch.set_num_threads(cores)
    done, t_total = 0, 0.0
    with torch.no_grad():
        O.register_pair(items[0][0][None, :256], items[0][1][None, :256], items[0][2][None], items[0][3][None],
                        items[0][4], items[0][5], sd)  # warm the thread pool / allocator
        for it in items[:max_pairs]:
            t0 = time.perf_counter()
            O.register_pair(it[0][None], it[1][None], it[2][None], it[3][None], it[4], it[5], sd)
            t_total += time.perf_counter() - t0
            done += 1
            if t_total > budget_s:
                break
    return {"value": round(done / t_total, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d of the step's %d synthetic pairs, oracle/scream_ref.register_pair (A1-A10, fp32, torch-CPU %d threads), %.1f s"
                      % (done, len(items), cores, t_total)}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def rank_env(base_env, rank, world, port):
    """Environment of child rank `rank` (torchrun's contract: one process per GPU, LOCAL_RANK = device index)."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    return env


def launch_ranks(world, argv, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks as CHILDREN of this (GPU-free) process.
    Rank 0's stdout (the one JSON line) is forwarded; every rank's stderr goes to ours.  Returns the exit code:
    0 only if every rank exited 0.  A failing rank takes the others down (they would hang in the next collective)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=rank_env(os.environ, r, world, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    t_end = None if timeout_s is None else time.monotonic() + timeout_s
    rc, out0 = 0, b""
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            if pending:
                if 0 in pending:  # keep rank 0's pipe drained
                    try:
                        out0 += procs[0].communicate(timeout=0.2)[0] or b""
                    except subprocess.TimeoutExpired:
                        pass
                else:
                    time.sleep(0.2)
                if t_end is not None and time.monotonic() > t_end:
                    print("bench.py: ranks still running after %.0f s; stopping them" % timeout_s, file=sys.stderr)
                    rc = rc or 124
                    for q in pending:
                        procs[q].terminate()
                    t_end = None
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    if procs[0].stdout is not None and not procs[0].stdout.closed:
        out0 += procs[0].stdout.read() or b""
    # stdout carries the ONE JSON line; anything else a library printed there (gloo's connection banner) goes to stderr
    for line in out0.decode(errors="replace").splitlines():
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per step per GPU")
    ap.add_argument("--alternate", type=int, default=None,
                    help="N > 0: every step runs WHOLE on one of N HIP streams in turn (N steps in flight, out of phase).  The default "
                         "(2) unless --lanes is given: 1 686 pairs/s against 1 639 for --lanes 2 on the same box, "
                         "profiles/r03_bench_alternate_vs_lanes.txt")
    ap.add_argument("--lanes", type=int, default=None,
                    help="split every step's pairs into this many concurrent sub-batches that start together (scream_amd/lanes.py; "
                         "rounds 1-2's default was 2); --lanes 1 = one step at a time on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustain", dest="sustain", action="store_false", help="skip the >= 3 s sustained-rate leg")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the isolated calls behind roofline.secondary (counter passes: tools/pmc_traffic.py counts steps by the "
                         "dispatches of the embedding kernel)")
    ap.add_argument("--no-power", action="store_true",
                    help="do not sample rocm-smi beside the variant steps (under rocprofv3 its preload has initialised the GPU in "
                         "this process, and starting another program from it is refused on the GPU boxes)")
    ap.add_argument("--gen-procs", type=int, default=0, help="worker processes for synthetic data (0 = auto)")
    ap.add_argument("--workload", default="3dmatch", choices=["3dmatch", "kitti", "uniform64k"],
                    help="3dmatch = BASELINE configs[1] (the headline, default); kitti / uniform64k = configs[3] / [4] stress runs")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: every rank joins a gloo group, all-gathers its rank and rank 0 prints them")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # bare `python bench.py --gpus N`: be the launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    if args.dry_run:  # exercises launch_ranks + the rendezvous only (tests/test_host_cpu.py); no GPU, no kernels
        import torch
        import torch.distributed as tdist
        from scream_amd import dist as sdist
        if os.environ.get("SCREAM_BENCH_DRY_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        sdist.init_from_env("gloo")
        seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            tdist.all_gather(seen, torch.tensor([rank], dtype=torch.int64))
            tdist.barrier()
            tdist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [int(t.item()) for t in seen],
                              "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "steps": args.steps}), flush=True)
        return

    global WORKLOAD
    WORKLOAD = args.workload
    B = args.pairs
    items = make_items([rank * B + i for i in range(B)], args.gen_procs)  # before any GPU call (fork safety)

    import torch
    import torch.distributed as tdist
    from scream_amd import _lib, lanes, ops
    from scream_amd import dist as sdist
    from scream_amd.evaluate import gt_pose_metric
    from scream_amd.geometry import register_batch
    from scream_amd.model import PointTransformer
    from scream_amd.packing import PackedBatch
    from scream_amd.synthetic import make_state_dict

    # RCCL ("nccl") over xGMI in production.  SCREAM_BENCH_BACKEND=gloo + SCREAM_BENCH_SINGLE_DEVICE=1 exist only to
    # rehearse the multi-rank control flow on a one-GPU box (ranks then share cuda:0 and exchange rows through the CPU).
    backend = os.environ.get("SCREAM_BENCH_BACKEND", "nccl")
    rank, world, local = sdist.init_from_env(backend)
    if os.environ.get("SCREAM_BENCH_SINGLE_DEVICE", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    lib = _lib.load()

    sd = make_state_dict(0, 256, 6, 6)
    net = PointTransformer(256, 6, 6)
    net.load_state_dict(sd)
    net = net.to(dev).eval()

    # Default (--alternate 2): consecutive steps run whole on two alternating HIP streams, so two steps are in flight OUT OF
    # PHASE -- one's power-cap-bound layer tails beside the other's projections, each filling the other's partial last rounds.
    # --lanes L (rounds 1-2): the step's pairs as L concurrent sub-batches (scream_amd/lanes.py) that start together and join
    # at the step's end.  Inputs are resident per lane.
    class Lane:
        def __init__(self, its):
            self.batch = PackedBatch.from_pairs([it[0].to(dev) for it in its], [it[1].to(dev) for it in its],
                                                [it[3].reshape(3).to(dev) for it in its])
            self.s = torch.tensor([it[4] for it in its], dtype=torch.float32, device=dev)
            self.c = torch.stack([it[5] for it in its]).to(dev)
            self.T_gt = torch.stack([gt_pose_metric(it[2], it[3], it[4], it[5]) for it in its]).to(dev)
            # with random weights almost nothing passes the distance threshold, so A8/A9 see K ~ 0; the
            # "registered prediction" variant swaps in GT-registered src + 1 cm noise after the forward (K ~ overlap * N)
            reg = torch.zeros(self.batch.rows_src, 3)
            for k, it in enumerate(its):
                rng = np.random.default_rng(1000 + k)
                r0 = int(self.batch.cloud_row0_host[k])
                reg[r0:r0 + it[0].shape[0]] = (it[2] @ it[0].T + it[3]).T + torch.from_numpy(
                    rng.normal(scale=0.01 * it[4], size=it[0].shape).astype(np.float32))
            self.reg_pred = reg.to(dev)

    if args.alternate is None:
        args.alternate = 2 if args.lanes is None else 0
    if args.alternate > 0 or args.lanes is None:
        args.lanes = 1
    lane_parts = [Lane([items[i] for i in rg]) for rg in lanes.split_weighted([it[0].shape[0] + it[1].shape[0] for it in items], args.lanes)]
    alt_streams = lanes.lane_streams(dev, args.alternate) if args.alternate > 0 else []
    step_no = [0]
    src_len = [n for ln in lane_parts for n in ln.batch.src_len]
    tgt_len = [n for ln in lane_parts for n in ln.batch.tgt_len]
    rows_total = sum(ln.batch.rows_total for ln in lane_parts)
    pair_ids = torch.arange(rank * B, (rank + 1) * B, device=dev, dtype=torch.float32)
    # one set of receive buffers PER STREAM in flight: two steps on alternating streams must not all-gather into the same tensors
    gathered = [[torch.empty(B, sdist.ROW_WIDTH, device=coll_dev) for _ in range(world)] for _ in range(max(1, args.alternate))] if world > 1 else None

    dis_thresh = 1.5 if args.workload == "kitti" else 0.1  # evaluate_kitti.py:109 / evaluate_3d_match.py:178

    def lane_step(ln, trace, registered=False):
        src_pred = net.forward_packed(ln.batch, trace=trace)                                  # A1-A6
        if registered:
            src_pred = ln.reg_pred
        T, n_corr, idx, dmin, valid = register_batch(ln.batch, src_pred, ln.s, ln.c, dis_thresh)  # A7-A9
        re, te = ops.transformation_error_batched(T, ln.T_gt)                                 # A10
        return re, te, n_corr

    def step(trace=None, registered=False):
        if alt_streams:  # the whole step on the next stream; nothing joins until fence()
            k = step_no[0] % len(alt_streams)
            step_no[0] += 1
            with torch.cuda.stream(alt_streams[k]):
                return step_on_current(trace, registered, k)
        return step_on_current(trace, registered)

    def step_on_current(trace=None, registered=False, slot=0):
        outs = lanes.run(dev, lane_parts, lambda ln: lane_step(ln, trace, registered))
        re, te, n_corr = (torch.cat([o[i] for o in outs]) for i in range(3))
        if world > 1:  # the path's only exchange: per-pair metric rows
            rows = torch.zeros(B, sdist.ROW_WIDTH, device=dev)
            rows[:, 0], rows[:, 4], rows[:, 5] = pair_ids, re, te
            tdist.all_gather(gathered[slot], rows.to(coll_dev))
        return re, te, n_corr

    def fence():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    first = None
    for _ in range(args.warmup):
        o = step()
        if first is None:
            if alt_streams:
                torch.cuda.synchronize()  # the step ran on a side stream
            first = tuple(t.clone() for t in o)
    cap = 200 * len(lane_parts) * max(args.steps, 1)  # ~135 records per forward
    trace = lib.scream_trace_create(cap)
    assert trace, "scream_trace_create failed"
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step(trace)
    fence()
    elapsed = time.perf_counter() - t0
    # the timed steps did the work: every step's RE / TE / correspondence counts -- functions of the whole forward, search
    # and solve -- are finite and, the path being deterministic, bit-identical to the first warm-up step's
    checksum = None
    if last is not None:
        assert all(bool(torch.isfinite(t.float()).all()) for t in last), "non-finite outputs in the timed region"
        if first is not None:
            assert all(torch.equal(a, b) for a, b in zip(first, last)), "timed step differs from the first warm-up step"
        checksum = {"sum_re_deg": round(float(last[0].double().sum().item()), 6), "sum_te": round(float(last[1].double().sum().item()), 6),
                    "sum_correspondences": int(last[2].sum().item()), "equal_to_first_warmup_step": first is not None}
    if world > 1:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sustained rate: >= 3 s of back-to-back steps (same step, no trace), the same count on every rank -------
    n_sus = max(args.steps, int(np.ceil(3.0 / max(elapsed / max(args.steps, 1), 1e-4)))) if args.sustain else 0
    sustained = None
    if n_sus:
        fence()
        t_s = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        sus_elapsed = time.perf_counter() - t_s
        if world > 1:
            t = torch.tensor([sus_elapsed], device=coll_dev, dtype=torch.float64)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            sus_elapsed = float(t.item())
        sustained = (round(B * world * n_sus / sus_elapsed, 3), n_sus, round(sus_elapsed, 3))

    # ---- the same step with realistic correspondence counts (outside the headline's timed region) -------------
    # rank 0 also samples the socket power and the shader clock beside it: the split GEMM runs at the 1400 W cap
    n_var = min(args.steps, 5)
    power_samples, sampling = [], [rank == 0 and not args.no_power]

    def sample_power():
        import re
        import subprocess
        pat = re.compile(r"^card%d,.*\((\d+)Mhz\),\d,\(\d+Mhz\),\w,(\d+\.\d+)\s*$" % local)
        while sampling[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
                for line in out.splitlines():
                    m = pat.match(line.strip())
                    if m:
                        power_samples.append((int(m.group(1)), float(m.group(2))))
            except Exception:
                return
            time.sleep(0.02)

    import threading
    sampler = threading.Thread(target=sample_power, daemon=True)
    fence()
    sampler.start()
    t1 = time.perf_counter()
    n_run = 0
    while n_run < n_var or (sampling[0] and world == 1 and time.perf_counter() - t1 < 1.0):  # >= 1 s so rocm-smi sees it
        re_v, te_v, k_v = step(None, registered=True)
        n_run += 1
    fence()
    var_elapsed = time.perf_counter() - t1
    sampling[0] = False
    sampler.join(timeout=10)
    n_var = n_run
    steady = power_samples[len(power_samples) // 2:]  # the governor needs a few hundred ms to settle
    power = ({"socket_power_w": round(sum(p for _, p in steady) / len(steady), 1),
              "sclk_mhz": round(sum(c for c, _ in steady) / len(steady)), "samples": len(steady),
              "source": "rocm-smi --showpower --showclocks, sampled beside the variant steps"} if steady else None)
    variant = {"value": round(B * world * n_var / var_elapsed, 3), "unit": "pairs/s", "steps": n_var,
               "what": "same step, src_pred replaced after the forward by GT-registered src + 1 cm noise (SURVEY.md 8d)",
               "mean_correspondences": round(float(k_v.float().mean().item()), 1),
               "success_RE5_TE_0.3_fraction": round(float(((re_v < 5) & (te_v < 0.3)).float().mean().item()), 3)}

    # ---- the kernels north_star names a roofline for, outside the timed region: HIP events on the launch stream
    # around back-to-back calls of the 1-NN stage and of the gather + Kabsch solve on lane 0's batch (the whole step's unless
    # --lanes > 1), with the realistic correspondences of the registered-prediction variant
    def ev_ms(f, n=20):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    if args.secondary:
        ln0 = lane_parts[0]
        b0 = ln0.batch
        tgt_xyz0, tgt_row0_0 = b0.xyz[b0.rows_src:], (b0.tgt_row0 - b0.rows_src).contiguous()
        nn_call = lambda: ops.nn_search(ln0.reg_pred, tgt_xyz0, b0.src_row0, b0.src_len_dev, tgt_row0_0, b0.tgt_len_dev, ln0.s,
                                        max(b0.src_len), max(b0.tgt_len), dis_thresh)
        idx0, _, valid0 = nn_call()
        kab_call = lambda: ops.kabsch_corr(b0.xyz[: b0.rows_src], tgt_xyz0, b0.src_row0, b0.src_len_dev, tgt_row0_0, idx0, valid0, ln0.s, ln0.c)
        nn_ms, kab_ms = ev_ms(nn_call), ev_ms(kab_call)
        # the embedding and the K^T V finalize of the stem (every row of the batch) the same way: inside the timed region their
        # launches overlap the other step in flight, so the in-forward event times are not the kernels' own
        from scream_amd.model import pe_dim_t
        w_ = lambda k: sd[k].to(dev).contiguous()
        emb_args = (b0.xyz, b0.tile_cloud, b0.center, pe_dim_t().to(dev), w_("embedding.weight")[:, :, 0].contiguous(), w_("embedding.bias"),
                    w_("pre_norm.weight"), w_("pre_norm.bias"))
        emb_ms = ev_ms(lambda: ops.pe_embed_ln(*emb_args, frag=True))
        kv_part = torch.zeros(b0.rows_total // 128, 8, 1056, device=dev)
        kv_img = torch.zeros(2 * b0.n_pairs, lib.scream_kv_image_bytes(), device=dev, dtype=torch.uint8)
        kvf_ms = ev_ms(lambda: ops.kv_finalize_image(kv_part, b0.cloud_row0, b0.cloud_len, 0, 0, 2 * b0.n_pairs, 2 * b0.n_pairs, out=kv_img))
        k_corr = float(kab_call()[1].sum().item())
        nn_flop = 8.0 * sum(n * m for n, m in zip(b0.src_len, b0.tgt_len))                       # N M (3 fma + add + compare), SURVEY.md 8d
        nn_bytes = float(sum(12 * n + 12 * m + 9 * n for n, m in zip(b0.src_len, b0.tgt_len)))   # src_pred, tgt in; idx, dmin, valid out
        kab_bytes = float(17 * sum(b0.src_len) + 12 * k_corr + 64 * b0.n_pairs)                  # src + idx + valid per point, one target row per correspondence, T out


    # ---- per-kernel times recorded inside the timed region ------------------------------------
    ms = (C.c_float * cap)()
    kind = (C.c_int32 * cap)()
    mm = (C.c_int64 * cap)()
    nn = (C.c_int32 * cap)()
    kk = (C.c_int32 * cap)()
    starts = (C.c_float * cap)()
    cnt = lib.scream_trace_read(trace, cap, ms, kind, mm, nn, kk)
    assert cnt > 0, "trace empty"
    assert lib.scream_trace_read_starts(trace, cap, starts) == cnt
    lib.scream_trace_destroy(trace)
    # lanes overlap in time: the GEMM's busy time is the UNION of its launch intervals, not their sum
    iv = sorted((starts[i], starts[i] + ms[i]) for i in range(cnt) if kind[i] < 100)
    gemm_busy_ms, hi = 0.0, -1.0
    for a, b_ in iv:
        if b_ > hi:
            gemm_busy_ms += b_ - max(a, hi)
            hi = b_
    by, by_shape = {}, {}
    for i in range(cnt):
        e = by.setdefault(kind[i], {"launches": 0, "ms": 0.0, "padded_flops": 0.0})
        e["launches"] += 1
        e["ms"] += ms[i]
        e["padded_flops"] += 2.0 * mm[i] * nn[i] * kk[i] 
        if kind[i] < 100:
            e2 = by_shape.setdefault((kind[i], mm[i], nn[i], kk[i]), {"launches": 0, "ms": 0.0})
            e2["launches"] += 1
            e2["ms"] += ms[i]
    gemm_ms = sum(v["ms"] for k_, v in by.items() if k_ < 100)
    gemm_launches = sum(v["launches"] for k_, v in by.items() if k_ < 100)
    rows_true = sum(src_len) + sum(tgt_len)
    pad_eff = rows_true / rows_total
    algo_flops_step = sum(gemm_flops_per_pair(n, m) for n, m in zip(src_len, tgt_len))
    achieved = algo_flops_step * args.steps / (gemm_busy_ms * 1e-3) / 1e12
    by_kernel = []
    for k_, v in sorted(by.items(), key=lambda kv: (kv[0] >= 100, kv[0])):
        row = {"kernel": GEMM_NAMES.get(k_, str(k_)), "launches": v["launches"],
               "avg_ms": round(v["ms"] / v["launches"], 4), "share_of_step": round(v["ms"] / (elapsed * 1e3), 4)}
        if k_ < 100:
            row["tflops_algorithmic"] = round(v["padded_flops"] * pad_eff / (v["ms"] * 1e-3) / 1e12, 2)
        by_kernel.append(row)
    by_gemm_shape = [{"epilogue": k_[0], "M": int(k_[1]), "N": int(k_[2]), "K": int(k_[3]), "launches": v["launches"],
                      "avg_ms": round(v["ms"] / v["launches"], 4),
                      "tflops_padded": round(2.0 * k_[1] * k_[2] * k_[3] / (v["ms"] / v["launches"] * 1e-3) / 1e12, 1)}
                     for k_, v in sorted(by_shape.items())]

    secondary = None
    if args.secondary:
        # roofline.secondary: the HBM- / VALU-bound kernels of the step against THEIR roofs (algorithmic bytes or flops per call over
        # the HIP-event time of the call; peaks from MI355X_MICROARCH.md: HBM 8 TB/s, 78.6 TFLOP/s of unpacked fp32 fma on 256 CUs)
        def hbm_row(name, ms_call, nbytes, note):
            return {"kernel": name, "bound": "hbm", "avg_ms": round(ms_call, 4), "algorithmic_bytes_per_call": round(nbytes),
                    "achieved": round(nbytes / (ms_call * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(nbytes / (ms_call * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "note": note}
        secondary = [{"kernel": "scream_nn_search (A7: target prep + key init, nn_search_kernel, finalize)", "bound": "valu", "avg_ms": round(nn_ms, 4),
                      "algorithmic_flop_per_call": nn_flop, "achieved": round(nn_flop / (nn_ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_VALU_TFLOPS,
                      "unit": "TFLOP/s", "frac": round(nn_flop / (nn_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
                      "algorithmic_bytes_per_call": round(nn_bytes), "algorithmic_GBps": round(nn_bytes / (nn_ms * 1e-3) / 1e9, 2),
                      "note": "brute force over LDS-staged targets: N M x 8 flop over 12 (N + M) + 9 N bytes -- VALU-bound by three orders of "
                              "magnitude, not HBM-bound (SURVEY.md 8d); HIP events around back-to-back calls on the launch stream"}]
        secondary.append(hbm_row("pe_embed_ln_kernel (A1)", emb_ms, b0.rows_total * (12 + 1024.0),
                                 "12 B of coordinates in, 1 KB of features out per row; HIP events around back-to-back calls"))
        secondary.append(hbm_row("kv_finalize_image_kernel (A3 reduce, second stage; the stem's launch: every cloud of the batch)", kvf_ms,
                                 kv_part.numel() * 4.0 + kv_img.numel(),
                                 "the K^T V partials of every 128-row tile in (33 KB per tile), one operand image per cloud out: 90 MB per "
                                 "launch behind a reduction whose depth is a cloud's ~40 tiles -- latency, not bandwidth"))
        secondary.append(hbm_row("kabsch_corr_kernel (A8 + A9)", kab_ms, kab_bytes,
                                 "17 B per source point + 12 B per correspondence (%.0f per pair with the registered prediction): one workgroup "
                                 "per pair, latency-bound" % (k_corr / b0.n_pairs)))


    gb = net.gemm_backend
    split_backend = gb in ("h2", "x3", "h1")
    n_prod = MFMAS_PER_PRODUCT[gb]
    fused = split_backend and net.fused_tail
    # ---- HBM traffic: algorithmic bytes computed here, counter bytes from the committed PMC record of this command
    algo_b = sum(activation_bytes_per_pair(n, m, fused)[0] for n, m in zip(src_len, tgt_len))
    design_b = sum(activation_bytes_per_pair(n, m, fused)[1] for n, m in zip(src_len, tgt_len))
    weight_b = float(sum(p.numel() for p in net.parameters()) * {"h2": 4, "x3": 6, "f32": 4, "h1": 2}[gb])  # every weight image is read at least once per step
    trec, tpath = load_traffic_record()
    traffic = None
    if trec is not None:
        # the dominant kernel's bytes per launch: every instantiation of it pooled (the layer tail has two since round 3: with and
        # without the next layer's query stages) -- bytes per step over launches per step
        dom_k = [v for k_, v in trec["kernels"].items() if k_.startswith("tail_kernel" if fused else "gemm")]
        dom = None
        if dom_k:
            n_launch = sum(v["dispatches_fetch_pass"] for v in dom_k) / max(trec.get("steps_in_fetch_pass", 1.0), 1e-9)
            dom = {"fetch_bytes_per_launch": sum(v["fetch_bytes_per_step"] for v in dom_k) / n_launch,
                   "write_bytes_per_launch": sum(v["write_bytes_per_step"] for v in dom_k) / n_launch}
        per_launch_algo = None
        if fused:  # the layer-tail kernel by itself: Q' + y per token and application (x is re-read from the projection's pass).
            # Per launch AT THE LANE COUNT OF THE COUNTER RECORD (a launch covers 1 / lanes of the step's rows): like for like
            per_launch_algo = (2048.0 * sum(6 * (n + m) + 12 * n for n, m in zip(src_len, tgt_len)) + (1024.0 * 6 * sum(src_len) if net.fuse_next_q and gb != "x3" else 0.0)) / (18.0 * trec.get("lanes", 1))  # (+ Q' of the next layer where the tail writes it)
        traffic = {"counter_bytes_per_step": round(trec["hbm_bytes_per_step"] * B / PAIRS_PER_GPU),
                   "algorithmic_bytes_per_step": round(algo_b + weight_b), "design_bytes_per_step": round(design_b + weight_b),
                   "ratio": round(trec["hbm_bytes_per_step"] * B / PAIRS_PER_GPU / (algo_b + weight_b), 3),
                   "dominant_kernel_counter_bytes_per_launch": None if dom is None else round(dom["fetch_bytes_per_launch"] + dom["write_bytes_per_launch"]),
                   "dominant_kernel_algorithmic_bytes_per_launch": None if per_launch_algo is None else round(per_launch_algo),
                   "dominant_kernel_ratio": None if (dom is None or per_launch_algo is None) else round((dom["fetch_bytes_per_launch"] + dom["write_bytes_per_launch"]) / per_launch_algo, 3),
                   "lanes_of_counter_record": trec.get("lanes", 1), "gemm_backend_of_counter_record": trec.get("gemm_backend"),
                   "counters": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes, %d lane(s)" % trec.get("lanes", 1),
                   "source": tpath}
    peak = PEAK_16BIT_DENSE_TFLOPS / n_prod if split_backend else PEAK_FP32_MATRIX_TFLOPS
    kernel_desc = (("tail_kernel<%s> + proj_ring_kernel / gemm_split_kernel<%s> (fp32 operands split into %s, %d x v_mfma_f32_32x32x16_%s per "
                    "product, fp32 accumulate; peak = 16-bit dense 2500 TFLOP/s / %d; achieved = fp32-equivalent algorithmic GEMM "
                    "flops of SURVEY.md 8d over the time a GEMM-class launch was running; the attention-apply products and the "
                    "fused K^T V epilogue's own MFMAs are not counted)")
                   % (("SplitH2", "SplitH2", "2 fp16 planes with weight-derived power-of-two scales", 3, "f16", 3) if gb == "h2" else
                      ("SplitH1", "SplitH1", "ONE fp16 plane -- the reference's autocast mode, NOT fp32-accurate", 1, "f16", 1) if gb == "h1" else
                      ("SplitBf3", "SplitBf3", "3 bf16 planes", 6, "bf16", 6))) if split_backend else \
                  ("gemm_f32_kernel (v_mfma_f32_32x32x2_f32; all epilogue instantiations; the fused K^T V epilogue's own "
                   "MFMAs are not counted as algorithmic flops)")
    if rank == 0:
        total_pairs = B * world * args.steps
        out = {
            "metric": "registration pairs/sec", "value": round(total_pairs / elapsed, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "sustained_value": None if sustained is None else sustained[0],
            "sustained": None if sustained is None else {"steps": sustained[1], "seconds": sustained[2]},
            "vs_baseline": None, "dtype": "f16 (autocast mirror; not the fp32 path)" if gb == "h1" else "f32", "data": "synthetic",
            "config": {"workload": {"3dmatch": "BASELINE configs[1]: synthetic 3DMatch_test-like pairs, voxel 0.0625 m, ~5k points/cloud",
                                    "kitti": "BASELINE configs[3]: synthetic KITTI_test-like pairs, voxel 0.7 m, ~13-16k points/cloud",
                                    "uniform64k": "BASELINE configs[4]: 65 536 uniform points per cloud"}[args.workload]
                                   + ", batch-of-pairs=%d per GPU, A1-A10 per pair (forward 6+6 layers d_model 256, 1-NN thresh %g, "
                                     "Kabsch, RE/TE), random-init weights seed 0" % (B, dis_thresh),
                       "pairs_per_step_per_gpu": B, "lanes": len(lane_parts), "steps_in_flight": max(1, args.alternate), "mean_src_points": round(float(np.mean(src_len)), 1),
                       "mean_tgt_points": round(float(np.mean(tgt_len)), 1), "parallelism": "dp%d (pairs sharded, metric-row all-gather)" % world,
                       "rccl_ranks": tdist.get_world_size() if world > 1 else 1, "collective_backend": backend if world > 1 else None,
                       "gemm_backend": net.gemm_backend},
            "roofline": {"bound": "mfma", "kernel": kernel_desc,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "secondary": secondary,
                         "mfma_tflops_issued": round(achieved * n_prod, 1),
                         "frac_of_fp32_matrix_peak": round(achieved / PEAK_FP32_MATRIX_TFLOPS, 4),
                         **({"frac_of_measured_mfma_rate_at_power_cap": round(achieved * n_prod / MEASURED_MFMA_AT_POWER_CAP_TFLOPS[gb], 4)} if split_backend else {}),
                         "launches": gemm_launches, "avg_ms": round(gemm_ms / gemm_launches, 4),
                         "busy_ms_per_step": round(gemm_busy_ms / args.steps, 3),
                         "summed_launch_ms_per_step": round(gemm_ms / args.steps, 3),
                         "algorithmic_gflop_per_step": round(algo_flops_step / 1e9, 1),
                         "row_padding_efficiency": round(pad_eff, 4), "by_kernel": by_kernel,
                         "by_gemm_shape": by_gemm_shape},
        }
        out["output_checksum"] = checksum
        out["variant_registered_pred"] = variant
        out["power"] = power
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(items, sd, 16)
        print(json.dumps(out), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
