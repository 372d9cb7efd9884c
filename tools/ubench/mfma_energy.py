#!/usr/bin/env python3
"""Joules per flop of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 on changing register operands (random data and
all-zero data), 1 wave per SIMD on every CU: `build` here, `run` on the GPU (rocm-smi sampled beside the launches)."""
import ctypes, os, re, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "mfma_energy.so")


def build():
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mfma_energy.hip"), "-o", SO])


def run():
    import torch
    dev = "cuda:0"
    lib = ctypes.CDLL(SO)
    V = ctypes.c_void_p
    lib.mfma_burn_launch.argtypes = [ctypes.c_int, V, V, ctypes.c_int, V]
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = torch.randn(4096 * 8, device=dev, generator=g).to(torch.bfloat16)
    zero = torch.zeros(4096 * 8, device=dev, dtype=torch.bfloat16)
    # fp16 operands as the 2-plane split makes them: leading plane = values scaled to ~2^11, second plane = their rounding residuals
    v = torch.randn(4096 * 8, device=dev, generator=g) * 2048.0
    h0 = v.to(torch.float16)
    h1 = (v - h0.float()).to(torch.float16)
    rnd_h = torch.where(torch.arange(4096 * 8, device=dev) % 16 < 8, h0, h1).view(torch.bfloat16)  # alternate 16-byte pieces
    out = torch.empty(256 * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
            m = re.search(r"\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,\((\d+)Mhz\),\d,.*,(\d+\.\d+)\s*$", o.strip().splitlines()[-1])
            if m: samples.append((time.time(), int(m.group(3)), float(m.group(4))))
            time.sleep(0.05)
    th = threading.Thread(target=sampler, daemon=True); th.start()  # (daemon: a traceback in the main thread must end the process)
    iters = 2000
    flop = 256 * 4 * iters * 64 * 2.0 * 32 * 32 * 16  # blocks x waves x iterations x instruction(-pair)s x flops
    print("%-34s %9s %9s %9s %10s %12s %9s" % ("variant", "ms", "sclk MHz", "power W", "J/launch", "pJ/flop(dyn)", "PFLOP/s"))
    for tag, data in (("random operands", rnd), ("zero operands", zero)):
        for shape, name in ((0, "32x32x16"), (1, "16x16x32 (x2)"), (2, "32x32x16 f16")):
            if shape == 2 and tag.startswith("random"):
                data = rnd_h
            call = lambda: lib.mfma_burn_launch(shape, data.data_ptr(), out.data_ptr(), iters, st)
            call(); torch.cuda.synchronize(); time.sleep(0.4)
            t0 = time.time(); n = 0
            while time.time() - t0 < 2.5:
                for _ in range(5): call()
                torch.cuda.synchronize(); n += 5
            t1 = time.time()
            win = [s for s in samples if t0 + 0.5 < s[0] < t1 - 0.1]
            sclk = sum(s[1] for s in win) / max(len(win), 1); pw = sum(s[2] for s in win) / max(len(win), 1)
            ms = (t1 - t0) / n * 1e3
            print("%-34s %9.3f %9.0f %9.0f %10.3f %12.3f %9.3f" % (name + ", " + tag, ms, sclk, pw, pw * ms * 1e-3, (pw - 296.0) * ms * 1e-3 / flop * 1e12, flop / ms / 1e12),
                  flush=True)
            time.sleep(0.5)
    stop[0] = True; th.join()


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
