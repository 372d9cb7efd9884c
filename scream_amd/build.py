"""Build libscream_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library
is a plain C-ABI shared object (include/scream_hip.h) loaded with ctypes."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscream_hip.so")
ARCH = "gfx950"

# parity-critical files keep every fp32 operation individually rounded (see the file headers)
SOURCES = {
    "gemm_f32.hip": [],
    "gemm_split.hip": [],
    # the MFMA results of this kernel are consumed by vector instructions (every chunk's epilogue rides under the next chunk's
    # matrix instructions) while its row operand planes fill the whole AGPR file: accumulators in VGPRs, no moves between the files
    "proj_ring.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # several code instances of the same arithmetic (an attention apply in the open / riding under another stage) must round
    # identically, whatever hipcc makes of each: no fma contraction in this file
    "tail_split.hip": ["-ffp-contract=off"],
    "embed.hip": [],
    "attention.hip": [],
    "forward.hip": [],
    "nn_search.hip": ["-ffp-contract=off"],
    "kabsch.hip": ["-ffp-contract=off"],
    "icp_grid.hip": ["-ffp-contract=off"],
}
# SCREAM_HIPCC_EXTRA="file.hip:-flag,-flag;file2.hip:-flag": extra compiler flags per source (experiments; empty in the product)
EXTRA_DEFINES = {}
for _item in filter(None, os.environ.get("SCREAM_HIPCC_EXTRA", "").split(";")):
    EXTRA_DEFINES[_item.split(":")[0]] = _item.split(":", 1)[1].split(",")
ASM_LOADS = ("gemm_split.hip", "tail_split.hip", "proj_ring.hip")  # verified after code generation, see verify_one
# kernels that must not touch scratch at all: a spill inside a ring stage costs a round trip per stage, and hipcc orders a scratch
# reload against the LDS-DMA in flight with vmcnt(0) -- the ring would drain once per stage (the layer tail's history)
NO_SCRATCH = {"proj_ring.hip": ("proj_ring_kernel",), "tail_split.hip": ("tail_kernelINS_7SplitH2", "tail_kernelINS_7SplitH1")}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libscream_hip.so")
    return exe


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "scream_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str = None) -> str:
    """out: write the library there instead of scream_amd/libscream_hip.so (experiment builds loaded with SCREAM_LIB=)."""
    global LIB
    if out:
        LIB, force = out, True
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-Wall", "-Wno-unused-function"]

    def compile_one(item):
        src, extra = item
        extra = extra + EXTRA_DEFINES.get(src, [])
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = common + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    def verify_one(src):
        """The files in ASM_LOADS request operands with inline asm and wait for them with hand-counted s_waitcnt vmcnt(N);
        whether hipcc's register allocation respects that is a property of the generated code, not of the source.  So the
        build proves it on the very code it ships (same flags, assembly instead of an object): tools/asm_inflight_check.py
        walks every kernel and the build FAILS if any instruction touches a register whose load is still outstanding."""
        sys.path.insert(0, os.path.join(HERE, "..", "tools"))
        import asm_inflight_check as chk
        asm = os.path.join(objdir, src.replace(".hip", ".s"))
        cmd = common + SOURCES[src] + EXTRA_DEFINES.get(src, []) + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", asm]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr))
        for name, body in chk.kernels(asm):
            if any(k in name for k in NO_SCRATCH.get(src, ())) and any(("scratch_" in l or "buffer_store" in l) for l in body):
                raise RuntimeError("%s: %s spills registers to scratch (this hipcc allocates registers differently from the one the "
                                   "kernel was written with)" % (src, name))
            bad = chk.check_kernel(name, body)
            if bad:
                raise RuntimeError("%s: %s touches a register whose asm load is still in flight (%d places, first: %s); "
                                   "this hipcc allocates registers differently from the one the kernels were written with"
                                   % (src, name, len(bad), bad[0][1]))
            sd = chk.check_store_data_hazard(name, body)
            if sd:
                raise RuntimeError("%s: %s overwrites the data of a wide store %d wait states behind it (%s ; %s)"
                                   % (src, name, sd[0][3], sd[0][1], sd[0][2]))
            mh = chk.check_mfma_asm_read_hazard(name, body)
            if mh:
                raise RuntimeError("%s: %s reads a matrix-instruction result with an asm vector instruction %d wait states behind it "
                                   "(%s ; %s)" % (src, name, mh[0][3], mh[0][2], mh[0][1]))
            hz = chk.check_scalar_base_hazard(name, body)
            if hz:
                raise RuntimeError("%s: %s uses a scalar base %d wait states after a VALU write of it (%s -> %s)"
                                   % (src, name, hz[0][3], hz[0][2], hz[0][1]))
        return src

    with ThreadPoolExecutor(max_workers=4) as ex:
        checks = [ex.submit(verify_one, src) for src in ASM_LOADS] if os.environ.get("SCREAM_BUILD_VERIFY", "1") != "0" else []
        objs = list(ex.map(compile_one, SOURCES.items()))
        for c in checks:
            c.result()
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
