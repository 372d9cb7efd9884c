#!/usr/bin/env python3
"""Static check of hand-counted vector-memory waits in a hipcc-generated .s file.

The split GEMM and the layer-tail kernel load row operands with inline asm (`global_load_dwordx4 vN, ...`), so the
compiler does not know those registers are pending until a hand-placed `s_waitcnt vmcnt(N)`.  It may then legally copy,
spill or reuse such a register before the wait -- silently reading stale bytes.  This walks every kernel of the file in
program order with the in-order model of the vector-memory queue (loads, LDS-DMA and stores retire oldest first;
`vmcnt(N)` leaves the N youngest outstanding) and reports every instruction that touches a VGPR whose load is still
outstanding.  Second check (check_scalar_base_hazard): the wait states hipcc does not insert in front of an asm memory
instruction whose scalar base was just written by a VALU instruction.  Third and fourth: the data registers of a wide asm
store, and asm vector instructions that read a matrix-instruction result (check_store_data_hazard, check_mfma_asm_read_hazard).

    python tools/asm_inflight_check.py file.s [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
ACC = 1000  # accumulation registers a0 .. a255 are tracked as 1000 .. 1255 (loads may target them on gfx90a and later)


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(2)) + (ACC if m.group(1) == "a" else 0))
        else:
            base = ACC if m.group(3) == "a" else 0
            out.update(range(base + int(m.group(4)), base + int(m.group(5)) + 1))
    return out


def check_kernel(name, body, max_states_per_block=6):
    """body: list of asm lines of one kernel.  Walks the control-flow graph (every basic block with every distinct queue
    state that reaches it, capped per block) and returns a list of (line_no, text, registers) violations."""
    # ---- basic blocks
    label_at = {}
    for n, raw in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            label_at[m.group(1)] = n
    leaders = sorted(set([0] + list(label_at.values())))
    branch_re = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
    for n, raw in enumerate(body):
        if branch_re.match(raw) or re.match(r"^\s*s_endpgm", raw):
            if n + 1 < len(body):
                leaders.append(n + 1)
    leaders = sorted(set(leaders))
    block_of = {}
    for k, st in enumerate(leaders):
        en = leaders[k + 1] if k + 1 < len(leaders) else len(body)
        block_of[st] = en

    def run_block(st, queue, bad):
        outstanding = [set(x) for x in queue]
        en = block_of[st]
        succ = [en] if en < len(body) else []
        for n in range(st, en):
            line = body[n].split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":"):
                continue
            op = line.split()[0]
            args = line[len(op):]
            mb = branch_re.match(body[n])
            if mb:
                tgt = label_at[mb.group(2)]
                succ = [tgt] if mb.group(1) == "s_branch" else [tgt] + ([en] if en < len(body) else [])
                continue
            if op == "s_endpgm":
                succ = []
                continue
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", line)
                if m:
                    keep = int(m.group(1))
                    outstanding = outstanding[len(outstanding) - keep:] if keep else []
                continue
            pending = set().union(*outstanding) if outstanding else set()
            is_vmem = op.startswith(("global_", "buffer_", "scratch_", "flat_"))
            touched = regs_of(args)
            if is_vmem and op.startswith("global_load") and "_lds_" not in op:
                dest = regs_of(args.split(",")[0])
                hit = ((touched - dest) & pending) | (dest & pending)
                if hit:
                    bad[n] = (n, line, sorted(hit))
                outstanding.append(dest)
            else:
                if touched & pending:
                    bad[n] = (n, line, sorted(touched & pending))
                if is_vmem:
                    outstanding.append(set())  # LDS-DMA, stores, scratch traffic: a queue slot, no VGPR result we track
            if len(outstanding) > 64:
                outstanding = outstanding[-64:]
        return succ, tuple(frozenset(x) for x in outstanding)

    bad = {}
    seen = {}
    work = [(0, ())]
    while work:
        st, queue = work.pop()
        states = seen.setdefault(st, set())
        if queue in states or len(states) >= max_states_per_block:
            continue
        states.add(queue)
        succ, out = run_block(st, queue, bad)
        for nx in succ:
            work.append((nx, out))
    return [bad[k] for k in sorted(bad)]


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_scalar_base_hazard(name, body, need=5):
    """gfx9 / CDNA hazard: a VALU instruction that writes an SGPR (v_readlane_b32, v_readfirstlane_b32 -- how hipcc restores
    a spilled scalar -- or a VALU compare into an SGPR pair) must be `need` wait states away from a vector-memory instruction
    that reads that SGPR as its address.  hipcc pads its own memory instructions but does not look inside asm statements,
    which is where this project's register loads and stores with a scalar base live.  Straight-line scan backwards from every
    global_* instruction with an s[..] address (a label in between only shortens the distance seen: conservative)."""
    ins = []
    for raw in body:
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    bad = []
    for i, t in enumerate(ins):
        op = t.split()[0]
        if not op.startswith("global_") or " s[" not in t:
            continue
        addr = sregs_of(t.split(",")[-1])
        ws = 0
        for j in range(i - 1, -1, -1):
            tt = ins[j]
            o = tt.split()[0]
            if o.startswith("v_") and not o.startswith("v_mfma") and sregs_of(tt[len(o):].split(",")[0]) & addr:
                bad.append((i, t, tt, ws))
                break
            ws += int(tt.split()[1]) + 1 if o == "s_nop" else 1
            if ws >= need:
                break
    return bad


def check_store_data_hazard(name, body, need=2):
    """gfx9 / CDNA hazard: a vector-memory store of more than 64 bits reads its data VGPRs for a few cycles after issue; a
    VALU instruction that overwrites one of them within `need` wait states corrupts the stored value.  hipcc pads its own
    stores; an asm store is opaque to it."""
    ins = []
    for raw in body:
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    bad = []
    for i, t in enumerate(ins):
        if not re.match(r"global_store_dwordx[34]\b", t):
            continue
        data = regs_of(t.split(",")[1])
        ws = 0
        for j in range(i + 1, len(ins)):
            tt = ins[j]
            o = tt.split()[0]
            if o.startswith("v_") and regs_of(tt[len(o):].split(",")[0]) & data:
                bad.append((i, t, tt, ws))
                break
            ws += int(tt.split()[1]) + 1 if o == "s_nop" else 1
            if ws >= need:
                break
    return bad


ASM_VALU = re.compile(r"v_fma_mix(lo|hi)_f16\b|v_max_f32(_e32|_e64)? v\d+, 0, v\d+")  # the vector instructions the kernels write as inline asm


def check_mfma_asm_read_hazard(name, body, need=20):
    """A vector instruction that reads a VGPR written by a matrix instruction must be `need` wait states behind it (up to 18 for the
    16-pass ones; no hardware interlock).  hipcc pads the instructions it knows; GCNHazardRecognizer does not look inside inline asm.
    With accumulators in AGPRs a compiler-known v_accvgpr_read sits in between; proj_ring.hip is built with MFMA results in VGPRs
    and its rides read them through asm v_fma_mix.  Every intervening instruction counts as one wait state (a lower bound), in
    program order."""
    ins = []
    for raw in body:
        t = raw.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    last = {}  # VGPR -> (index, text) of the matrix instruction that wrote it last
    bad = []
    for i, t in enumerate(ins):
        o = t.split()[0]
        ops = t[len(o):].split(",")
        if o.startswith("v_mfma") or o.startswith("v_smfmac"):
            for r in regs_of(ops[0]):
                if r < ACC:
                    last[r] = i
            continue
        if ASM_VALU.match(t):
            for r in regs_of(",".join(ops[1:])):
                if r in last:
                    ws = sum((int(ins[k].split()[1]) + 1) if ins[k].startswith("s_nop") else 1 for k in range(last[r] + 1, i))
                    if ws < need:
                        bad.append((i, t, ins[last[r]], ws))
        if o.startswith(("v_", "ds_read", "ds_load", "global_load", "buffer_load")) and ops and ops[0].strip():
            for r in regs_of(ops[0]):
                last.pop(r, None)
    return bad


def kernels(path):
    lines = open(path).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, nm) in enumerate(starts):
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        yield nm, lines[i:end]


def verify_source(src, flags, out_s, want=""):
    """Compile `src` to assembly with `flags` (the flags of the object that will run) and apply all three checks to every kernel
    whose mangled name contains `want`; raises RuntimeError naming the first violation.  The ablation / stamp builds of the
    tuning tools call this on EVERY variant before they produce a library: a -D switch that removes memory operations changes
    what a hand-counted vmcnt(N) leaves in flight (round 2, gpurun_out/r2i: the "no weight DMA" variant of the layer tail was
    launched with its ring waits still counting twelve pieces that no longer existed; the row operands were consumed while
    pending, their registers recycled as per-lane 64-bit addresses, the loads landed in them, and the next access left the
    aperture -- HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION)."""
    import subprocess
    r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", *flags, src, "-o", out_s],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr))
    n = 0
    for nm, body in kernels(out_s):
        if want not in nm:
            continue
        n += 1
        bad, hz, sd = check_kernel(nm, body), check_scalar_base_hazard(nm, body), check_store_data_hazard(nm, body)
        mh = check_mfma_asm_read_hazard(nm, body)
        if mh:
            raise RuntimeError("%s %s: %s reads a matrix-instruction result with an asm vector instruction %d wait states behind it (%s ; %s)" % (src, flags, nm, mh[0][3], mh[0][2], mh[0][1]))
        if bad:
            raise RuntimeError("%s %s: %s touches a register whose asm load is still in flight (%d places, first: %s)" % (src, flags, nm, len(bad), bad[0][1]))
        if hz:
            raise RuntimeError("%s %s: %s uses a scalar base %d wait states after a VALU write of it (%s -> %s)" % (src, flags, nm, hz[0][3], hz[0][2], hz[0][1]))
        if sd:
            raise RuntimeError("%s %s: %s overwrites the data of a wide store %d wait states behind it (%s ; %s)" % (src, flags, nm, sd[0][3], sd[0][1], sd[0][2]))
    return n


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for nm, body in kernels(path):
        if want not in nm:
            continue
        bad = check_kernel(nm, body)
        n_asm = sum(1 for l in body if re.match(r"\s*global_load_dwordx4 [va]", l))
        print("%s: %d register loads, %d violations" % (nm[:70], n_asm, len(bad)))
        for n, line, regs in bad[:12]:
            print("   +%d  %s   <- pending v%s" % (n, line, regs))
        total += len(bad)
        mh = check_mfma_asm_read_hazard(nm, body)
        if mh:
            print("   %d asm reads of a matrix-instruction result fewer than 20 wait states behind it, first: %s ; %s (%d)" % (len(mh), mh[0][2], mh[0][1], mh[0][3]))
            total += len(mh)
        hz = check_scalar_base_hazard(nm, body)
        if hz:
            print("   %d scalar-base hazards (VALU writes an SGPR < 5 wait states before a memory instruction uses it):" % len(hz))
            for i, t, tt, ws in hz[:6]:
                print("      %s  ->  %s   (%d wait states)" % (tt, t, ws))
        total += len(hz)
        sd = check_store_data_hazard(nm, body)
        if sd:
            print("   %d store-data hazards (VALU overwrites a wide store's data register < 2 wait states behind it):" % len(sd))
            for i, t, tt, ws in sd[:6]:
                print("      %s  ;  %s   (%d wait states)" % (t, tt, ws))
        total += len(sd)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
