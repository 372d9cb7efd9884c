#!/usr/bin/env python3
"""Static check of hand-counted vector-memory waits in a hipcc-generated .s file.

The split GEMM and the layer-tail kernel load row operands with inline asm (`global_load_dwordx4 vN, ...`), so the
compiler does not know those registers are pending until a hand-placed `s_waitcnt vmcnt(N)`.  It may then legally copy,
spill or reuse such a register before the wait -- silently reading stale bytes.  This walks every kernel of the file in
program order with the in-order model of the vector-memory queue (loads, LDS-DMA and stores retire oldest first;
`vmcnt(N)` leaves the N youngest outstanding) and reports every instruction that touches a VGPR whose load is still
outstanding.

    python tools/asm_inflight_check.py file.s [kernel-name-substring]
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
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


def kernels(path):
    lines = open(path).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, nm) in enumerate(starts):
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        yield nm, lines[i:end]


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for nm, body in kernels(path):
        if want not in nm:
            continue
        bad = check_kernel(nm, body)
        n_asm = sum(1 for l in body if re.match(r"\s*global_load_dwordx4 v", l))
        print("%s: %d register loads, %d violations" % (nm[:70], n_asm, len(bad)))
        for n, line, regs in bad[:12]:
            print("   +%d  %s   <- pending v%s" % (n, line, regs))
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
