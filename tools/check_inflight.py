"""Static check of the column kernel's ISA: no instruction may read or write a VGPR that is the destination of a
ds_read issued from an inline-asm statement and not yet covered by an s_waitcnt.

The serial sweeps of k_column_ps issue their LDS reads through asm statements and say themselves when a value is
needed (ps_lds_read2 / ps_lds_wait in mckpp_kernels_ps.hip).  Between the two the value is an ordinary C++ variable
for the compiler, which may copy it (phi copies, live-range splits) - and a copy made while the read is in flight
copies whatever the register held before: the hardware does not interlock LDS returns.  This script walks the
generated assembly, keeps for every asm-issued read the number of LDS operations issued since (they retire in
order; an s_waitcnt lgkmcnt(N) leaves the youngest N), and reports every instruction outside an asm statement that touches
a register with a read still outstanding.  It is a forward data-flow analysis over the basic blocks of every
function: where paths merge, a read counts as outstanding if it is on any of them.

    python tools/check_inflight.py            (compiles mckpp_kernels_ps.hip with the library's flags)
    python tools/check_inflight.py file.s
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mckpp_f90_amd", "csrc")


def isa_text():
    out = subprocess.run(["make", "-s", "-C", CSRC, "isa"], capture_output=True, text=True, timeout=1800)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    return out.stdout


def regs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _kernels(text):
    """-> [(name, [(line number, text, in_asm)])] for every function of the assembly"""
    out, cur, inasm = [], None, False
    for i, line in enumerate(text.split("\n"), 1):
        s = line.strip()
        if line.startswith("_Z") and ":" in line and not line.startswith("\t"):
            cur = (line.split(":")[0], [])
            out.append(cur)
            continue
        if cur is None:
            continue
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if s[0] == "." and not s.endswith(":"):
            continue
        cur[1].append((i, s, inasm))
    return out


def _blocks(instrs):
    """basic blocks: list of dicts {label, ins: [(line, text, inasm)], succ: [block indices]}"""
    blocks, cur = [], {"label": None, "ins": []}
    for i, s, inasm in instrs:
        if s.endswith(":"):
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": s[:-1], "ins": []}
            continue
        cur["ins"].append((i, s, inasm))
        op = s.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    index = {b["label"]: n for n, b in enumerate(blocks) if b["label"]}
    for n, b in enumerate(blocks):
        succ = []
        last = b["ins"][-1][1].split() if b["ins"] else [""]
        op = last[0]
        if op == "s_branch":
            succ = [index[last[1]]] if last[1] in index else []
        elif op.startswith("s_cbranch"):
            if last[1] in index:
                succ.append(index[last[1]])
            if n + 1 < len(blocks):
                succ.append(n + 1)
        elif op in ("s_endpgm", "s_setpc_b64"):
            succ = []
        elif n + 1 < len(blocks):
            succ = [n + 1]
        b["succ"] = succ
    return blocks


def _transfer(state, block, report=None):
    """state: {frozenset(dest VGPRs): (LDS operations issued since, line of the read)} -> state at the block's end"""
    st = dict(state)
    for i, s, inasm in block["ins"]:
        ops = re.split(r"[ ,\t]+", s)
        op = ops[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                st = {d: v for d, v in st.items() if v[0] < n}   # the youngest n operations may be outstanding
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime"):
            st = {d: (v[0] + 1, v[1]) for d, v in st.items()}
            if inasm and op.startswith("ds_read"):
                st[frozenset(regs(ops[1]))] = (0, i)
            continue
        if inasm:
            continue
        if report is not None:
            used = set()
            for t in ops[1:]:
                used |= regs(t)
            for d, v in st.items():
                if used & d:
                    report.append((i, s, v[1]))
                    break
    return st


def check(text):
    """-> list of (kernel, line number, instruction, line of the read): forward data-flow over the basic blocks, the
    states of merging paths joined register by register (the fewest operations issued since the read)."""
    found = []
    for name, instrs in _kernels(text):
        blocks = _blocks(instrs)
        if not blocks:
            continue
        entry = [None] * len(blocks)
        entry[0] = {}
        work = [0]
        while work:
            n = work.pop()
            out = _transfer(entry[n], blocks[n])
            for m in blocks[n]["succ"]:
                if entry[m] is None:
                    entry[m] = dict(out)
                    work.append(m)
                else:
                    changed = False
                    for d, v in out.items():
                        if d not in entry[m] or v[0] < entry[m][d][0]:
                            entry[m][d] = v
                            changed = True
                    if changed:
                        work.append(m)
        seen = set()
        for n, b in enumerate(blocks):
            if entry[n] is None:
                continue
            rep = []
            _transfer(entry[n], b, rep)
            for i, s, ln in rep:
                if i not in seen:
                    seen.add(i)
                    found.append((name, i, s, ln))
    found.sort(key=lambda t: t[1])
    return found


if __name__ == "__main__":
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else isa_text()
    bad = check(text)
    for k, i, s, ln in bad[:40]:
        print(f"{(k or '')[:48]} line {i}: {s}   <- ds_read of line {ln} still outstanding")
    print(f"{len(bad)} instruction(s) touch a register with an asm-issued LDS read in flight")
    sys.exit(1 if bad else 0)
