#!/usr/bin/env python3
"""Hazard lint for the inline-assembly vector instructions in the MFMA kernels.

The Winograd transforms use `v_pk_add_f32` with negate modifiers through inline assembly (hipcc scalarises a float2
subtraction).  The compiler's hazard recognizer pads the matrix-pipe hazards of gfx950 with s_nop for the instructions it
knows, but an asm statement is opaque to it: nothing is inserted for what the statement reads or writes.  Whether a hazard
bites then depends on scheduling and register allocation.  Observed (round 2, a software-pipelined variant of the Winograd
consumer): the compiler moved the transform of one register pair in between the MFMAs and the MFMA that consumed the result
followed ONE instruction later -- it read the old register; one transform-domain position of one patch group was wrong in
three kernel variants, everything else right.

This script compiles a .hip source to gfx950 assembly and checks every inline-asm vector instruction against the MFMAs
around it (wait states: one per instruction, N + 1 for `s_nop N`; rule A follows every control-flow path -- fall-through and
the target of each branch met inside the window, loop back-edges included; rules B-D look back in textual order):

  A  asm writes a VGPR that an MFMA reads (SrcA/B/C) fewer than 2 wait states later  (VALU write -> MFMA read; LLVM pads 2 for
     instructions it knows).  VALIDATED: the failing build had three of these, every passing build none.  An ERROR.
  B  asm writes a VGPR that an MFMA issued fewer than (passes - 1) wait states earlier reads
  C  asm reads  a VGPR that is the result of an MFMA issued fewer than (passes + 3) wait states earlier
  D  asm writes a VGPR that is the result of an MFMA issued fewer than (passes + 3) wait states earlier
     B-D use LLVM's distances for the instructions it knows; builds with many B/C/D sites pass every parity test (the
     kernels read MFMA results first through compiler-visible adds, which are padded), so they are reported as NOTES only.

usage: python tools/isa_lint.py [-v] csrc/conv_mfma.hip [...]      exit status 1 if any rule-A site exists
"""
import re
import subprocess
import sys

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-"]

REG = re.compile(r"(?<![\w.])([va])(?:\[(\d+):(\d+)\]|(\d+)(?!\w))")


def regs(tok):
    """register set of one operand token: {('v', n), ...}"""
    out = set()
    for m in REG.finditer(tok):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def passes(op):
    """upper bound of the MFMA's pass count from its shape (4 cycles per pass)"""
    m = re.search(r"_(\d+)x(\d+)x(\d+)", op)
    mm = int(m.group(1)) if m else 16
    return 2 if mm == 4 else 16 if mm == 32 else 8


LABELS = {}   # {function: {label: instruction index}} of the last parse() call


def parse(asm_text):
    """-> {function: [instr]}, instr = dict(op, dst, srcs, asm, ws, line[, target]); fills LABELS"""
    LABELS.clear()
    funcs, cur, in_asm = {}, None, False
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^([A-Za-z_][\w.$]*):", raw)
        if m and not raw.startswith(".L") and not raw.startswith("\t"):
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None and re.match(r"^\.L[\w.$]+:", line) and not line.startswith(".Lfunc_end"):
            LABELS.setdefault(cur, {})[line.split(":")[0]] = len(funcs[cur])       # label -> index of the next instruction
            continue
        if cur is None or line.startswith((".", ";")) or line.endswith(":"):
            if line.startswith(".Lfunc_end"):
                cur = None
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        # modifiers like neg_lo:[0,1] contain commas: re-join tokens until brackets balance
        toks, buf = [], ""
        for t in ops:
            buf = t if not buf else buf + "," + t
            if buf.count("[") == buf.count("]"):
                toks.append(buf)
                buf = ""
        ws = 1
        if op == "s_nop" and toks:
            ws = int(toks[0], 0) + 1
        ins = {"op": op, "asm": in_asm, "ws": ws, "line": ln, "text": code}
        if op.startswith(("s_cbranch", "s_branch")) and toks:
            ins["target"] = toks[0]
        if op.startswith("v_") and toks:
            ins["dst"] = regs(toks[0])
            ins["srcs"] = [regs(t.split(" ")[0]) for t in toks[1:]]
        else:
            ins["dst"], ins["srcs"] = set(), []
        funcs[cur].append(ins)
    return funcs


def lint_function(name, instrs):
    findings = []
    for i, ins in enumerate(instrs):
        if not (ins["asm"] and ins["op"].startswith("v_")):
            continue
        rd = set().union(*ins["srcs"]) if ins["srcs"] else set()
        wr = ins["dst"]
        # forward: rule A, along EVERY control-flow path of the next two wait states: fall-through, and the target of each
        # branch met on the way (conditional: both ways; s_branch: the target only) -- loop back-edges included
        labels = LABELS.get(name, {})
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, gap = stack.pop()
            while j < len(instrs) and gap < 2 and (j, gap) not in seen:
                seen.add((j, gap))
                nxt = instrs[j]
                if nxt["op"].startswith("v_mfma") and wr & set().union(*nxt["srcs"]):
                    findings.append((name, ins["line"], "A", f"{ins['text']}  ->  {nxt['text']} ({gap} wait states)"))
                gap += nxt["ws"]
                if "target" in nxt:
                    if nxt["target"] in labels:
                        stack.append((labels[nxt["target"]], gap))
                    if nxt["op"].startswith("s_branch"):
                        break
                if nxt["op"] in ("s_endpgm", "s_setpc_b64"):
                    break
                j += 1
        # backward: rules B, C, D
        gap = 0
        for prv in reversed(instrs[:i]):
            if gap >= 20:
                break
            if prv["op"].startswith("v_mfma"):
                p = passes(prv["op"])
                srcs = set().union(*prv["srcs"]) if prv["srcs"] else set()
                if gap < p - 1 and wr & srcs:
                    findings.append((name, ins["line"], "B", f"{prv['text']}  ->  {ins['text']} ({gap} wait states)"))
                if gap < p + 3 and rd & prv["dst"]:
                    findings.append((name, ins["line"], "C", f"{prv['text']}  ->  {ins['text']} ({gap} wait states)"))
                if gap < p + 3 and wr & prv["dst"]:
                    findings.append((name, ins["line"], "D", f"{prv['text']}  ->  {ins['text']} ({gap} wait states)"))
            gap += prv["ws"]
    return findings


def lint_text(asm_text):
    out = []
    for name, instrs in parse(asm_text).items():
        out += lint_function(name, instrs)
    return out


def compile_to_asm(src, extra=()):
    r = subprocess.run([HIPCC, *FLAGS, *extra, src], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return r.stdout


def main():
    verbose = "-v" in sys.argv
    bad = 0
    for src in [a for a in sys.argv[1:] if a != "-v"]:
        text = compile_to_asm(src)
        f = lint_text(text)
        n_asm = sum(1 for fn in parse(text).values() for i in fn if i["asm"] and i["op"].startswith("v_"))
        errs = [x for x in f if x[2] == "A"]
        notes = len(f) - len(errs)
        print(f"{src}: {n_asm} inline-asm vector instructions, {len(errs)} rule-A hazards, {notes} notes (B-D)")
        for name, line, rule, what in (f if verbose else errs)[:60]:
            print(f"  [{rule}] {name[:60]} asm line {line}: {what}")
        bad += len(errs)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
