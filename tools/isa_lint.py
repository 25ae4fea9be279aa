#!/usr/bin/env python3
"""ISA lint for a hipcc (ROCm 7.2, gfx950) code-generation defect that made k_step_hybrid2<7,3> fault in round 2
(DESIGN.md 3.4 has the evidence: rocgdb registers at the fault, the block in question).

The defect.  In a structurised join block reached by `s_cbranch_execz` the wave arrives with EXEC = 0 and the block is
supposed to begin by restoring EXEC (`s_or_b64 exec, exec, s[..]`).  When SGPR spills (v_writelane: legal there, they
ignore EXEC) sit at the head of such a block, the register allocator's rematerialised constants and split copies for
VGPRs (`v_mov_b32 vN, imm`, ...) can land behind those spills but IN FRONT of the EXEC restore.  They then execute with
the lanes that skipped the `if` switched off -- all of them when the branch was taken -- so those lanes' registers keep
stale contents, and whatever uses them after the join (a select between kernarg offsets, a grid index) reads garbage:
a wild address in one build, deterministic wrong numbers in another.

The rule.  For every target of an `s_cbranch_execz`: between the target and the first instruction that writes EXEC (or
branches) there must be no EXEC-dependent vector instruction (VALU other than v_readlane / v_writelane /
v_readfirstlane, LDS, VMEM).

Usage: isa_lint.py <file> [...]   files: compiler assembly (.s), or objects / shared libraries (.o, .so) whose gfx950
code objects are extracted and disassembled with llvm-objdump (what actually ships).  Exit status 1 on a violation.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LANE_OPS = ("v_readlane", "v_writelane", "v_readfirstlane")
VECTOR = ("v_", "ds_", "global_", "buffer_", "scratch_", "flat_", "image_")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def classify(text):
    """-> 'exec' (writes EXEC or ends the block), 'vector' (EXEC-dependent), or None."""
    sp = text.replace(",", " ").split()
    if not sp:
        return None
    op, dst = sp[0], (sp[1] if len(sp) > 1 else "")
    if op.startswith("s_") and ("saveexec" in op or dst == "exec" or op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc"))):
        return "exec"
    if op.startswith("v_cmpx"):
        return "exec"
    if op.startswith(VECTOR) and not op.startswith(LANE_OPS):
        return "vector"
    return None


def lint_asm(path):
    lines = open(path).read().split("\n")
    targets = set(m.group(1) for m in (re.match(r"\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", l) for l in lines) if m)
    out, kernel = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not (m and m.group(1) in targets):
            continue
        bad = []
        for j in range(i + 1, len(lines)):
            t = lines[j].strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                break                                   # next block reached without an EXEC write (falls through)
            if not t or t.startswith(";") or t.startswith("."):
                continue
            c = classify(t)
            if c == "exec":
                break
            if c == "vector":
                bad.append("line %d: %s" % (j + 1, t))
        if bad:
            out.append((kernel, m.group(1), bad))
    return out


def lint_disassembly(text):
    """llvm-objdump -d --no-show-raw-insn output of one gfx950 code object."""
    funcs = []                                           # (name, [(addr, text, branch_target_offset or None)])
    cur = None
    for l in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):", l)
        if not (m and cur):
            continue
        ins, addr = m.group(1), int(m.group(2), 16)
        tgt = None
        if ins.startswith("s_cbranch_execz"):
            mt = re.search(r"<\S+\+0x([0-9a-f]+)>", l)
            tgt = int(mt.group(1), 16) if mt else None
        cur[1].append((addr, ins, tgt))
    out = []
    for name, ins in funcs:
        if not ins:
            continue
        base = ins[0][0]
        index = {a - base: k for k, (a, _, _) in enumerate(ins)}
        for off in sorted(set(t for _, _, t in ins if t is not None)):
            k = index.get(off)
            if k is None:
                continue
            bad = []
            while k < len(ins):
                c = classify(ins[k][1])
                if c == "exec":
                    break
                if c == "vector":
                    bad.append("+0x%x: %s" % (ins[k][0] - base, ins[k][1]))
                k += 1
            if bad:
                out.append((name, "+0x%x" % off, bad))
    return out


def lint_binary(path):
    tmp = tempfile.mkdtemp(prefix="isa_lint_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.check_call([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out, n = [], 0
        for co in sorted(glob.glob(local + ".*gfx950*")):
            n += 1
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 universal_newlines=True).stdout
            out += lint_disassembly(dis)
        if n == 0:
            raise RuntimeError("%s: no gfx950 code object found" % path)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def lint(path):
    return lint_asm(path) if path.endswith(".s") else lint_binary(path)


def main(argv):
    n_bad = 0
    for p in argv:
        for kernel, where, bad in lint(p):
            n_bad += 1
            print("%s: %s: %s: %d EXEC-dependent instruction(s) before the EXEC restore of an execz join" % (p, kernel, where, len(bad)))
            for b in bad[:8]:
                print("    " + b)
    print("isa_lint: %d violation(s) in %d file(s)" % (n_bad, len(argv)))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
