#!/usr/bin/env python3
"""Second ISA check for wrong code around divergent regions (companion of tools/exec_restore_check.py).

exec_restore_check.py looks at ONE place: between the skip target of a divergent `if` and its EXEC restore.  This tool follows
the EXEC nesting through a whole kernel (linear scan of the structured control flow hipcc emits: s_and_saveexec_b64 / s_or_saveexec_b64
open a region, `s_or_b64 exec, exec, sX` closes the region opened with sX) and reports every register COPY INTO SPILL SPACE made
while EXEC is partial -- v_accvgpr_write_b32 aN, vM and scratch_store_* -- whose destination is read again after the region has
closed (at a shallower nesting depth) with no full-mask rewrite in between.  Lanes that were switched off inside the region did
not take part in the copy; reading the slot outside the region returns their STALE value.  That is the signature of a live
range split / spill placed inside a divergent region for a value that lives through it.

usage: tools/partial_exec_copies.py file.o|lib.so [substring of the (mangled) kernel name]"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exec_restore_check as E
import tempfile

SAVE = ("s_and_saveexec_b64", "s_or_saveexec_b64", "s_andn2_saveexec_b64")


def regs(tok):
    """a3 -> ['a3'];  a[4:7] -> ['a4'..'a7'];  v[2:3] -> [...]"""
    m = re.fullmatch(r"([avs])\[(\d+):(\d+)\]", tok)
    if m:
        return [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return [tok] if re.fullmatch(r"[avs]\d+", tok) else []


def scan_kernel(body):
    """body: [(addr, op, args)].  Returns list of findings (addr of the copy, op, slot, depth at copy, addr of the stale read)."""
    depth, stack = 0, []          # stack of saved-exec sgpr names
    last_write = {}               # slot -> (addr, depth at write, op)
    findings = []
    for a, op, args in body:
        toks = [t.strip() for t in args.split(",")] if args else []
        if op in SAVE and toks:
            stack.append(toks[0].replace(" ", ""))
            depth = len(stack)
            continue
        if op in ("s_or_b64", "s_mov_b64") and len(toks) >= 2 and toks[0] == "exec":
            src = toks[-1].replace(" ", "")
            if src in stack:
                while stack and stack[-1] != src:
                    stack.pop()
                stack.pop()
                depth = len(stack)
            continue
        # writes into spill space
        if op == "v_accvgpr_write_b32" and toks:
            for r in regs(toks[0]):
                last_write[r] = (a, depth, op)
            continue
        if op.startswith("scratch_store") and toks:
            # scratch_store_dword off, v5, off offset:132   -> slot = the address expression
            slot = "scratch:" + ",".join(t for t in toks if not re.fullmatch(r"v\d+|v\[\d+:\d+\]", t))
            last_write[slot] = (a, depth, op)
            continue
        # reads of spill space
        slots = []
        if op == "v_accvgpr_read_b32" and len(toks) >= 2:
            slots = regs(toks[1])
        elif op.startswith("scratch_load") and toks:
            slots = ["scratch:" + ",".join(t for t in toks[1:] if not re.fullmatch(r"v\d+|v\[\d+:\d+\]", t))]
        elif op.startswith("v_mfma") or op.startswith("v_"):
            for t in toks[1:]:
                slots += [r for r in regs(t.split(" ")[0]) if r.startswith("a")]
        for s in slots:
            w = last_write.get(s)
            if w and w[1] > depth:
                findings.append((w[0], w[2], s, w[1], a, depth))
    return findings


def main(argv):
    pat = argv[2] if len(argv) > 2 else ""
    total = bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in E.code_objects(argv[1], tmp):
            for name, body in E.kernels(co):
                if pat not in name:
                    continue
                total += 1
                f = scan_kernel(body)
                if f:
                    bad += 1
                    uniq = sorted(set((x[0], x[1], x[2], x[3]) for x in f))
                    print("%s: %d partial-EXEC spill copies read back outside their region; first: %s" %
                          (name[:110], len(uniq), ["%x %s %s depth %d" % u for u in uniq[:6]]))
    print("kernels scanned: %d, with findings: %d" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
