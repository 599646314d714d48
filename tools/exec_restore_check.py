#!/usr/bin/env python3
"""Scan the gfx950 ISA of built objects for vector instructions that execute BEFORE the EXEC restore of a join block.

Found in round 2: under register pressure hipcc 7.2 splits the live range of a VGPR array around a
divergent `if` and places the VGPR -> AGPR copies of the split (v_accvgpr_write_b32 aN, vM) at the top of the join block,
AHEAD of the `s_or_b64 exec, exec, s[..]` that ends the `if`.  The copies then run under the `if`'s partial EXEC mask: lanes
that skipped the `if` keep a stale AGPR and read it back later under full EXEC.  psvowr_bwd_kernel<3,1,64,4,256> (two hidden
layers) lost part of its sigma_g sum that way.  This tool makes the pattern visible for every kernel of the library:

    for every `s_and_saveexec_b64 sX, ..` + `s_cbranch_execz T` -- and, since round 3, every `s_xor_b64 exec, exec, sX` +
    `s_cbranch_execz T`, the `else` arm of a diamond (hipcc turns `acc += cond ? v : 0` into one: zero in the flow block, v in
    the else body; bsim_bwd_kernel<4,2,64,4,16,1> lost d = 3 of two scale sums that way) -- it lists the vector instructions
    between the skip target T and the `s_or_b64 exec, exec, sX` that ends that region.  Normally T IS the restore.  Register copies found there
    (v_accvgpr_write / v_accvgpr_read / v_mov / scratch stores and loads: what a live-range split or a spill inserts) are
    class A -- the failure above -- and make the exit code 1; other vector instructions there are the tail of the `if` body
    (address arithmetic of a divergent loop, for example: they run for the active lanes only either way) and are listed as
    class B for information.  Scalar instructions, waits, nops and v_readlane / v_writelane (which ignore EXEC) are skipped.

usage: tools/exec_restore_check.py psvo_amd/csrc/libpsvo_hip.so      (or any number of .o files)
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
BR = re.compile(r"^(s_branch|s_cbranch_\w+)$")


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(obj, tmp):
    """gfx950 code objects of an object file or of a linked library (whose .hip_fatbin holds one bundle per source file)"""
    fat = os.path.join(tmp, "fat")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    data = open(fat, "rb").read()
    offs, i = [], data.find(MAGIC)
    while i >= 0:
        offs.append(i)
        i = data.find(MAGIC, i + 1)
    for n, o in enumerate(offs):
        part = os.path.join(tmp, "bundle%d" % n)
        co = os.path.join(tmp, "co%d" % n)
        with open(part, "wb") as f:
            f.write(data[o:offs[n + 1] if n + 1 < len(offs) else len(data)])
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + co, "--unbundle"], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            yield co


def kernels(co):
    out = subprocess.run([LLVM + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    name, body = None, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        m = INS.match(line)
        if m and name:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if name:
        yield name, body


def harmless(op):
    # scalar work, and the lane accesses of SGPR spills (v_writelane / v_readlane ignore EXEC)
    return op.startswith("s_") or op in ("v_nop", "v_writelane_b32", "v_readlane_b32")


def check(body):
    """[(address of the restore, [(addr, op, args) ...])]: for every `s_and_saveexec_b64 sX, ..` + `s_cbranch_execz T`, the
    vector instructions between the skip target T and the `s_or_b64 exec, exec, sX` that ends that `if` (same block)."""
    index = {a: i for i, (a, _, _) in enumerate(body)}
    bad = []
    for i, (a, op, args) in enumerate(body):
        if op != "s_cbranch_execz" or i + 1 >= len(body):
            continue
        # the instruction that set EXEC for the region this branch skips: `s_and_saveexec_b64 sX, ..` (the `then` arm),
        # or -- the `else` arm of an if / else diamond, found in round 3 -- `s_xor_b64 exec, exec, sX` behind the
        # `s_or_saveexec_b64 sX, sX` of the flow block: the region then ends at `s_or_b64 exec, exec, sX` all the same
        saved = None
        for j in range(i - 1, max(-1, i - 4), -1):
            opj, argj = body[j][1], body[j][2].replace(" ", "")
            if opj in ("s_and_saveexec_b64", "s_or_saveexec_b64"):
                saved = body[j][2].split(",")[0].strip()
                break
            if opj == "s_xor_b64" and argj.startswith("exec,exec,"):
                saved = argj[len("exec,exec,"):]
                break
        if saved is None:
            continue
        try:
            off = int(args.split()[0])
        except (ValueError, IndexError):
            continue
        if off >= 32768:
            off -= 65536
        t = index.get(body[i + 1][0] + 4 * off)
        if t is None or t <= i:
            continue
        found, k = [], t
        while k < len(body) and k < t + 400:
            ak, opk, argk = body[k]
            if opk == "s_or_b64" and argk.replace(" ", "") == "exec,exec," + saved.replace(" ", ""):
                if found:
                    bad.append((ak, found))
                break
            if BR.match(opk) or opk in ("s_endpgm", "s_and_saveexec_b64", "s_or_saveexec_b64"):
                break            # (another structure: not the plain `if` this tool looks at)
            if not harmless(opk):
                found.append((ak, opk, argk))
            k += 1
    # the same `if` without a skip branch (short bodies): s_and_saveexec_b64 sX .. body .. s_or_b64 exec, exec, sX in one
    # block -- everything in between runs under the partial mask, which is right for the body and wrong for a copy of a value
    # that lives THROUGH the `if`; liveness is not known here, so every AGPR / scratch copy in such a body is reported
    for i, (a, op, args) in enumerate(body):
        if op != "s_and_saveexec_b64":
            continue
        saved = args.split(",")[0].strip().replace(" ", "")
        if i + 1 < len(body) and body[i + 1][1] == "s_cbranch_execz":
            continue
        found, k = [], i + 1
        while k < len(body) and k < i + 400:
            ak, opk, argk = body[k]
            if opk == "s_or_b64" and argk.replace(" ", "") == "exec,exec," + saved:
                if found:
                    bad.append((ak, found))
                break
            if BR.match(opk) or opk in ("s_endpgm", "s_and_saveexec_b64", "s_or_saveexec_b64"):
                break
            if is_copy(opk) and opk not in ("v_mov_b32_e32", "v_mov_b64_e32"):
                found.append((ak, opk, argk))
            k += 1
    return bad


# what a live-range split or a spill WRITES (reads of an AGPR / of scratch inside a divergent region are harmless)
COPIES = ("v_accvgpr_write_b32", "v_mov_b32_e32", "v_mov_b64_e32", "v_accvgpr_mov_b32")


def is_copy(op):
    return op in COPIES or op.startswith("scratch_store")


def scan(objs, verbose=True):
    """-> (kernels scanned, [(file, kernel, n class A, n class B)])"""
    n_kernels, rows = 0, []
    for obj in objs:
        with tempfile.TemporaryDirectory() as tmp:
            try:
                cos = list(code_objects(obj, tmp))
            except subprocess.CalledProcessError:
                continue          # (no device code in this file)
            for co in cos:
                for name, body in kernels(co):
                    n_kernels += 1
                    bad = check(body)
                    if not bad:
                        continue
                    ops = [op for _, f in bad for _, op, _ in f]
                    na = sum(1 for op in ops if is_copy(op))
                    rows.append((os.path.basename(obj), name, na, len(ops) - na))
                    if verbose:
                        print("%s %s: %s -- %d copies (class A), %d other vector instructions (class B) ahead of an EXEC "
                              "restore: %s" % ("A" if na else "B", os.path.basename(obj), name, na, len(ops) - na,
                                               ", ".join(sorted(set(ops))[:8])))
    return n_kernels, rows


def main(objs):
    n, rows = scan(objs)
    a = sum(1 for r in rows if r[2])
    print("kernels scanned: %d, class A: %d, class B only: %d" % (n, a, len(rows) - a))
    return 1 if a else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
