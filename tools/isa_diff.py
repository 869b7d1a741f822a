#!/usr/bin/env python
"""Is the code the GPU would run still the code that last ran on one? Compiles yolov5m_amd/csrc at a git revision and at the working tree
to gfx950 assembly (hipcc --cuda-device-only -S: no GPU involved) and compares every kernel instantiation's instruction stream (comments,
directives and basic-block numbering removed). Rounds 4-6 had no GPU: the default path is claimed to be round 3's code at the instruction
level (`619fd98` = round 3's HEAD, the last revision whose library passed the GPU suite on hardware) -- this prints that claim kernel by kernel.
usage: python tools/isa_diff.py [rev] > profiles/rNN_isa_diff.txt        (default rev: 619fd98)"""
import difflib
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = {"y5m_detect.hip", "y5m_loss.hip"}


def asm_of(csrc_dir, include_dir, out_dir):
    res = {}
    for path in sorted(glob.glob(os.path.join(csrc_dir, "*.hip"))):
        base = os.path.basename(path)
        asm = os.path.join(out_dir, base + ".s")
        flags = ["-ffp-contract=off"] if base in EXACT else []
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + include_dir, "--cuda-device-only", "-S",
                        "-o", asm, path] + flags, check=True, capture_output=True)
        res[base] = kernels(open(asm).read())
    return res


def kernels(text):
    """symbol -> normalised instruction list of every .amdhsa kernel in an assembly file"""
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M))
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^(\w+):", ln)
        if m and m.group(1) in names:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        s = ln.split(";")[0].strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith(".") and not s.startswith(".LBB"):
            continue
        s = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", s)
        cur.append(re.sub(r"\s+", " ", s))
    return out


def demangle(names):
    try:
        o = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, o)) if len(o) == len(names) else {n: n for n in names}
    except Exception:
        return {n: n for n in names}


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "619fd98"
    with tempfile.TemporaryDirectory() as d:
        old = os.path.join(d, "old")
        files = subprocess.run(["git", "-C", ROOT, "ls-tree", "-r", "--name-only", rev, "yolov5m_amd/csrc", "include"], capture_output=True, text=True, check=True).stdout.split()
        for f in files:                                      # (same relative layout: the sources include "../../include/y5m.h")
            dst = os.path.join(old, f)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, "wb") as fh:
                fh.write(subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{f}"], capture_output=True, check=True).stdout)
        os.makedirs(os.path.join(d, "a")); os.makedirs(os.path.join(d, "b"))
        A = asm_of(os.path.join(old, "yolov5m_amd", "csrc"), os.path.join(old, "include"), os.path.join(d, "a"))
        B = asm_of(os.path.join(ROOT, "yolov5m_amd", "csrc"), os.path.join(ROOT, "include"), os.path.join(d, "b"))
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "yolov5m_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
    print(f"# tools/isa_diff.py: kernels of yolov5m_amd/csrc at {rev} against the working tree (HEAD {head}{' + uncommitted changes' if dirty else ''}), gfx950, hipcc -O3")
    print("# same = identical instruction stream; SAME-LENGTH = same number of instructions and the same opcode multiset (register renaming /")
    print("# operand order); CHANGED = anything else; NEW / GONE = instantiation exists on one side only")
    tot = {"same": 0, "SAME-LENGTH": 0, "CHANGED": 0, "NEW": 0, "GONE": 0}
    gone_same = gone_near = 0
    for f in sorted(set(A) | set(B)):
        ka, kb = A.get(f, {}), B.get(f, {})
        dm = demangle(sorted(set(ka) | set(kb)))
        rows = []
        for k in sorted(set(ka) | set(kb)):
            if k not in ka:
                st = "NEW"
            elif k not in kb:
                st = "GONE"
            elif ka[k] == kb[k]:
                st = "same"
            elif len(ka[k]) == len(kb[k]) and sorted(x.split(" ")[0] for x in ka[k]) == sorted(x.split(" ")[0] for x in kb[k]):
                st = "SAME-LENGTH"
            else:
                st = "CHANGED"
            tot[st] += 1
            rows.append((st, dm[k], len(ka.get(k, [])), len(kb.get(k, []))))
        print(f"\n## {f}: " + ", ".join(f"{sum(1 for r in rows if r[0] == s)} {s}" for s in tot if any(r[0] == s for r in rows)))
        # an instantiation that is GONE because its template grew a parameter (the round-4/5 forms live next to the round-3 ones behind a
        # new trailing argument): find the NEW instantiation of the same kernel with the identical instruction stream
        def base(sym):                                      # _Z<len><name>...: the function's own name
            m = re.match(r"_Z(\d+)", sym)
            return sym[m.end():m.end() + int(m.group(1))] if m else sym
        new_syms = [k for k in kb if k not in ka]
        twin = {}
        for k in ka:
            if k in kb:
                continue
            best = None
            m0 = re.match(r"_Z(\d+)", k)
            cut = k.find("Ev", m0.end() + int(m0.group(1))) if m0 else -1
            prefix = k[:cut] if cut > 0 else k[:m0.end() + int(m0.group(1))]      # the old template arguments, still open: a twin extends them
            for n in new_syms:
                if base(n) != base(k) or not n.startswith(prefix):
                    continue
                if kb[n] == ka[k]:
                    best = (0, n, "same (as)")
                    break
                ndiff = sum(max(i2 - i1, j2 - j1) for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, ka[k], kb[n], autojunk=False).get_opcodes() if tag != "equal")
                ops = lambda v: sorted(x.split(" ")[0] for x in v)
                rel = "SAME-LENGTH" if (len(ka[k]) == len(kb[n]) and ops(ka[k]) == ops(kb[n])) else "CHANGED"
                if best is None or ndiff < best[0]:
                    best = (ndiff, n, rel)
            if best is not None:
                twin[k] = best
        for st, name, la, lb in rows:
            sym = next(k for k in dm if dm[k] == name)
            if st == "GONE" and sym in twin:
                nd, n, rel = twin[sym]
                if nd == 0:
                    gone_same += 1
                else:
                    gone_near += 1
                print(f"  {rel:12s} {la:6d} -> {len(kb[n]):6d}  {name[:100]}  ->  {dm[n][:100]}" + (f"   ({nd} differing instruction(s))" if nd else ""))
            elif st == "NEW" and any(sym == t[1] for t in twin.values()):
                continue
            elif st != "same":
                print(f"  {st:12s} {la:6d} -> {lb:6d}  {name[:150]}")
    print("\ntotal: " + ", ".join(f"{v} {k}" for k, v in tot.items()) + f"; of the GONE ones {gone_same} live on instruction-for-instruction under a longer template signature, {gone_near} with the differences listed")


if __name__ == "__main__":
    main()
