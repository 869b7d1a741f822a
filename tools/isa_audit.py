#!/usr/bin/env python
"""Static audit of the gfx950 code hipcc generates for every kernel of yolov5m_amd/csrc (no GPU needed: hipcc cross-compiles).
Per kernel instantiation: VGPRs / AGPRs / SGPRs, spilled registers, scratch bytes, static LDS bytes, waves per SIMD the
register count allows (512 VGPRs per SIMD lane, allocation granule 8), and the instruction mix of its innermost loops (MFMA,
LDS, global / buffer memory, VALU, SALU, waits, barriers) together with two patterns that cost time on CDNA4 and are easy to
write by accident: IEEE divisions (v_div_fixup_f32), a load immediately followed by `s_waitcnt vmcnt(0)` inside a loop (a
dependent memory round trip per iteration: "dep"), and loops with several FULL waits for their loads (branches around loads make
the compiler wait for everything in flight at every use: "FULL-WAITS"). usage: tools/isa_audit.py [file.hip ...] > profiles/rNN_isa_audit.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolov5m_amd", "csrc")
EXACT = {"y5m_detect.hip", "y5m_loss.hip"}


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names) + "\n", capture_output=True, text=True).stdout
        out = out.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def audit(path):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        flags = ["-ffp-contract=off"] if os.path.basename(path) in EXACT else []
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", "-o", asm, path] + flags, check=True, capture_output=True)
        text = open(asm).read()
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]
        name = g("name")
        meta[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                          vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                          lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")))
    # function bodies
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\.Lfunc_end\d+:", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    rows = []
    names = sorted(meta)
    pretty = dict(zip(names, demangle(names)))
    for n in names:
        md, body = meta[n], bodies.get(n, "")
        lines = [l.strip() for l in body.splitlines()]
        ops = [l.split()[0] for l in lines if l and not l.startswith((";", ".", "_")) and not l.endswith(":")]
        tot = {}
        for o in ops:
            tot[classify(o)] = tot.get(classify(o), 0) + 1
        # loops: a header label ("This Inner Loop Header" / "This Loop Header") owns every following block LLVM annotates with
        # its name ("in Loop: Header=BBx_y" or "Parent Loop BBx_y"); nested loops are counted inside their parents as well
        blocks, cur_lbl, cur_note = [], None, ""
        for l in body.splitlines():
            st = l.strip()
            m_ = re.match(r"^(\.LBB\d+_\d+):(.*)$", st)
            if m_:
                cur_lbl, cur_note = m_.group(1), m_.group(2)
                blocks.append([cur_lbl, cur_note, []])
                continue
            if blocks and st and not st.startswith((";", ".")):
                blocks[-1][2].append(st)
            elif blocks and st.startswith(";") and ("Loop" in st):
                blocks[-1][1] += " " + st                       # (continuation lines of the block comment)
        loops = []
        for i, (lbl, note, _) in enumerate(blocks):
            if "Loop Header" not in note:
                continue
            name_ = lbl[2:]                                      # "BBx_y"
            ll = list(blocks[i][2])
            for lbl2, note2, ins2 in blocks[i + 1:]:
                if ("Header=" + name_) in note2 or ("Parent Loop " + name_) in note2:
                    ll += ins2
                elif "Loop" not in note2:
                    break
            loops.append(ll)
        def summarise(ll):
            mix = {}
            for a in ll:
                k = classify(a.split()[0])
                mix[k] = mix.get(k, 0) + 1
            dep = sum(1 for a, b in zip(ll, ll[1:]) if a.startswith(("buffer_load", "global_load")) and b.startswith("s_waitcnt vmcnt(0)"))
            full = sum(1 for a in ll if a.startswith("s_waitcnt vmcnt(0)"))
            loads = sum(1 for a in ll if a.startswith(("buffer_load", "global_load")))
            return (len(ll), mix, dep, full, loads)
        loops = [summarise(ll) for ll in loops]
        loops.sort(key=lambda t: -t[0])
        steady = steady_paths(body)
        granule = (md["vgpr"] + 7) // 8 * 8              # (.vgpr_count is the unified total on gfx90a+: arch VGPRs + AGPRs)
        waves = min(8, 512 // max(granule, 8))
        rows.append((pretty[n], md, waves, len(ops), tot, loops[:2], sum(1 for o in ops if o.startswith("v_div_fixup")), steady[:2]))
    return rows


def steady_paths(body):
    """The STEADY-STATE iteration of every loop that holds MFMAs: the static loop mix counts both sides of every branch (a
    persistent kernel's epilogue sits inside its unit loop and is taken once per tile), so for each loop header this finds the
    cheapest cycle header -> ... -> header, in instructions, that passes through every basic block of the loop that issues
    MFMAs or fetches / stages operands (buffer_load_dwordx4, ds_write_b128), and returns its instruction mix. Basic blocks are split at labels AND behind every branch; edges are the branch
    targets and the fall-through. Returns [(instructions, mix, full vmcnt(0) waits)] sorted by MFMA count."""
    import heapq
    blocks = []                    # [label or None, loop note, instructions]
    note = ""
    for l in body.splitlines():
        st = l.strip()
        m_ = re.match(r"^(\.LBB\d+_\d+):(.*)$", st)
        if m_:
            note = m_.group(2)
            blocks.append([m_.group(1), note, []])
            continue
        if not blocks:
            blocks.append([None, "", []])
        if st.startswith(";"):
            if "Loop" in st:
                blocks[-1][1] += " " + st
                note = blocks[-1][1]
            continue
        if not st or st.startswith("."):
            continue
        blocks[-1][2].append(st)
        if st.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            blocks.append([None, note, []])            # the instructions behind a branch are a new basic block of the same loop
    idx = {b[0]: i for i, b in enumerate(blocks) if b[0]}
    succ = []
    for i, (lbl, _n, ins) in enumerate(blocks):
        out = []
        last = ins[-1] if ins else ""
        if last.startswith(("s_cbranch", "s_branch")):
            t = last.split()[-1]
            if t in idx:
                out.append(idx[t])
        if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    res = []
    for h, (lbl, n_, _ins) in enumerate(blocks):
        if not lbl or "Loop Header" not in n_:
            continue
        name_ = lbl[2:]
        # (a loop's latch may be laid out IN FRONT of its header: every block annotated with the header's name belongs to it)
        members = {h} | {j for j, b in enumerate(blocks) if re.search(r"(Header=|Parent Loop )" + name_ + r"\b", b[1])}
        need = need_mfma = [j for j in sorted(members) if any(a.startswith("v_mfma") for a in blocks[j][2])]
        if not need:
            continue
        # ... and through every block that STAGES operands (16-byte LDS stores): the prefetch of the next tile sits behind an
        # `if (more)` that only the last iteration skips, and the cheapest cycle would skip it too
        # (same for the 16-byte buffer loads that fetch them; the kernels' epilogues use global_load / global_store)
        need = sorted(set(need) | {j for j in members if any(a.startswith(("ds_write_b128", "buffer_load_dwordx4")) for a in blocks[j][2])})

        def shortest(src, dst):
            dist, heap = {src: 0}, [(0, src, [src])]
            while heap:
                d, u, path = heapq.heappop(heap)
                if u == dst and len(path) > 1:
                    return d, path
                if d > dist.get(u, 1 << 30):
                    continue
                for v in succ[u]:
                    if v not in members:
                        continue
                    nd = d + len(blocks[v][2])
                    if nd < dist.get(v, 1 << 30) or v == dst:
                        if v != dst:
                            dist[v] = nd
                        heapq.heappush(heap, (nd, v, path + [v]))
            return None, None
        def cycle(req):
            stops = [h] + [j for j in req if j != h] + [h]
            path = [h]
            for a, b in zip(stops, stops[1:]):
                d, p_ = shortest(a, b)
                if p_ is None:
                    return None
                path += p_[1:]
            return path
        path = cycle(need) or cycle(need_mfma)          # (staging blocks that no single cycle visits in layout order: MFMA blocks only)
        if path is None:
            continue
        ins = [a for j in path[:-1] for a in blocks[j][2]]
        mix = {}
        for a in ins:
            k = classify(a.split()[0])
            mix[k] = mix.get(k, 0) + 1
        res.append((len(ins), mix, sum(1 for a in ins if a.startswith("s_waitcnt vmcnt(0)"))))
    res.sort(key=lambda t: -t[1].get("mfma", 0))
    return res


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print("# tools/isa_audit.py -- hipcc --offload-arch=gfx950 -O3 of yolov5m_amd/csrc at HEAD (static: no GPU involved)")
    print("# per kernel: VGPRs (of which AGPRs) / SGPRs, spills (v/s), scratch B, static LDS B, waves per SIMD by registers, instructions;")
    print("# then its one or two largest loops: instruction count and mix (STATIC: both sides of every branch), `dep` = loads directly followed by")
    print("# s_waitcnt vmcnt(0), FULL-WAITS = s_waitcnt vmcnt(0) count when the loop holds >= 4 loads and >= 2 of them;")
    print("# `steady-state iteration` (MFMA loops): the cheapest cycle header -> header through every MFMA-issuing block of the loop = what one")
    print("# iteration executes when no once-per-tile branch (epilogue, tile change) is taken")
    for f in files:
        print(f"\n## {os.path.basename(f)}")
        for name, md, waves, nops, tot, loops, ndiv, steady in audit(f):
            short = re.sub(r"\(.*", "", name)[:86]
            print(f"{short:86s} v{md['vgpr']:3d}(a{md['agpr']:<3d}) s{md['sgpr']:3d} spill {md['vspill']}/{md['sspill']} scratch {md['scratch']:4d} "
                  f"lds {md['lds']:6d} waves/SIMD {waves} instr {nops:5d}" + (f" IEEE-div {ndiv}" if ndiv else ""))
            for n_, mix, dep, full, loads in loops:
                if n_ >= 24:
                    print("    loop %4d: " % n_ + " ".join(f"{k} {mix[k]}" for k in ("mfma", "lds", "vmem", "valu", "salu", "wait", "barrier") if k in mix)
                          + (f"  dep {dep}" if dep else "") + (f"  FULL-WAITS {full} for {loads} loads" if (full >= 2 and loads >= 4) else ""))
            for n_, mix, full in steady:
                print("    steady-state iteration %4d: " % n_ + " ".join(f"{k} {mix[k]}" for k in ("mfma", "lds", "vmem", "valu", "salu", "wait", "barrier") if k in mix)
                      + f"  ({mix.get('valu', 0) / max(mix.get('mfma', 1), 1):.2f} VALU per MFMA, {full} x vmcnt(0))")


if __name__ == "__main__":
    main()
