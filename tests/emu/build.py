"""Builds build/emu/liby5m_emu.so: the kernel sources of yolov5m_amd/csrc compiled for x86-64 against the CPU lane-level
executor (tests/emu/include). TEST INFRASTRUCTURE -- same C ABI as liby5m.so, loaded only by tests/emu/harness.py."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "yolov5m_amd", "csrc")
# Y5M_EMU_ASAN=1: the same library with AddressSanitizer on every global / heap access of the kernels (stack instrumentation off:
# the fibers switch stacks behind the sanitizer's back). Load it in a python started with LD_PRELOAD=<asan runtime>
# (tests/emu/asan_run.sh): an out-of-bounds read or write of a kernel on a tensor -- silent on the GPU -- aborts with a report.
ASAN = os.environ.get("Y5M_EMU_ASAN") == "1"
FAST_EXP = os.environ.get("Y5M_EMU_FAST_EXP") == "1"       # __expf as the GPU computes it (see include/hip/hip_runtime.h)
RCP_ULP = os.environ.get("Y5M_EMU_RCP_ULP", "")          # systematic n-ulp bias of v_rcp_f32 (sensitivity experiments)
OUT_DIR = os.path.join(ROOT, "build", "emu_asan" if ASAN else ("emu_fastexp" if FAST_EXP else "emu") + (f"_rcp{RCP_ULP}" if RCP_ULP else ""))
LIB = os.path.join(OUT_DIR, "liby5m_emu.so")
CXX = os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-pthread",
         "-I", os.path.join(HERE, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "include")]
if FAST_EXP:
    FLAGS += ["-DY5M_EMU_FAST_EXP"]
if RCP_ULP:
    FLAGS += [f"-DY5M_EMU_RCP_ULP={int(RCP_ULP)}"]
if ASAN:
    FLAGS += ["-fsanitize=address", "-shared-libasan", "-mllvm", "-asan-stack=0", "-fno-omit-frame-pointer", "-g1"]
EXACT = {"y5m_detect.hip", "y5m_loss.hip"}          # same rule as csrc/Makefile: no FMA contraction in the bit-exact units

sys.path.insert(0, HERE)
from translate import translate  # noqa: E402


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    return h.hexdigest()


def build(verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = ([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] +
            [os.path.join(ROOT, "include", "y5m.h"), os.path.join(HERE, "emu_rt.cpp"), os.path.join(HERE, "translate.py"),
             os.path.join(HERE, "build.py"), os.path.join(HERE, "include", "emu_rt.h"),
             os.path.join(HERE, "include", "hip", "hip_runtime.h")])
    stamp = os.path.join(OUT_DIR, "stamp")
    dig = _digest(deps)
    if os.path.exists(LIB) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read() == dig:
                return LIB

    def one(name):
        src = os.path.join(CSRC, name)
        cpp = os.path.join(OUT_DIR, name.replace(".hip", ".cpp"))
        with open(src) as f:
            text = translate(f.read(), src)
        with open(cpp, "w") as f:
            f.write(f'#line 1 "{src}"\n' + text)
        obj = cpp[:-4] + ".o"
        cmd = [CXX] + FLAGS + (["-ffp-contract=off"] if name in EXACT else []) + ["-c", cpp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emu build of {name} failed:\n{r.stderr[-6000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, srcs))
    rt = os.path.join(OUT_DIR, "emu_rt.o")
    subprocess.check_call([CXX] + FLAGS + ["-c", os.path.join(HERE, "emu_rt.cpp"), "-o", rt])
    subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if ASAN else []) +
                          ["-o", LIB] + objs + [rt])
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


def build_variant(tag, edits):
    """build/emu/liby5m_emu_<tag>.so: the executor's library with some kernel sources EDITED (edits: {file name: function text ->
    text}) -- the deliberately broken kernels of tests/test_emu_checks.py. Only the edited translation units are recompiled; the
    rest are the objects of the regular build."""
    build()
    vdir = os.path.join(OUT_DIR, "variant_" + tag)
    os.makedirs(vdir, exist_ok=True)
    lib = os.path.join(OUT_DIR, f"liby5m_emu_{tag}.so")
    objs = []
    for name in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        if name not in edits:
            objs.append(os.path.join(OUT_DIR, name.replace(".hip", ".o")))
            continue
        src = os.path.join(CSRC, name)
        with open(src) as f:
            text = f.read()
        new = edits[name](text)
        if new == text:
            raise RuntimeError(f"variant {tag}: the edit of {name} changed nothing (the source has moved on: update the test)")
        cpp = os.path.join(vdir, name.replace(".hip", ".cpp"))
        with open(cpp, "w") as f:
            f.write(f'#line 1 "{src}"\n' + translate(new, src))
        obj = cpp[:-4] + ".o"
        r = subprocess.run([CXX] + FLAGS + (["-ffp-contract=off"] if name in EXACT else []) + ["-c", cpp, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"emu variant build of {name} failed:\n{r.stderr[-4000:]}")
        objs.append(obj)
    subprocess.check_call([CXX, "-shared", "-fPIC", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if ASAN else []) +
                          ["-o", lib] + objs + [os.path.join(OUT_DIR, "emu_rt.o")])
    return lib


if __name__ == "__main__":
    build(verbose=True)
