"""Developer tool: run the test functions of a tests/test_gpu_*.py module on the CPU lane-level executor and print one line
per parametrised case (status, seconds). Usage: python tests/emu/run_gpu_tests.py test_gpu_conv [name filter] [--skip=regex] [--only=regex]
TEST INFRASTRUCTURE (tests/test_emu_kernels.py holds the curated subset that runs in the -m "not gpu" suite)."""
import importlib
import itertools
import os
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


def expand(fn):
    """[(id, kwargs)] of a test function's @pytest.mark.parametrize product"""
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    axes = []
    for m in marks:
        names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
        vals = []
        for v in m.args[1]:
            v = getattr(v, "values", v) if hasattr(v, "values") and hasattr(v, "marks") else v
            if len(names) == 1:
                vals.append({names[0]: v[0] if (isinstance(v, tuple) and hasattr(m.args[1][0], "marks")) else v})
            else:
                vals.append(dict(zip(names, v)))
        axes.append(vals)
    out = []
    for combo in itertools.product(*axes) if axes else [()]:
        kw = {}
        for d in combo:
            kw.update(d)
        out.append(kw)
    return out


def main():
    import torch
    from emu.harness import emulated
    modname = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    import re
    skip = only = None
    for a in sys.argv[2:]:
        if a.startswith("--skip="):
            skip = re.compile(a[7:])
        if a.startswith("--only="):
            only = re.compile(a[7:])
    mod = importlib.import_module(modname)
    mod.DEV = "cpu"
    import numpy as np
    cache = {}

    def golden(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(TESTS, "golden", name + ".npz"), allow_pickle=False)
        return cache[name]
    with emulated():
        for name in sorted(n for n in dir(mod) if n.startswith("test_") and filt in n):
            fn = getattr(mod, name)
            if not callable(fn) or (skip is not None and skip.search(name)) or (only is not None and not only.search(name)):
                continue
            import inspect
            params = inspect.signature(fn).parameters
            for kw in expand(fn):
                if "golden" in params:
                    kw = dict(kw, golden=golden)
                mp = None
                if "monkeypatch" in params:
                    import pytest
                    mp = kw["monkeypatch"] = pytest.MonkeyPatch()
                for pn in params:                      # module-level fixtures without arguments
                    fx = getattr(mod, pn, None)
                    if pn not in kw and fx is not None and hasattr(fx, "_get_wrapped_function"):
                        kw[pn] = fx._get_wrapped_function()()
                t = time.time()
                try:
                    fn(**kw)
                    st = "ok"
                except BaseException as e:  # noqa: BLE001
                    st = "FAIL " + type(e).__name__ + ": " + str(e).replace("\n", " ")[:300]
                    if os.environ.get("EMU_TB"):
                        traceback.print_exc()
                finally:
                    if mp is not None:
                        mp.undo()
                show = {k: v for k, v in kw.items() if k not in ("golden", "monkeypatch") and not hasattr(getattr(mod, k, None), "_get_wrapped_function")}
                print(f"{time.time() - t:7.2f}s {name} {show} {st}", flush=True)


if __name__ == "__main__":
    main()
