"""Developer tool: the non-default dispatch paths that are still in the tree (environment knobs of DESIGN.md section 8), each in
a fresh process on the CPU lane-level executor: layer-local bf16 backward parity of the whole model at 2 x 64 x 64
(tests/test_gpu_model.py::_per_layer_backward) and the bf16 forward against the oracle. Knobs are read once per process, so
every variant is a child. Usage: python tests/emu/knob_variants.py [--list]     TEST INFRASTRUCTURE."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = ["", "Y5M_R4_KERNELS=31 Y5M_POOL_TILE=1 Y5M_HEAD_PACK16=1 Y5M_CONV_HALO_NS2=1 Y5M_PACK_ONCE=1",     # (every staged, never-timed variant at once)
            "Y5M_R4_KERNELS=31", "Y5M_POOL_TILE=1", "Y5M_HEAD_PACK16=1", "Y5M_CONV_HALO_NS2=1", "Y5M_BWD_PW=0", "Y5M_BWD_PW_MIN_M=0", "Y5M_BWD_STEM=0", "Y5M_LAZY_RES=0", "Y5M_MERGE_C3=0", "Y5M_WGRAD_DIRECT=0",
            "Y5M_WGRAD_ROWS=0", "Y5M_WGRAD_AFTER_DGRAD=0", "Y5M_SLOTS=2", "Y5M_SPARSE_HEAD=0", "Y5M_BN_FUSE=0", "Y5M_CONV_HALO=0",
            "Y5M_CONV_GEMM8=0", "Y5M_CONV_GEMM8=1", "Y5M_CONV_PW=0", "Y5M_CONV_MULTI=0"]
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(%r))); sys.path.insert(0, os.path.dirname(%r))
import torch
from emu.harness import emulated
import test_gpu_model as T
T.DEV = "cpu"
with emulated():
    worst, ndx, names, kinds = T._per_layer_backward(2, 64)
    T.test_forward_bf16_vs_oracle("train")
    T.test_forward_bf16_vs_oracle("eval")
print("RESULT worst", {k: "%%.1e" %% v for k, v in worst.items()}, "dx-checked", ndx, "bwd_pw", kinds.count("bwd_pw"), "bwd_stem", kinds.count("bwd_stem"),
      "kernels", len(names))
''' % (HERE, HERE)

if __name__ == "__main__":
    if "--list" in sys.argv:
        print("\n".join(v or "(default)" for v in VARIANTS))
        sys.exit(0)
    bad = 0
    for v in VARIANTS:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=")
            env[k] = val
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=1800)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        ok = r.returncode == 0 and line
        bad += 0 if ok else 1
        print(f"{v or '(default)':40s} {'ok  ' + line[0][7:] if ok else 'FAIL ' + (r.stderr.strip().splitlines() or ['?'])[-1][:200]}", flush=True)
    sys.exit(1 if bad else 0)
