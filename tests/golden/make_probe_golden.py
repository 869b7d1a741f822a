"""tests/golden/emu_probes.npz: what the CPU lane-level executor (tests/emu) returns for the wave-primitive probes of
tests/emu/probes.hip. The CPU suite pins the executor to it (tests/test_emu_probes.py); the GPU suite holds the HARDWARE against
it (tests/test_gpu_emu_probes.py). usage: python tests/golden/make_probe_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "emu"))
import probes  # noqa: E402

if __name__ == "__main__":
    r = probes.run_emu()
    np.savez_compressed(os.path.join(HERE, "emu_probes.npz"), **r)
    print({k: v.shape for k, v in r.items()})
