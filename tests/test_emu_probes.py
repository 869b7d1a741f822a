"""CPU: the wave-level primitives the executor models by hand (MFMA lane layouts, ds_read_b64_tr_b16, DPP, cross-lane reads, buffer-
resource bounds, LDS-DMA placement) produce the committed table tests/golden/emu_probes.npz -- and the two MFMA shapes equal a plain
numpy matrix product on asymmetric operands, a truth that does not depend on the executor. The same table is what
tests/test_gpu_emu_probes.py holds the HARDWARE against."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

pytestmark = pytest.mark.skipif(not os.path.exists(os.environ.get("Y5M_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")),
                                reason="host clang of the ROCm image not present")


def test_executor_reproduces_the_probe_table(golden):
    import probes
    g = golden("emu_probes")
    r = probes.run_emu()
    assert set(r) == set(g.files)
    for k in r:
        assert np.array_equal(r[k], g[k]), k


def test_mfma_models_equal_a_numpy_matrix_product():
    import probes
    r, e = probes.run_emu(), probes.expected_mfma()
    for k in ("mfma_bf16", "mfma_f32"):
        assert np.array_equal(r[k], e[k]), k
        assert np.abs(e[k].view(np.float32)).max() > 1.0


def test_probe_library_for_the_gpu_builds():
    """hipcc cross-compiles the probe kernels for gfx950 (the library travels with the snapshot; __graft_entry__.build() makes it)"""
    import probes
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    assert os.path.exists(probes.build_gpu())
