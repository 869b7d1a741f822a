#!/bin/bash
# Run a python command with the AddressSanitizer build of the CPU lane-level executor (tests/emu/build.py, Y5M_EMU_ASAN=1):
#   tests/emu/asan_run.sh tests/emu/run_gpu_tests.py test_gpu_conv
# Every load / store a kernel makes on a tensor is checked against the tensor's allocation (torch's CPU allocator goes through the
# interposed malloc). Leak detection is off (python), stack instrumentation is off (fibers).
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export Y5M_EMU_ASAN=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:detect_stack_use_after_return=0:use_sigaltstack=0
LD_PRELOAD=$RT exec python -u "$@"
