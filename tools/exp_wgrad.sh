#!/bin/bash
# timing-only ablations of wgrad_kernel (results are WRONG by construction; libs built with -DY5M_EXP=n):
# bit1 (2) = global loads hit the zero page, bit2 (4) = no atomics, bit3 (8) = no ds_read/MFMA after the first chunk
for shape in "64 192 40 40 192 3 1 50" "64 96 80 80 96 3 1 50" "64 384 20 20 384 3 1 50" "64 192 40 40 192 1 1 100" "64 96 80 80 96 1 1 100" "64 384 40 40 192 1 1 100"; do
  for e in 0 2 4 8 6 14; do
    if [ $e = 0 ]; then unset Y5M_LIB; else export Y5M_LIB=$PWD/build/exp/liby5m_e$e.so; fi
    echo -n "exp=$e  "; python tools/conv_bench.py wgrad $shape 2>/dev/null
  done
done
