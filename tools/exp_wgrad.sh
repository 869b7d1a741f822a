#!/bin/bash
# timing-only ablations of the weight-gradient kernels (results are WRONG by construction). Build the libraries first:
#   make -C yolov5m_amd/csrc exp EXP="1 2 4 6 7 8 16 32 48 64 128 512 560"
# wgrad_kernel (4 waves):  bit0 (1) = one ds_read_b128 per fragment instead of two ds_read_b64_tr_b16, bit1 (2) = no LDS
#   stores after the first chunk (the loads then are dead code too), bit2 (4) = no global loads after the first chunk,
#   bit3 (8) = no MFMAs (fragment reads kept alive by an xor)
# wgrad_pc_kernel (Y5M_WGRAD_PC): 16 = producers do not store, 32 = producers do not load, 64 = consumers without MFMAs,
#   128 = consumers without fragment reads, 256 = no barrier in the loop, 512 = no atomics
shape="${SHAPE:-64 192 40 40 192 3 1 50}"
for e in 0 1 2 4 6 7 8; do
  if [ $e = 0 ]; then unset Y5M_LIB; else export Y5M_LIB=$PWD/build/exp/liby5m_e$e.so; fi
  [ $e = 0 ] || [ -f "$Y5M_LIB" ] || continue
  echo -n "4-wave kernel exp=$e  "; Y5M_WGRAD_PC=0 python tools/conv_bench.py wgrad $shape 2>/dev/null
done
for e in 0 16 32 48 64 128 512 560; do
  if [ $e = 0 ]; then unset Y5M_LIB; else export Y5M_LIB=$PWD/build/exp/liby5m_e$e.so; fi
  [ $e = 0 ] || [ -f "$Y5M_LIB" ] || continue
  for pc in 1 9; do echo -n "producer/consumer pc=$pc exp=$e  "; Y5M_WGRAD_PC=$pc python tools/conv_bench.py wgrad $shape 2>/dev/null; done
done
