#!/bin/bash
# timing-only ablations of wgrad_kernel (results are WRONG by construction; libs built with -DY5M_EXP=n into build/exp/):
# bit0 (1) = one ds_read_b128 per fragment instead of two ds_read_b64_tr_b16, bit1 (2) = no LDS stores after the first
# chunk, bit2 (4) = no global loads after the first chunk, bit3 (8) = no MFMAs (fragment reads kept alive by an xor)
for pct in 50 100; do
for shape in "64 192 40 40 192 3 1 50" "64 96 80 80 96 3 1 50" "64 384 20 20 384 3 1 50"; do
  for e in 0 1 2 4 6 7 8; do
    if [ $e = 0 ]; then unset Y5M_LIB; else export Y5M_LIB=$PWD/build/exp/liby5m_e$e.so; fi
    echo -n "res_pct=$pct exp=$e  "; Y5M_WGRAD_RES_PCT=$pct python tools/conv_bench.py wgrad $shape 2>/dev/null
  done
done
done
