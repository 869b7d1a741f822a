#!/bin/bash
# isolated pointwise launches (conv_bench.py): shapes of the model + single-chunk / two-chunk probes
for shape in "$@"; do
  for kind in fwd dgrad; do echo -n "occ=${Y5M_CONV_PW_OCC:-4} $kind $shape: "; python tools/conv_bench.py $kind $shape 100 2>/dev/null | tail -1 | sed 's/.*: //'; done
done
