for cfg in "1024 8" "512 8" "256 8" "512 16" "1024 16" "2048 8" "768 12"; do set -- $cfg; echo "== blocks=$1 minch=$2";
for a in "64 48 160 160 48 1 1" "64 192 40 40 192 1 1" "64 192 40 40 192 3 1" "64 16 320 320 48 3 1" "64 384 20 20 384 1 1"; do Y5M_WGRAD_BLOCKS=$1 Y5M_WGRAD_MINCH=$2 python tools/conv_bench.py wgrad $a 30 2>/dev/null; done; done
