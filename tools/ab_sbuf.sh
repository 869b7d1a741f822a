run() { echo "== $1"; env $1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'],'img/s', d['ms_per_step'],'ms', 'conv',r['family_ms_per_step'].get('conv_igemm'),'wgrad',r['family_ms_per_step'].get('wgrad'))
"; }
run "Y5M_CONV_SBUF_KT=0 Y5M_WGRAD_SBUF=0"
run "Y5M_CONV_SBUF_KT=3 Y5M_WGRAD_SBUF=0"
run "Y5M_CONV_SBUF_KT=6 Y5M_WGRAD_SBUF=0"
run "Y5M_CONV_SBUF_KT=12 Y5M_WGRAD_SBUF=0"
run "Y5M_CONV_SBUF_KT=0 Y5M_WGRAD_SBUF=1"
run "Y5M_CONV_SBUF_KT=6 Y5M_WGRAD_SBUF=1"
