for a in 0 1 2 3 7; do echo "ablate=$a"; PNODE_ABLATE=$a python bench.py --steps 20 --warmup 15 --no-cpu-baseline --diag-fixed-dt 0.01 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print(d['roofline']['phase_ms'], d['ms_per_step'])
except Exception as e: print('fail', e)"; done
