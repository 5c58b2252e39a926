for a in 0 $((64+256*100)) $((64+256*300)) $((64+256*600)) $((64+256*1200)); do echo "ablate=$a"; PNODE_ABLATE=$a python bench.py --steps 20 --warmup 15 --no-cpu-baseline --diag-fixed-dt 0.01 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print(d['roofline']['phase_ms']['lq'], d['ms_per_step'])"; done
