for b in 128 256 512 768 1024 2048; do echo "B=$b"; python bench.py --steps 20 --warmup 15 --no-cpu-baseline --diag-fixed-dt 0.01 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print(d['roofline']['phase_ms'], d['ms_per_step'])"; done
