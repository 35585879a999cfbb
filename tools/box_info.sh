#!/bin/bash
# What kind of box is this?  (the pool's boxes differ by 1.5 x on the latency-chain kernels)
for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/power_dpm_force_performance_level; do [ -r $f ] && echo "$f: $(cat $f)"; done 2>/dev/null
rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>/dev/null | grep -E "GPU\[0\]|Power|sclk|mclk|fclk|Performance|Max" | head -14
python - <<'PY'
import json,subprocess,sys
out=subprocess.run([sys.executable,'bench.py','--no-cpu-baseline','--no-extras','--steps','200','--warmup','30'],capture_output=True,text=True).stdout
d=json.loads([l for l in out.split('\n') if l.startswith('{')][-1])
print('value',d['value'],d['stage_ms'])
PY
