#!/bin/bash
# Is the first timed region slower than the ones after it (clock state after the sparse warm-up launches)?
for ph in 0 0 0 300 300 300; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --preheat-ms $ph | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('preheat', d['preheat_ms'], 'value %.2f M' % (d['value'] / 1e6), 'kernel ms per region', [round(x, 2) for x in d['repeats']['kernel_ms']])"
done
