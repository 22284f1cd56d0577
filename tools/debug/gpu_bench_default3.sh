#!/bin/bash
# the driver's command line a few times over: spread of `value` (region 0)
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('preheat', d['preheat_ms'], 'value %.2f M' % (d['value'] / 1e6), 'frac %.3f' % d['roofline']['frac'], 'kernel ms per region', [round(x, 2) for x in d['repeats']['kernel_ms']])"
done
