#!/bin/bash
# 64x2048 / K = 200 throughput against the number of resident streams.  usage: tools/geo64_streams.sh "512 768 896"
for s in $1; do
  timeout 900 python bench.py --geometry 64x2048 --keyframes 200 --kf-cap 8192 --streams $s --bags 4 --prime 2400 --steps 40 --warmup 10 --no-cpu --no-check --no-isolated --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(sys.argv[1], j['value'], j['ms_per_step'], j.get('truncated_streams'))" $s
done
