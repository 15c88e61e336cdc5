#!/bin/bash
# usage (GPU box, repo root): bash tools/probe/run.sh  -> bench throughput alone and next to each synthetic load
run() { timeout 200 python bench.py --no-cpu --no-profile --steps 300 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"])'; }
echo "alone: $(run)"
for cfg in "valu 1" "valu 2" "lds 1" "lds 2" "mem 2"; do
  set -- $cfg
  ./tools/probe/interfere $1 $2 60 > /tmp/probe_$1_$2.log 2>&1 &
  PID=$!
  echo "$1 $2: bench $(run)   probe: $(sleep 1; wait $PID; cat /tmp/probe_$1_$2.log)"
done
# the probes alone (their own rate without the pipeline next to them)
for cfg in "valu 1" "lds 1" "mem 2"; do set -- $cfg; echo "alone $1 $2: $(./tools/probe/interfere $1 $2 5)"; done
