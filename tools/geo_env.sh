#!/bin/bash
# bench throughput on another geometry for several environments.  usage: tools/geo_env.sh "ENV=a ENV=b" <bench args...>
envs=$1; shift
for e in $envs; do for rep in 1 2; do
  v=$(env $e python bench.py "$@" --no-cpu --no-profile --no-check --no-isolated 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])")
  echo "$e $v"
done; done
