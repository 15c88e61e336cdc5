#!/bin/bash
# A/B of library variants on another geometry.  usage: tools/geo_ab.sh "libA.so libB.so" <bench args...>
libs=$1; shift
for l in $libs; do for rep in 1 2; do
  v=$(ALEGO_LIB=$PWD/a-lego-loam_amd/$l python bench.py "$@" --no-cpu --no-profile --no-check --no-isolated 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])")
  echo "$l $v"
done; done
