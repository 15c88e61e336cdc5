#!/bin/bash
# development sweep: bench throughput for (library variant, environment) pairs.  usage: tools/sweep.sh "lib1 lib2" "ENV=a ENV=b" [streams] [steps]
libs=$1; envs=$2; streams=${3:-2048}; steps=${4:-60}; extra=${SWEEP_ARGS:-}
for l in $libs; do for e in $envs; do
  v=$(env $e ALEGO_LIB=$PWD/a-lego-loam_amd/$l python bench.py --streams $streams --steps $steps --warmup 10 --no-cpu --no-profile --no-check --no-isolated $extra 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'])")
  echo "$l $e $v"
done; done
