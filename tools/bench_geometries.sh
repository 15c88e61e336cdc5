#!/bin/bash
# Run on the GPU box: bench lines + rocprofv3 kernel summaries for the reference geometry (16x4000, utility.h:50-55) and for config 5's
# geometry (64x2048 with a 200-key-frame local map).  Outputs gpurun_out/<round>_geo_*; copy to profiles/ afterwards.
set -u
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {  # tag, bench args...
  local tag=$1; shift
  timeout 1200 python bench.py "$@" --no-cpu < /dev/null > gpurun_out/${R}_geo_${tag}.json 2> gpurun_out/${R}_geo_${tag}.err
  rm -rf /tmp/prof_geo
  timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_geo -o st --output-format csv -- python bench.py "$@" --no-cpu --no-profile < /dev/null > gpurun_out/${R}_geo_${tag}_under_rocprof.json 2> /tmp/geo.log
  find /tmp/prof_geo -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_geo_${tag}_kernel_stats.csv \;
  # HBM traffic per launch: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), as tools/refresh_profiles.sh does for the default workload
  local per=$(python - <<PY
import json
d = json.loads(open("gpurun_out/${R}_geo_${tag}.json").read().strip().splitlines()[-1])
print(d["roofline"]["streams_per_launch"])
PY
)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcg_$c
    timeout 1200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcg_$c -o p --output-format csv -- python bench.py "$@" --steps 8 --warmup 0 --no-cpu --no-profile --no-check --no-isolated < /dev/null > /tmp/pmcg_$c.log 2>&1
  done
  python tools/pmc_traffic.py /tmp/pmcg_FETCH_SIZE /tmp/pmcg_WRITE_SIZE "$per" 16 > gpurun_out/${R}_geo_${tag}_pmc_traffic.json
  tail -c 600 gpurun_out/${R}_geo_${tag}.json; echo; head -8 gpurun_out/${R}_geo_${tag}_kernel_stats.csv
}
run 16x4000 --geometry 16x4000 --kf-cap 16384 --streams 2304 --steps 60 --warmup 10
run 64x2048_k200 --geometry 64x2048 --keyframes 200 --kf-cap 8192 --streams 768 --bags 4 --prime 2400 --steps 40 --warmup 10
# SQ instruction counters of config 5's geometry, every kernel alone on the chip (ImageProjection's share of the VALU instructions)
bash tools/pmc_geo_sq.sh gpurun_out/${R}_geo_64x2048_pmc_sq.json
ls -la gpurun_out/${R}_geo_*
