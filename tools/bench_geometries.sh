#!/bin/bash
# Run on the GPU box: bench lines + rocprofv3 kernel summaries for the reference geometry (16x4000, utility.h:50-55) and for config 5's
# geometry (64x2048 with a 200-key-frame local map).  Outputs gpurun_out/<round>_geo_*; copy to profiles/ afterwards.
set -u
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() {  # tag, bench args...
  local tag=$1; shift
  timeout 1200 python bench.py "$@" --no-cpu < /dev/null > gpurun_out/${R}_geo_${tag}.json 2> gpurun_out/${R}_geo_${tag}.err
  rm -rf /tmp/prof_geo
  timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_geo -o st --output-format csv -- python bench.py "$@" --no-cpu --no-profile < /dev/null > gpurun_out/${R}_geo_${tag}_under_rocprof.json 2> /tmp/geo.log
  find /tmp/prof_geo -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_geo_${tag}_kernel_stats.csv \;
  tail -c 600 gpurun_out/${R}_geo_${tag}.json; echo; head -8 gpurun_out/${R}_geo_${tag}_kernel_stats.csv
}
run 16x4000 --geometry 16x4000 --streams 768 --steps 60 --warmup 10
run 64x2048_k200 --geometry 64x2048 --keyframes 200 --kf-cap 8192 --streams 512 --bags 4 --prime 2400 --steps 40 --warmup 10
ls -la gpurun_out/${R}_geo_*
