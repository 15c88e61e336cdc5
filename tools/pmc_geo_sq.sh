#!/bin/bash
# SQ instruction counters per kernel for config 5's geometry (64x2048, 128 streams per launch, one stream group: every kernel alone on the chip).
# usage (GPU box): tools/pmc_geo_sq.sh <out.json> [extra bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf /tmp/pmc_gsq
ALEGO_STREAM_GROUPS=1 timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_gsq -o sq --output-format csv -- \
  python bench.py --geometry 64x2048 --keyframes 200 --kf-cap 8192 --streams 128 --bags 4 --prime 300 --steps 6 --warmup 0 --no-cpu --no-profile --no-check --no-isolated "$@" < /dev/null > /tmp/pmc_gsq.log 2>&1
python tools/pmc_agg.py /tmp/pmc_gsq 6 128 > "$out"; tail -2 /tmp/pmc_gsq.log | cut -c1-300
