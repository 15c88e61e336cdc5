#!/bin/bash
# VALU / SALU / LDS instructions of lo_assoc by phase (round 6): libraries built with -DLA_STOP_AFTER=<tick> end every sweep after that phase; consecutive differences are the phases.
# Build first:  for k in 1 4 5 6 7; do ALEGO_EXTRA_FLAGS=-DLA_STOP_AFTER=$k ALEGO_BUILD_DIR=build_la$k ALEGO_SO=libalego_la$k.so bash a-lego-loam_amd/build.sh; done
# usage (GPU box): bash tools/la_phase_counts.sh [streams] [geometry] [keyframes]      IP + FE + LO only (stages 3)
B=${1:-512}; GEO=${2:-16x1800}; KF=${3:-0}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 1 4 5 6 7 full; do
  lib=$PWD/a-lego-loam_amd/libalego_la$k.so; [ $k = full ] && lib=$PWD/a-lego-loam_amd/libalego_mi355x.so
  rm -rf /tmp/la_pc
  ALEGO_LIB=$lib ALEGO_STREAM_GROUPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/la_pc -o sq --output-format csv -- python tools/kernel_times.py $B 3 4 6 $GEO $KF < /dev/null > /tmp/la_pc.log 2>&1
  python tools/pmc_agg.py /tmp/la_pc 0 | B=$B python -c "import sys,json,os; d=json.load(sys.stdin); b=int(os.environ['B']); print('stop after $k:', {k: {a: round(v/b) for a,v in c.items() if a.startswith('SQ_')} for k,c in d.items() if k.startswith('lo_assoc')})"
done
