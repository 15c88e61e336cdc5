# VALU / SALU / LDS instructions of fe_ring_out by phase: libraries built with -DFO_STOP_AFTER=<tick> stop every ring after that phase; the differences
# between consecutive stops are the phases.  usage (GPU box): bash tools/fo_phase_counts.sh   (after: for k in 1 2 3 12 4 5 13 7; do
#   ALEGO_EXTRA_FLAGS=-DFO_STOP_AFTER=$k ALEGO_BUILD_DIR=build_s$k ALEGO_SO=libalego_s$k.so bash a-lego-loam_amd/build.sh; done)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 1 2 3 12 4 5 13 7 full; do
  lib=$PWD/a-lego-loam_amd/libalego_s$k.so; [ $k = full ] && lib=$PWD/a-lego-loam_amd/libalego_mi355x.so
  rm -rf /tmp/fo_pc
  ALEGO_LIB=$lib ALEGO_STREAM_GROUPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/fo_pc -o sq --output-format csv -- python tools/kernel_times.py 512 3 4 2 < /dev/null > /tmp/fo_pc.log 2>&1
  python tools/pmc_agg.py /tmp/fo_pc 0 | python -c "import sys,json; d=json.load(sys.stdin)['fe_ring_out']; print('stop after $k:', {a: round(b/512) for a,b in d.items() if a!='dispatches'})"
done
