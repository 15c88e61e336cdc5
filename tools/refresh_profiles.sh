#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh r02 < /dev/null'): writes gpurun_out/<round>_* ; copy them to profiles/ afterwards.
#   ${R}_bench.json                 the default bench.py line (HIP-event kernel table, roofline, cpu_baseline, parity)
#   ${R}_bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats (without the CPU legs)
#   ${R}_bench_kernel_stats.csv     rocprofv3's per-kernel summary of that run (all dispatches, priming included)
#   ${R}_bench_kernel_stats_steady.csv  the same trace restricted to the dispatches of bench.py's HIP-event pass, side by side
#   ${R}_pmc_sq.json                SQ counters per kernel (waves, wave cycles, VALU instructions, wait cycles)
#   ${R}_pmc_traffic.json           HBM bytes per launch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), taken after 700
#                                  priming scans: around scan 560 every stream fills its 50-key-frame window and rebuilds its map at once
set -u
R=${1:-r06}
MODE=${2:-all}    # all | traffic (only the two --pmc traffic passes) | pmc (traffic + SQ counters); the last two need gpurun_out/${R}_bench.json from an earlier call
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -rf /tmp/prof_stats /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
if [ "$MODE" = "all" ]; then
timeout 900 python bench.py < /dev/null > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o st --output-format csv -- python bench.py --no-cpu --no-check --no-isolated < /dev/null > gpurun_out/${R}_bench_under_rocprof.json 2> /tmp/st.log
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_bench_kernel_stats.csv \;
python tools/trace_tail_stats.py /tmp/prof_stats gpurun_out/${R}_bench_under_rocprof.json > gpurun_out/${R}_bench_kernel_stats_steady.csv
fi
[ -s gpurun_out/${R}_bench.json ] || timeout 600 python bench.py --no-cpu < /dev/null > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
PER=$(R=$R python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s_bench.json" % os.environ["R"]).read().strip().splitlines()[-1])
print(d["roofline"]["streams_per_launch"])
PY
)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- python bench.py --steps 8 --warmup 0 --prime 700 --no-cpu --no-profile --no-check --no-isolated < /dev/null > /tmp/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$PER" 16 > gpurun_out/${R}_pmc_traffic.json
[ "$MODE" = "traffic" ] && { ls -la gpurun_out/${R}_*; exit 0; }
# SQ counters (occupancy / issue statistics quoted in DESIGN.md section 4): one stream group, 512 streams
rm -rf /tmp/pmc_sq
ALEGO_STREAM_GROUPS=1 timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_sq -o sq --output-format csv -- python bench.py --streams 512 --steps 6 --warmup 0 --prime 700 --no-cpu --no-profile --no-check --no-isolated < /dev/null > /tmp/pmc_sq.log 2>&1
python tools/pmc_agg.py /tmp/pmc_sq 12 512 > gpurun_out/${R}_pmc_sq.json
# read requests by size (round 5): nearly every request of this pipeline is 128 bytes wide, so exact read bytes = 128 x TCC_EA0_RDREQ_128B + 64 x ..._64B + 32 x ..._32B
rm -rf /tmp/pmc_any
bash tools/pmc_any.sh gpurun_out/${R}_pmc_rdreq.json TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
ls -la gpurun_out/${R}_*
