# usage: tools/pmc_any.sh OUT.json COUNTER [COUNTER ...]   (one stream group of 512 streams; every kernel alone on the chip)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf /tmp/pmc_any
ALEGO_STREAM_GROUPS=1 timeout 900 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_any -o sq --output-format csv -- python bench.py --streams 512 --steps 6 --warmup 0 --prime 700 --no-cpu --no-profile --no-check --no-isolated < /dev/null > /tmp/pmc_any.log 2>&1
python tools/pmc_agg.py /tmp/pmc_any 12 > $out; tail -2 /tmp/pmc_any.log
