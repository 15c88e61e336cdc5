cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf /tmp/pmc_sq
ALEGO_STREAM_GROUPS=1 timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_sq -o sq --output-format csv -- python bench.py --streams 512 --steps 6 --warmup 0 --prime 700 --no-cpu --no-profile --no-check --no-isolated < /dev/null > /tmp/pmc_sq.log 2>&1
python tools/pmc_agg.py /tmp/pmc_sq 12 > gpurun_out/r03_pmc_sq_a.json; tail -3 /tmp/pmc_sq.log
