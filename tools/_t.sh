timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "fe_ or feature or teacher or random_param or geometr" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf /tmp/fo_pc
ALEGO_STREAM_GROUPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/fo_pc -o sq --output-format csv -- python tools/kernel_times.py 512 3 4 2 < /dev/null > /tmp/fo_pc.log 2>&1
python tools/pmc_agg.py /tmp/fo_pc 0 | python -c "import sys,json; d=json.load(sys.stdin)['fe_ring_out']; print('full:', {a: round(b/512) for a,b in d.items() if a!='dispatches'})"
bash tools/sweep.sh "libalego_base.so libalego_mi355x.so libalego_base.so libalego_mi355x.so" "X=1" 2048 60
