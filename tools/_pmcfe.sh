cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ALEGO_STREAM_GROUPS=1
for v in 0 1; do
  rm -rf /tmp/pf$v
  ALEGO_FE_PICK1=$v timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pf$v -o sq --output-format csv -- python tools/kernel_times.py 384 3 6 10 > /tmp/pf$v.log 2>&1
  python tools/pmc_agg.py /tmp/pf$v 6 | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'fe_pick' in k or 'cc_lds16' in k or 'fe_voxel' in k: print(k, {a:round(b) for a,b in v.items()})
"
done
