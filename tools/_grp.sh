for cfg in "4 4" "6 8" "8 8" "8 4" "2 4"; do
  set -- $cfg
  echo "groups=$1 hwq=$2: $(ALEGO_STREAM_GROUPS=$1 GPU_MAX_HW_QUEUES=$2 timeout 200 python bench.py --no-cpu --no-profile --steps 100 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
