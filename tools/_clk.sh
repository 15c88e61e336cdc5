timeout 300 python bench.py --no-cpu --no-profile --steps 4000 > /tmp/b.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do sleep 1.5; echo "$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ')"; done
wait $BP
python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
