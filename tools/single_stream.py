"""Single-stream replay (BASELINE config 3 as written): serial one-slot replay vs alego_stream_run.  usage: single_stream.py [lanes] [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alego_loader import load_package; load_package()
from alego_amd import synth
import bench
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
p = synth.default_params(16, 1800)
bags = bench.make_bags(p, 1, 0)
print(json.dumps(bench.single_stream(p, bags[0], 0, 560, steps, lanes)))
