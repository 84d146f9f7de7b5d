"""the shuffled-banded probe of bench.py on its own.  usage: python scripts/gpu_reorder_probe.py [m n]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ipx_amd import kkt, synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 2000000)
os.environ["IPXK_VERBOSE"] = "1"
print(json.dumps(bench.bench_banded_shuffled(kkt, synth, m, n), indent=1))
