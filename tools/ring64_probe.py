"""The trajectory leg of bench.py on its own (GPU box): a 64-slot trajectory buffer in one plain allocation against a ring of 64
separately placed sets (sgx_step_ring, pointers in a device table, one launch).  usage: python tools/ring64_probe.py [slots]"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bench_legs

args = bench.parse_args([])
rk = types.SimpleNamespace(rank=0, world=1, device_index=0)
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print(json.dumps(bench_legs.trajectory_leg(bench, rk, args, slots=slots), indent=1))
