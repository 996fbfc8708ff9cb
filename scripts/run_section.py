"""One section of bench.py by name: python3 scripts/run_section.py api_single_call [args as python literals]"""
import ast
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402

fn = getattr(bench, sys.argv[1] + "_section")
args = [ast.literal_eval(a) for a in sys.argv[2:]]
takes_dev = "dev" in fn.__code__.co_varnames[:fn.__code__.co_argcount]
res = fn(torch.device("cuda", 0), *args) if takes_dev else fn(*args)
print(json.dumps(res, indent=1, default=str))
