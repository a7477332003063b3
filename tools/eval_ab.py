"""GPU box: the validation leg of bench.py (config 2's model at list size 10 and at 100 candidates) under the current knobs:
   ULTR_FWD_WIDE=0 python tools/eval_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ultra_pytorch_amd import _lib
from ultra_pytorch_amd.ranking_model import init_flat_params
from ultra_pytorch_amd import hip_ops
lib = _lib.load()
cfg = bench.CONFIGS["2"]
shape = hip_ops.DnnShape(cfg["F"], cfg["hidden"], "elu")
p0 = init_flat_params(shape, 0).numpy()
for L in (10, 100):
    print("list", L, "forward tile rows:", lib.ultr_dnn_forward_tile_rows(shape.desc, cfg["B"] * L, 0))
print(json.dumps(bench.eval_leg(cfg, torch.device("cuda"), lib, p0), indent=1)[:1800])
