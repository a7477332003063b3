"""GPU box: the HOST side of one `algo.train(feed.get_batch(ds)[0])` call of the plugin API with a DeviceClickFeed (config 2):
pure-Python time per call with the C entry points swapped for no-ops (and read_loss not waiting), then a cProfile breakdown."""
import cProfile
import ctypes
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ultra_pytorch_amd.input_layer.device_click_feed import DeviceClickFeed  # noqa: E402
from ultra_pytorch_amd.utils import find_class  # noqa: E402

F, L, B = 136, 10, 256


class DS:
    pass


nq = 2000
rng = np.random.RandomState(99)
ds = DS()
ds.feature_size = F
ds.features = rng.uniform(-1, 1, size=(nq * L, F)).astype(np.float32)
ds.dids = list(range(nq * L))
ds.initial_list = np.arange(nq * L, dtype=np.int64).reshape(nq, L).tolist()
rel = rng.randint(0, 5, size=(nq, L))
rel[:, 0] = np.maximum(rel[:, 0], 1)
ds.labels = rel.tolist()
exp = {"learning_algorithm": "ultra_pytorch_amd.learning_algorithm.IPWrank", "learning_algorithm_hparams": "",
       "ranking_model": "ultra_pytorch_amd.ranking_model.DNN", "ranking_model_hparams": "hidden_layer_sizes=[256,256]",
       "max_candidate_num": L, "selection_bias_cutoff": L, "metrics": ["ndcg"], "metrics_topn": [1, 3, 5, 10]}
algo = find_class(exp["learning_algorithm"])(ds, exp)
feed = DeviceClickFeed(algo, B, "")
out = sys.stdout
sys.stdout = open(os.devnull, "w")
for _ in range(50):
    algo.train(feed.get_batch(ds)[0])
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    algo.train(feed.get_batch(ds)[0])
torch.cuda.synchronize()
real = (time.perf_counter() - t0) / n
# no-op C entries
eng = next(iter(algo._train_engines.values()))
NOOP3 = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(lambda a, b, c: 0)
NOOP2 = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)(lambda a, b: 0)
eng._fn_feed, eng._fn = NOOP3, NOOP2
eng.read_loss = lambda timeout_s=60.0: 0.0
t0 = time.perf_counter()
for _ in range(n):
    algo.train(feed.get_batch(ds)[0])
py = (time.perf_counter() - t0) / n
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    algo.train(feed.get_batch(ds)[0])
pr.disable()
sys.stdout = out
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(json.dumps({"us_per_call_real": 1e6 * real, "us_per_call_python_only": 1e6 * py}))
print(s.getvalue()[:3500])
