#!/usr/bin/env python3
"""bench.py — queries/sec of one training step of the hot path on N MI355X GPUs.

Default workload = BASELINE.json configs[1] (`--config 2`): MSLR-WEB10K-shaped synthetic data — F=136 features,
list_size L=10, B=256 queries per GPU per step, IPWrank + DNN[256,256], PBM clicks, Adagrad lr 0.05, clip 5.0.
`--config {3,4pair,4lambda,5}` times the other BASELINE configs the same way (they are parity-test cases first; their
lines are kept under profiles/, the driver's headline is config 2).
A "step" = model.train on one pre-built batch: gather + forward -> loss -> backward -> clip + optimizer (+ EM).
Batches are resident in HBM before the timed region starts (a pool of pre-staged batches is cycled), parameters /
optimizer state persist across steps, nothing is skipped or cached.
N > 1: one process per GPU, queries shard across ranks (weak scaling: B per GPU fixed), ONE sum per step of
[gradients | loss normalisers] - ultr_comm_allreduce (one kernel, peer reads over xGMI) or, when that path is
unavailable, the RCCL all-reduce - then every rank applies the identical update.

`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks (torch.distributed.run on
127.0.0.1); under torchrun (RANK / WORLD_SIZE set, the driver's form) it is one of the ranks.  Both forms print ONE JSON line.

`value` is the API-faithful figure of SURVEY 8(d): every step is followed by the read of its loss on the host (the
reference's `loss.item()`, here a spin on the update kernel's report in host-mapped memory); the same loop without the
per-step read is `queries_per_sec_no_host_sync`.

Prints ONE JSON line on rank 0 (see the contract in the task statement); extra keys: `roofline` (dominant kernel,
timed inside the timed region from its own dispatch packets), `cpu_baseline` (the oracle = a torch-CPU port of the
reference's step, timed on this box's host cores on the same workload), `torch_rocm_baseline` (the same step in stock
PyTorch-ROCm ops on this GPU: context), `kernel_us` (per-kernel average, calibration pass), `plugin_*` (the same step through
the reference-shaped plugin API), and for N > 1 `dp_exchanges` (BOTH gradient exchanges timed in this run: the one-kernel hipIpc
exchange and the RCCL all-reduce), `dp_exchange` (the one `value` was measured with), `rccl_ranks`, `dp_checks`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CLIP = 5.0
POOL = 16  # pre-staged batches per rank, cycled
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak.  The work of every DNN kernel is counted as
# fp32 multiply-adds and priced against THIS peak also where the products run as three f16 MFMAs on split operands (round 3): the
# results are fp32-accurate, the fraction then says how the kernel compares with a perfect fp32-matrix-core implementation


def h3_products_on():
    return any(os.environ.get(k, "1") != "0" for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3", "ULTR_WG_H3"))
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak
# profiling slots of the library (ultr_prof.h); slot 7 = forward + loss + backward fused in one launch (small batches)
KNAMES = ["dnn_fwd_kernel", "loss_kernel", "dnn_bwd_kernel", "dnn_wgrad_kernel", "grad_reduce_kernel", "update_kernel",
          "ndcg_list_kernel", "dnn_fb_kernel"]
KSLOTS = [0, 1, 2, 3, 4, 5, 7]

CONFIGS = {
    "2": dict(F=136, L=10, B=256, hidden=[256, 256], algo="softmax", lr=0.05, clicks=True, model="dnn", pool=16,
              workload="MSLR-WEB10K synthetic (136-d, list_size=10, batch=256/GPU): IPWrank + DNN[256,256], PBM clicks, "
                       "Adagrad lr 0.05, clip 5.0; one step = forward+loss+backward+clip+update"),
    "3": dict(F=136, L=20, B=512, hidden=[512, 256, 128], algo="dla", lr=0.05, clicks=True, model="dnn", pool=8,
              workload="MSLR-WEB30K synthetic (136-d, list_size=20, batch=512/GPU): DLA + DNN[512,256,128] + DenoisingNet "
                       "propensity, PBM clicks; one step = forward+dual loss+backward+separate clips+stateless Adagrad"),
    "4pair": dict(F=700, L=50, B=256, hidden=[512, 256, 128], algo="pairdebias", lr=0.005, clicks=True, model="dnn", pool=4,
                  workload="Yahoo! Set1 synthetic (700-d, list_size=50, batch=256/GPU): PairDebias + DNN[512,256,128], PBM "
                           "clicks; one step = forward+pairwise debiased loss+backward+clip+Adagrad+EM"),
    "4lambda": dict(F=700, L=50, B=256, hidden=[512, 256, 128], algo="lambdarank", lr=0.05, clicks=False, model="dnn", pool=4,
                    workload="Yahoo! Set1 synthetic (700-d, list_size=50, batch=256/GPU): LambdaRank + DNN[512,256,128], "
                             "relevance labels; one step = forward+lambda loss+backward+clip+Adagrad+EM"),
    "5": dict(F=220, L=100, B=1024, hidden=None, algo="softmax", lr=0.05, clicks=True, model="setrank", pool=2,
              workload="Istella-S synthetic (220-d, list_size=100, batch=1024/GPU): IPWrank + SetRank (d_model 256, 8 heads, "
                       "2 layers, dff 64), PBM clicks; one step = forward+loss+backward+clip+Adagrad"),
}


def dnn_dims(cfg):
    dims, k = [], cfg["F"]
    for m in cfg["hidden"] + [1]:
        dims.append((k, m))
        k = m
    return dims


def step_flops(cfg):
    """ALGORITHMIC flops of one step: 2*S forward + 2*S_hidden weight gradients + 2*(S - F*H1) dgrad per
    document; SetRank: 6 x the forward multiply-adds per token (8f.1)."""
    N = cfg["B"] * cfg["L"]
    if cfg["model"] == "setrank":
        F, L = cfg["F"], cfg["L"]
        mac = F * 64 + 64 * 256 + 2 * (2 * L * 256 + 256 * 256 + 2 * 256 * 64) + 256 * 64 + 64
        return 3 * 2.0 * mac * N
    dims = dnn_dims(cfg)
    s_all = sum(k * m for k, m in dims)
    return N * (6.0 * s_all - 2.0 * dims[0][0] * dims[0][1])


def algorithmic_work(cfg, P):
    """Per-launch ALGORITHMIC work of each DNN kernel slot."""
    N = cfg["B"] * cfg["L"]
    dims = dnn_dims(cfg)
    s_all = sum(k * m for k, m in dims)
    s_hidden = sum(k * m for k, m in dims[:-1])
    dgrad = 2.0 * N * (s_all - dims[0][0] * dims[0][1])
    return {
        0: ("mfma", 2.0 * N * s_all),
        1: ("hbm", 4.0 * 4 * N),  # scores, labels in; dscores out (+ weights)
        2: ("mfma", dgrad),
        3: ("mfma", 2.0 * N * s_hidden),
        4: ("hbm", 4.0 * P),  # the flat gradient written once (slab re-reads are overhead, not algorithmic)
        5: ("hbm", 4.0 * 5 * P),  # read g, read+write Adagrad sum, read+write params
        7: ("mfma", 2.0 * N * s_all + dgrad),  # forward + dgrad in one launch
    }


PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 matrix peak (v_mfma_f32_16x16x32_f16)
L2_STREAM_PEAK_TBS = 8 * 2048 * 2.4e9 / 1e12  # eight XCD-private L2s x 2 KB / clk x 2.4 GHz = 39.3 TB/s (MI355X_MICROARCH.md)


def issued_matrix_work(cfg, slot):
    """What the matrix cores actually ISSUE for kernel slot `slot` under the current knobs (DnnPlan::h3f / h3b rules, ultr_dnn.hip
    ultr_make_dnn_plan): flops on the f16 pipe (three v_mfma_f32_16x16x32_f16 per algorithmic product: x 3) and flops left on the
    fp32 pipe.  Returns (f16_issued_flops, f32_flops)."""
    N = cfg["B"] * cfg["L"]
    dims = dnn_dims(cfg)  # [(K_j, M_j)], last = the scorer (a row dot product: no matrix-core work worth counting)
    nl = len(dims)
    on = lambda k: os.environ.get(k, "1") != "0"
    knob_f = on("ULTR_FB_H3") if slot == 7 else on("ULTR_FWD_H3")
    knob_b = on("ULTR_FB_H3") if slot == 7 else on("ULTR_BWD_H3")
    fused_all = all(m >= 256 and m % 32 == 0 for _, m in dims[:-1])  # the fused kernel takes the split-half build only when every layer has its copies
    f16 = f32 = 0.0
    for j, (k, m) in enumerate(dims):
        fl = 2.0 * N * k * m
        if j < nl - 1 and slot in (0, 7):  # forward product of layer j
            h3 = knob_f and m >= 256 and (fused_all if slot == 7 else True)
            f16, f32 = (f16 + 3 * fl, f32) if h3 else (f16, f32 + fl)
        elif j == nl - 1 and slot in (0, 7):
            f32 += fl
        if j >= 1 and slot in (2, 7):      # dgrad product du_j = dz_j . W_j (none for layer 0: the layer-0 shortcut)
            h3 = knob_b and j < nl - 1 and k >= 256 and k % 32 == 0 and (fused_all if slot == 7 else True)
            f16, f32 = (f16 + 3 * fl, f32) if h3 else (f16, f32 + fl)
    if slot == 3:
        fl = 2.0 * N * sum(k * m for k, m in dims[:-1])
        # weight gradients: the split-half launch (dnn_wgrad_h3_kernel) from ULTR_WG_H3_MIN_ROWS rows when every layer allows 16-byte paths
        h3 = os.environ.get("ULTR_WG_H3", "1") != "0" and all(k % 4 == 0 and m % 4 == 0 for k, m in dims[:-1]) and \
            (os.environ.get("ULTR_WG_H3", "1") == "2" or N >= int(os.environ.get("ULTR_WG_H3_MIN_ROWS", "4096")))
        f16, f32 = (3 * fl, 0.0) if h3 else (0.0, fl)
    return f16, f32


def make_pool(cfg, rng, device):
    from ultra_pytorch_amd import synthetic
    pool = []
    for _ in range(cfg["pool"]):
        feats, docids, y = synthetic.make_batch(rng, cfg["B"], cfg["L"], cfg["F"], clicks=cfg["clicks"])
        pool.append((torch.from_numpy(feats).to(device), feats.shape[0], torch.from_numpy(docids).to(device),
                     torch.from_numpy(y).to(device), (feats, docids, y)))
    return pool


def oracle_stepper(cfg, structure="vectorised"):
    """(state0, step) for the CPU baseline: one training step of the oracle on host arrays."""
    from oracle import ultr_oracle as O
    from ultra_pytorch_amd import synthetic
    F, L, hidden, algo, lr = cfg["F"], cfg["L"], cfg["hidden"], cfg["algo"], cfg["lr"]
    ipw = synthetic.load_ipw()
    if cfg["model"] == "setrank":
        sc = (F, 256, 8, 2, 64)

        def step(st, batch):
            feats, ids, y = batch
            r = O.train_step_setrank_softmax(st["p"], st["s"], sc, feats, ids, y, ipw_list=ipw, lr=lr, max_norm=CLIP)
            return dict(p=r["params"], s=r["state"])
        return step
    if algo == "softmax":
        def step(st, batch):
            feats, ids, y = batch
            r = O.train_step_softmax(st["p"], st["s"], F, hidden, feats, ids, y, ipw_list=ipw, lr=lr, max_norm=CLIP)
            return dict(p=r["params"], s=r["state"])
    elif algo == "dla":
        def step(st, batch):
            feats, ids, y = batch
            r = O.dla_step(st["p"], st["aux"], F, hidden, feats, ids, y, lr=lr, max_norm=CLIP,
                           fresh_optimizers=(structure == "reference"))
            return dict(p=r["params"], s=st["s"], aux=r["prop_params"])
    else:
        fn = O.pairdebias_step if algo == "pairdebias" else O.lambdarank_step
        kw = dict(loops=True) if (structure == "reference" and algo == "pairdebias") else {}

        def step(st, batch):
            feats, ids, y = batch
            r = fn(st["p"], st["s"], st["aux"][:L], st["aux"][L:], F, hidden, feats, ids, y, lr=lr, max_norm=CLIP, **kw)
            return dict(p=r["params"], s=r["state"], aux=np.concatenate([r["t_plus"].ravel(), r["t_minus"].ravel()]))
    return step


def cpu_state0(cfg, params0):
    L = cfg["L"]
    aux = None
    if cfg["algo"] == "dla":
        aux = np.zeros(L + 1, np.float32)
    elif cfg["algo"] in ("pairdebias", "lambdarank"):
        aux = np.ones(2 * L, np.float32)
    return dict(p=params0.copy(), s=np.zeros_like(params0), aux=aux)


def _numa_node0_cpus():
    """CPUs of NUMA node 0 this process may use, ONE hardware thread per physical core first (SMT siblings at the end): a
    thread count up to the core count then never doubles up on a core."""
    try:
        txt = open("/sys/devices/system/node/node0/cpulist").read().strip()
        cpus = []
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed]
    except Exception:
        cpus = sorted(os.sched_getaffinity(0))
    first, rest, seen = [], [], set()
    for c in cpus:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except Exception:
            sib = str(c)
        (rest if sib in seen else first).append(c)
        seen.add(sib)
    return first + rest


def _pin_all_threads(cpus):
    """Pin every thread of this process (torch's intra-op pool included) to `cpus`; returns {tid: old mask}."""
    old = {}
    for t in os.listdir("/proc/self/task"):
        try:
            old[int(t)] = os.sched_getaffinity(int(t))
            os.sched_setaffinity(int(t), cpus)
        except Exception:
            pass
    return old


def _unpin(old):
    for t, m in old.items():
        try:
            os.sched_setaffinity(t, m)
        except Exception:
            pass


def cpu_baseline(cfg, pool, params0, budget_s=10.0):
    """The oracle's step (vectorised torch-CPU port of the reference's train()) on this box's host cores.
    Small GEMMs do not scale to every core of a big host and an oversubscribed baseline would flatter the GPU, so the thread
    count is CHOSEN BY MEASUREMENT - and by the same kind of measurement that is reported: for every candidate the threads are
    first pinned to that many distinct physical cores of ONE NUMA node, warmed up, then timed over three chunks of steps (the
    median chunk = the candidate's sustained rate, `threads_candidates_sustained`); a larger count has to win by 20 %, and a
    choice whose long run does not sustain its figure hands over to the next smaller count (`long_runs`).  (Round 3 picked by a 12-step probe: 32 threads at 3.5 ms/step, which then sustained 8.2 ms on the driver's box.)
    `value` = the MEDIAN over chunks of the long run at that count; `sustained_over_probe` compares it with the candidate's own
    figure and anything beyond 1.3x is flagged.  `cores` = threads actually used."""
    ncpu = os.cpu_count() or 1
    B = cfg["B"]
    step = oracle_stepper(cfg)
    npool = len(pool)
    node0 = _numa_node0_cpus()

    def run(nsteps, st, i0):
        t = 0.0
        for i in range(i0, i0 + nsteps):
            t0 = time.perf_counter()
            st = step(st, pool[i % npool][4])
            t += time.perf_counter() - t0
        return t, st

    heavy = cfg["model"] == "setrank" or cfg["F"] * cfg["L"] * B > 2e6
    best, probe = None, {}
    old_masks = {}
    try:
        for th in sorted({1, 4, 8, 16, 32, 64} if not heavy else {8, 32, 64}):
            if th > len(node0):
                continue
            torch.set_num_threads(th)
            st = cpu_state0(cfg, params0)
            _, st = run(1, st, 0)  # spins the pool up: its threads exist now ...
            old = _pin_all_threads(node0[:th])  # ... and are pinned BEFORE anything is timed
            old_masks = old_masks or old
            t1, st = run(1, st, 1)  # its duration sizes the chunks (~0.5 s each, >= 3 steps)
            per = max(1, min(50, int(0.5 / max(t1, 1e-4)))) if not heavy else 1
            per = max(per, 1 if heavy else 3)
            _, st = run(per, st, 2)  # one untimed chunk on the pinned cores: clocks, caches and the allocator settle
            ts, i0 = [], 2 + per
            for r in range(3):
                t, st = run(per, st, i0)
                i0 += per
                ts.append(t / per)
            probe[th] = float(np.median(ts))
            if best is None or probe[th] < 0.80 * probe[best]:  # ascending: a larger count must win by more than 20 %
                best = th

        def long_run(th):
            torch.set_num_threads(th)
            st = cpu_state0(cfg, params0)
            _, st = run(1, st, 0)
            _pin_all_threads(node0[:th])
            _, st = run(1 if heavy else 4, st, 1)
            n, t_used, chunks = 0, 0.0, []
            chunk = 1 if heavy else 10
            while t_used < budget_s or n < (2 if heavy else 20):
                t, st = run(chunk, st, 5 + n)
                chunks.append(t / chunk)
                n += chunk
                t_used += t
            return n, t_used, chunks

        # the long run at the chosen count; if it does not sustain what the candidate measurement promised (round 4, one box: 32
        # threads 4.0 ms as a candidate, 9.4 ms over 770 steps) the next smaller counts get the same long run - the first stable
        # one is reported, every attempt is listed
        attempts = []
        order = [best] + sorted((t for t in probe if t < best), key=lambda t: probe[t])[:2]
        for th in order:
            n, t_used, chunks = long_run(th)
            ratio = float(np.median(chunks)) / probe[th]
            attempts.append({"threads": th, "ms_per_step_median": round(1e3 * float(np.median(chunks)), 3), "sustained_over_probe": round(ratio, 3)})
            if 1 / 1.3 <= ratio <= 1.3:
                best = th
                break
        else:
            th = min(attempts, key=lambda a: a["ms_per_step_median"])["threads"]
            if th != order[-1]:
                n, t_used, chunks = long_run(th)
            best = th
    finally:
        _unpin(old_masks)
    med, mean = float(np.median(chunks)), t_used / n
    ratio = med / probe[best]
    if not (1 / 1.3 <= ratio <= 1.3):
        print("WARNING: cpu_baseline is UNSTABLE on this host: %d threads sustained %.2f ms/step when chosen vs %.2f ms/step in the long run (x%.2f)"
              % (best, 1e3 * probe[best], 1e3 * med, ratio), file=sys.stderr)
    out = {"value": B / med, "unit": "queries/sec", "cores": best, "kind": "port",
           "value_mean": B / mean, "ms_per_step_median": 1e3 * med, "ms_per_step_mean": 1e3 * mean,
           "threads_candidates_sustained": {str(k): round(1e3 * v, 3) for k, v in probe.items()},
           "threads_candidates_unit": "ms/step, median of 3 chunks, threads pinned to distinct physical cores of NUMA node 0 before timing",
           "sustained_over_probe": ratio, "stable": bool(1 / 1.3 <= ratio <= 1.3), "long_runs": attempts,
           "sample": "%d steps (median of %d chunks) of the same workload after warm-up, oracle/ultr_oracle (vectorised torch-CPU "
                     "restatement), %d threads (a larger count has to beat a smaller one by 20 %%; the first count whose long run sustains its candidate figure) pinned to NUMA node 0 (%d of the host's %d CPUs)"
                     % (n, len(chunks), best, len(node0), ncpu)}
    if cfg["algo"] in ("dla", "pairdebias"):
        # the reference's own structure (per-step optimizer construction for DLA, the 2-level Python pair loop
        # for PairDebias) next to the vectorised port, so that the ratio is not quoted against an artificially slow CPU
        step_r = oracle_stepper(cfg, structure="reference")
        st = cpu_state0(cfg, params0)
        n2, t2 = 0, 0.0
        while t2 < budget_s / 2 or n2 < 2:
            t0 = time.perf_counter()
            st = step_r(st, pool[n2 % npool][4])
            t2 += time.perf_counter() - t0
            n2 += 1
        out["reference_structure"] = {"value": B * n2 / t2, "unit": "queries/sec", "cores": best, "ms_per_step": 1e3 * t2 / n2,
                                      "what": "per-step torch.optim.Adagrad construction (dla.py:153-154)" if cfg["algo"] == "dla"
                                      else "2-level Python pair loop (pairwise_debias.py:142-157)"}
    return out


class _DataSet:
    def __init__(self, feature_size):
        self.feature_size = feature_size


def plugin_rates(cfg, pool, device, steps):
    """The same step through the reference-shaped plugin API (what main.py sees): IPWrank.train(input_feed) with
    (a) the host-numpy feed the reference's ClickSimulationFeed emits (f64 features, f32 ids: numpy marshal + PCIe +
    loss.item() + the per-step print) and (b) input_layer.DeviceClickFeed (dataset resident in HBM, clicks on device)."""
    from ultra_pytorch_amd.utils import find_class
    F, L, B = cfg["F"], cfg["L"], cfg["B"]
    exp = {"learning_algorithm": "ultra_pytorch_amd.learning_algorithm.IPWrank", "learning_algorithm_hparams": "",
           "ranking_model": "ultra_pytorch_amd.ranking_model.DNN",
           "ranking_model_hparams": "hidden_layer_sizes=%s" % json.dumps(cfg["hidden"]),
           "max_candidate_num": L, "selection_bias_cutoff": L, "metrics": ["ndcg"], "metrics_topn": [1, 3, 5, 10]}
    algo = find_class(exp["learning_algorithm"])(_DataSet(F), exp)
    feeds = []
    for _, _, _, _, (feats, ids, y) in pool[:8]:
        feed = {algo.letor_features_name: feats.astype(np.float64)}
        for l in range(L):
            feed[algo.docid_inputs_name[l]] = ids[l].astype(np.float32)
            feed[algo.labels_name[l]] = y[l].astype(np.float32)
        feeds.append(feed)
    devnull = open(os.devnull, "w")
    saved_out = sys.stdout
    sys.stdout = devnull  # the per-step " Loss ..." print is part of the API; keep it off the terminal, not off the clock
    try:
        for i in range(20):
            algo.train(dict(feeds[i % len(feeds)]))
        n = max(100, min(steps, 300))  # secondary figure: its own loop length (the driver runs --steps 20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            algo.train(dict(feeds[i % len(feeds)]))
        torch.cuda.synchronize()
        host_rate = B * n / (time.perf_counter() - t0)
        # device feed
        from ultra_pytorch_amd.input_layer.device_click_feed import DeviceClickFeed
        nq = 20000
        rng = np.random.RandomState(99)

        class DS:
            pass
        ds = DS()
        ds.features = rng.uniform(-1, 1, size=(nq * L, F)).astype(np.float32)
        ds.dids = list(range(nq * L))
        ds.initial_list = np.arange(nq * L, dtype=np.int64).reshape(nq, L).tolist()
        rel = rng.randint(0, 5, size=(nq, L))
        rel[:, 0] = np.maximum(rel[:, 0], 1)
        ds.labels = rel.tolist()
        feed_obj = DeviceClickFeed(algo, B, "")
        for i in range(50):
            algo.train(feed_obj.get_batch(ds)[0])
        n = 500  # 25 ms: its own loop length whatever --steps says (100 calls measured 4.99 - 5.10 M on boxes where 500 measure 5.2 M)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            algo.train(feed_obj.get_batch(ds)[0])
        torch.cuda.synchronize()
        dev_rate = B * n / (time.perf_counter() - t0)
    finally:
        sys.stdout = saved_out
        devnull.close()
    return host_rate, dev_rate


def torch_rocm_baseline(cfg, pool, params0, device, budget_s=4.0):
    """The same step in stock PyTorch-ROCm ops on this GPU (tools/torch_rocm_baseline.py): what the reference does with its
    tensors on 'cuda'.  Context for the hand-written path, never the target; loss.item() each step, as the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch_rocm_baseline as TB
    from ultra_pytorch_amd import synthetic
    st = TB.Stepper(cfg, params0, synthetic.load_ipw() if cfg["algo"] == "softmax" else None, device, cfg["lr"], CLIP)
    staged = [st.stage(b[4]) for b in pool]
    for i in range(5):
        st.step(staged[i % len(staged)])
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s or n < 10:
        st.step(staged[n % len(staged)])
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"value": cfg["B"] / dt, "unit": "queries/sec", "ms_per_step": 1e3 * dt, "steps": n,
            "what": "stock torch ops on this GPU (nn.LayerNorm / nn.Linear / autograd / clip_grad_norm_ / torch.optim.Adagrad, "
                    "fp32, loss.item() each step): the reference's step with device='cuda' - context, not the target"}


class Workload:
    """Everything one BASELINE config needs on one rank: model shape, replicated parameters / optimizer state, a pool of
    pre-staged batches resident in HBM, the step engine."""

    def __init__(self, key, device, rank=0, pg=None, attention_dtype="fp32"):
        from ultra_pytorch_amd import engine, hip_ops, synthetic
        cfg = self.cfg = CONFIGS[key]
        F, L, B = cfg["F"], cfg["L"], cfg["B"]
        self.key, self.device, self.attention_dtype = key, device, attention_dtype
        if cfg["model"] == "setrank":
            from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
            self.shape = hip_ops.SetRankShape(F, 256, 8, 2, 64, attention_dtype=attention_dtype)
            self.params0 = init_setrank_params(self.shape, seed=0).numpy()
            self.eng_cls = engine.SetRankStepEngine
        else:
            from ultra_pytorch_amd.ranking_model import init_flat_params
            self.shape = hip_ops.DnnShape(F, cfg["hidden"], "elu")
            self.params0 = init_flat_params(self.shape, seed=0).numpy()
            self.eng_cls = engine.StepEngine
        self.P = self.shape.n_params
        self.params = torch.from_numpy(self.params0.copy()).to(device)  # identical replicas on every rank
        self.state = None if cfg["algo"] == "dla" else torch.zeros_like(self.params)
        self.aux = None
        if cfg["algo"] == "dla":
            self.aux = torch.zeros(L + 1, device=device)
        elif cfg["algo"] in ("pairdebias", "lambdarank"):
            self.aux = torch.ones(2 * L, device=device)
        self.ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=device) if cfg["algo"] == "softmax" else None
        self.pool = make_pool(cfg, np.random.RandomState(1234 + rank), device)
        self.pg = pg
        self.eng = self.make_engine()

    def make_engine(self, **kw):
        cfg = self.cfg
        return self.eng_cls(self.shape, cfg["B"], cfg["L"], self.device, algo=cfg["algo"], learning_rate=cfg["lr"],
                            max_gradient_norm=CLIP, process_group=self.pg, **kw)

    def step(self, i, e=None):
        f, nd, ids, y, _ = self.pool[i % len(self.pool)]
        return (e or self.eng).train_step(self.params, self.state, f, nd, ids, y, aux=self.aux, ipw_table=self.ipw)


def dtype_label(cfg, attention_dtype="fp32"):
    """The arithmetic the step ISSUES under the current knobs (values, accumulation and every row-wise phase are f32 everywhere)."""
    if cfg["model"] == "dnn":
        return (("f32 (products of layers with >= 256 outputs: three f16 MFMAs on split hi/lo f16 operands, f32 accumulate - "
                 "f32-accurate, same 1e-5 parity bar; ULTR_FB_H3 / ULTR_FWD_H3 / ULTR_BWD_H3 = 0 for f32 MFMAs)") if h3_products_on()
                else "f32")
    on = lambda k: os.environ.get(k, "1") != "0"
    parts = []
    if on("ULTR_SR_H3"):
        parts.append("Linear forward / dgrad GEMMs")
    if on("ULTR_SR_WG_H3"):
        parts.append("d x d weight gradients")
    if attention_dtype == "fp32" and on("ULTR_SR_ATTN_H3"):
        parts.append("attention backward")
    s = "f32"
    if parts:
        s += (" (%s: three f16 MFMAs on split hi/lo f16 operands, f32 accumulate - f32-accurate, 1e-5 parity bar; attention forward, "
              "LayerNorms, softmax, loss: f32; ULTR_SR_H3 / ULTR_SR_WG_H3 / ULTR_SR_ATTN_H3 = 0 for f32 MFMAs)" % ", ".join(parts))
    if attention_dtype != "fp32":
        s += " (self-attention operands plain f16, f32 accumulate: ordering-level parity)"
    return s


SHORT_RUN_SPINUP_MS = 150.0  # untimed steps in front of every short run of another config (disclosed in its record)
FP32_KNOBS = ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3", "ULTR_WG_H3", "ULTR_SR_H3", "ULTR_SR_ATTN_H3", "ULTR_SR_WG_H3")


def short_config_run(key, device, lib, steps, warmup, fp32=False):
    """One of the OTHER BASELINE configs on the driver's line: a short synced run - `steps` steps, each
    followed by the host's read of its loss - after `warmup` steps and (DNN) a calibration pass with every kernel timer armed.
    fp32=True: the same run with EVERY product on v_mfma_f32_16x16x4_f32 (the split-half knobs off: the reference's arithmetic,
    DNN.py:43-56) - VERDICT r05 item 3: the strict-fp32 figure of every config on the driver's line."""
    from ultra_pytorch_amd import _lib
    if fp32:
        keep = {k: os.environ.get(k) for k in FP32_KNOBS}
        try:
            for k in keep:
                os.environ[k] = "0"
            rec = short_config_run(key, device, lib, steps, warmup)
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            lib.ultr_config_reload()
        out = {k: rec[k] for k in ("ms_per_step", "queries_per_sec", "step_frac_of_fp32_mfma_peak", "dominant_kernel", "final_loss", "wall_s") if k in rec}
        out["what"] = " ".join("%s=0" % k for k in FP32_KNOBS) + ": every product on v_mfma_f32_16x16x4_f32"
        return out
    t_setup = time.perf_counter()
    W = Workload(key, device)
    cfg = W.cfg
    # the same treatment as the headline's timed region (`spinup_ms` there): building the workload leaves the GPU idle for a second and
    # the shader clock needs tens of milliseconds of load to settle - untimed steps first (a training run lasts minutes, not 8 ms;
    # without them these short runs read 3 - 5 % above the long runs of profiles/rNN_cfg*_bench.json)
    t_sp, spun = time.perf_counter(), 0
    while 1e3 * (time.perf_counter() - t_sp) < SHORT_RUN_SPINUP_MS:
        for i in range(8):
            W.step(spun + i)
        W.eng.read_loss()
        spun += 8
    for i in range(warmup):
        W.step(i)
    torch.cuda.synchronize()
    rec = {"workload": cfg["workload"], "spinup_ms": SHORT_RUN_SPINUP_MS, "spinup_steps": spun, "rewarm_ms_after_calibration": 30.0 if cfg["model"] == "dnn" else 0.0}
    dnn = cfg["model"] == "dnn"
    if dnn:
        tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
        ncal = 6
        _lib.check(lib.ultr_prof_enable(0xBF, 8 * ncal), "ultr_prof_enable")
        for i in range(ncal):
            W.step(i)
        torch.cuda.synchronize()
        _lib.check(lib.ultr_prof_collect(tot, cnt), "ultr_prof_collect")
        lib.ultr_prof_enable(0, 0)
        kus = {k: 1e3 * tot[k] / cnt[k] for k in KSLOTS if cnt[k] > 0}
        # reading back the event pairs leaves the GPU idle for a millisecond or two (clocks drop): untimed steps again (30 ms)
        t_sp = time.perf_counter()
        while 1e3 * (time.perf_counter() - t_sp) < 30.0:
            for i in range(8):
                W.step(i)
            W.eng.read_loss()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        W.step(i)
        W.eng.read_loss()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = W.eng.read_loss()
    flops = step_flops(cfg)
    rec.update({"ms_per_step": 1e3 * dt, "queries_per_sec": cfg["B"] / dt, "steps": steps, "warmup": warmup,
                "step_tflops": flops / dt / 1e12, "step_frac_of_fp32_mfma_peak": flops / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                "dtype": dtype_label(cfg), "final_loss": loss, "finite": bool(np.isfinite(loss))})
    if dnn:
        dom = max(kus, key=kus.get)
        bound, amount = algorithmic_work(cfg, W.P)[dom]
        ach = amount / (1e-6 * kus[dom]) / (1e12 if bound == "mfma" else 1e9)
        rec["dominant_kernel"] = {"kernel": KNAMES[dom], "avg_launch_us": kus[dom], "bound": bound, "achieved": ach,
                                  "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                  "frac": ach / (PEAK_FP32_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBS),
                                  "source": "calibration pass of %d steps in front of the timed steps, every kernel timer armed" % ncal}
        if bound == "mfma":
            # what the kernel ISSUES against the peak of the pipe it issues on (`frac` prices algorithmic fp32 flops against the fp32 peak)
            f16_issued, f32_issued = issued_matrix_work(cfg, dom)
            pipe_s = f16_issued / (PEAK_F16_MFMA_TFLOPS * 1e12) + f32_issued / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            rec["dominant_kernel"]["frac_of_issued_dtype_peak"] = pipe_s / (1e-6 * kus[dom])
            rec["dominant_kernel"]["mfma_dtype"] = ("f16 x3 (split hi/lo)" + ("" if f32_issued == 0 else " + f32")) if f16_issued > 0 else "f32"
        rec["kernel_us"] = {KNAMES[k]: round(v, 2) for k, v in kus.items()}
        rec["stage_kernels"] = stage_kernels(lib, cfg)
    else:
        rec["dominant_kernel"] = {"kernel": "whole step (21 launches, none dominant: profiles/r06_cfg5_pmc.md)", "bound": "mfma",
                                  "achieved": rec["step_tflops"], "unit": "TFLOP/s", "frac": rec["step_frac_of_fp32_mfma_peak"]}
    W.eng.close()
    del W
    torch.cuda.empty_cache()
    rec["wall_s"] = round(time.perf_counter() - t_setup, 2)
    return rec


def stage_kernels(lib, cfg):
    """Which kernels the forward / backward STAGES of this config's training step launch (ultr_dnn_forward_tile_rows /
    ultr_dnn_backward_tile_rows: host-only planners of the library) - the kernel_us slots are named after the stage."""
    from ultra_pytorch_amd import hip_ops
    if cfg["model"] != "dnn":
        return None
    shape = hip_ops.DnnShape(cfg["F"], cfg["hidden"], "elu")
    n = cfg["B"] * cfg["L"]

    def name(code, wide, tile, per_layer):
        if code >= 1000:
            return "%s<%d> (%d rows per workgroup, %d workgroups)" % (wide, (code - 1000 + 15) // 16, code - 1000, -(-n // (code - 1000)))
        if code == 0:
            return per_layer
        return "%s (%d rows per workgroup)" % (tile, code)
    fwd = lib.ultr_dnn_forward_tile_rows(shape.desc, n, 1)
    bwd = lib.ultr_dnn_backward_tile_rows(shape.desc, n)
    fused = n <= 16 * 256 and cfg["algo"] == "softmax" and os.environ.get("ULTR_NO_FUSED_FB", "0") != "1"
    if fused:
        return {"forward+loss+backward": "dnn_fb_kernel (16 rows per workgroup)"}
    return {"forward": name(fwd, "dnn_fwdw_kernel", "dnn_fwd_kernel", "per-layer launches (ultr_dnn_big.hip)"),
            "backward": name(bwd, "dnn_bwdw_kernel", "dnn_bwd2_kernel", "per-layer launches (ultr_dnn_big.hip)")}


def eval_leg(cfg, device, lib, params0):
    """validation() on the device (SURVEY 8 a12 / a13, 8d "Validation-only: 2 S per document"): DNN forward at the evaluated list
    length + pad mask + NDCG@{1,3,5,10}, the metric vector read on the host after every batch (the reference's .item() per
    metric, ipw_rank.py:204-210).  Two shapes: config 2's own (list_size 10) and max_candidate_num = 100 with ragged lists."""
    from ultra_pytorch_amd import _lib, engine, hip_ops, synthetic
    F, B = cfg["F"], cfg["B"]
    shape = hip_ops.DnnShape(F, cfg["hidden"], "elu")
    params = torch.from_numpy(params0.copy()).to(device)
    dims = dnn_dims(cfg)
    s_all = sum(k * m for k, m in dims)
    out = {}
    for L in (cfg["L"], 100):
        rng = np.random.RandomState(77 + L)
        ev = engine.EvalEngine(shape, B, L, device)
        batches = []
        for _ in range(4):
            feats, ids, y = synthetic.make_batch(rng, B, L, F, clicks=False)
            if L > cfg["L"]:  # ragged lists: a random tail of every list is PAD (id == n_docs), as data_utils.pad leaves them
                n_docs = feats.shape[0]
                lens = rng.randint(L // 2, L + 1, size=B)
                padm = np.arange(L)[:, None] >= lens[None, :]
                ids = np.where(padm, n_docs, ids).astype(np.int32)
                y = np.where(padm, 0.0, y).astype(np.float32)
            batches.append((torch.from_numpy(feats).to(device), feats.shape[0], torch.from_numpy(ids).to(device), torch.from_numpy(y).to(device)))
        live = float(np.mean([(b[2] < b[1]).sum().item() for b in batches]))

        def run(i):
            f, nd, ids, y = batches[i % len(batches)]
            return ev.run(params, f, nd, ids, y)
        for i in range(10):
            run(i)
        torch.cuda.synchronize()
        tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
        ncal = 20
        _lib.check(lib.ultr_prof_enable((1 << 0) | (1 << 6), 4 * ncal), "ultr_prof_enable")
        for i in range(ncal):
            run(i)
        torch.cuda.synchronize()
        _lib.check(lib.ultr_prof_collect(tot, cnt), "ultr_prof_collect")
        lib.ultr_prof_enable(0, 0)
        assert cnt[0] > 0 and cnt[6] > 0, "the eval kernels were not timed"
        fwd_us = 1e3 * tot[0] / cnt[0]
        ndcg_us = 1e3 * tot[6] / cnt[6]
        t_sp = time.perf_counter()  # (the event read-back left the GPU idle: 30 ms of untimed batches, as in short_config_run)
        while 1e3 * (time.perf_counter() - t_sp) < 30.0:
            for i in range(8):
                run(i)
            ev.read_ndcg()
        n = 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            run(i)
            nd_host = ev.read_ndcg()  # what validation() does: the launch's report in host-mapped memory
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for i in range(n):
            run(i)
        torch.cuda.synchronize()
        dt_ns = (time.perf_counter() - t0) / n
        N = B * L
        ndcg_bytes = 12.0 * N  # scores + labels + doc ids in
        fwd_flops = 2.0 * N * s_all
        out["list_size_%d" % L] = {
            "batch": B, "list_size": L, "live_documents_per_batch": live,
            "queries_per_sec": B / dt, "ms_per_batch": 1e3 * dt, "queries_per_sec_no_host_sync": B / dt_ns,
            "forward": {"kernel": "dnn_fwd_kernel", "avg_launch_us": fwd_us, "bound": "mfma", "achieved": fwd_flops / (1e-6 * fwd_us) / 1e12,
                        "unit": "TFLOP/s", "frac": fwd_flops / (1e-6 * fwd_us) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                        "algorithmic_per_launch": fwd_flops},
            "ndcg": {"kernel": "ndcg_list_kernel", "avg_launch_us": ndcg_us, "bound": "hbm", "achieved": ndcg_bytes / (1e-6 * ndcg_us) / 1e9,
                     "unit": "GB/s", "frac": ndcg_bytes / (1e-6 * ndcg_us) / 1e9 / PEAK_HBM_GBS, "algorithmic_per_launch": ndcg_bytes,
                     "limited_by": "a chain of dependent round trips (labels, write-through per-list values, arrival counter, read-back, host report), "
                                   "not bytes: %d KB per launch; fused into the forward's launch it measured SLOWER (24.8 us against 15.05 + 8.58)" % int(ndcg_bytes / 1024)},
            "ndcg_at_1_3_5_10": [float(v) for v in nd_host.tolist()],
            "what": "EvalEngine.run = ultr_dnn_forward_ndcg (one host call): the DNN forward, then ndcg_list_kernel (pad mask, label validation, rank "
                    "sort, DCG / IDCG at the cut-offs, batch means by the last wave), the ndcg vector read on the host after every batch from the "
                    "launch's host-mapped report (no stream synchronisation)",
        }
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: launch N ranks of this script (one per GPU) and pass rank 0's line on."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("ULTR_BENCH_SHARE_GPU", "0") != "1":
        sys.exit("bench.py --gpus %d needs %d GPUs on this node, found %d" % (args.gpus, args.gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS), help="BASELINE.json config (default 2 = the headline)")
    ap.add_argument("--attention-dtype", default="fp32", choices=["fp32", "fp16"],
                    help="config 5 only: operand type of SetRank's self-attention (fp16 = BASELINE config 5's fp16 MFMA attention, "
                         "ordering-level parity; fp32 = the 1e-5 parity path, default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (plugin API, device feed)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of configs 3 / 4pair / 4lambda / 5 and the eval leg")
    ap.add_argument("--no-dp-selftest", action="store_true", help="--gpus N > 1: skip the exchange self-test (delayed publishes) in front of the timed region")
    ap.add_argument("--dp-selftest-steps", type=int, default=1000)
    ap.add_argument("--spinup-ms", type=float, default=None,
                    help="untimed steps for this many ms in front of the W warm-up steps (shader clocks settle; default 300 for config 2)")
    ap.add_argument("--timer-stride", type=int, default=None, help="the dominant kernel is timed on every stride-th step of the timed region")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # does not return
    cfg = CONFIGS[args.config]
    F, L, B, HIDDEN, LR = cfg["F"], cfg["L"], cfg["B"], cfg["hidden"], cfg["lr"]
    light = args.config == "2"
    if args.steps is None:
        args.steps = 2000 if light else (200 if cfg["model"] == "dnn" else 20)
    if args.warmup is None:
        args.warmup = 200 if light else (20 if cfg["model"] == "dnn" else 3)

    # stdout must carry exactly ONE line (the JSON): RCCL / HIP print banners to fd 1 on init, so everything goes
    # to stderr until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    share = os.environ.get("ULTR_BENCH_SHARE_GPU", "0") == "1"  # TEST mode: every rank on cuda:0 over gloo - exercises this
    if share:                                                      # file's N > 1 logic on a one-GPU box (not a measurement)
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py --gpus %d needs %d GPUs on this node, found %d" % (args.gpus, args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from ultra_pytorch_amd import parallel
    _, _, _, pg = parallel.init_process_group_from_env(backend="gloo" if share else "nccl")

    from ultra_pytorch_amd import _lib, engine, hip_ops, synthetic
    lib = _lib.load()
    W = Workload(args.config, device, rank=rank, pg=pg, attention_dtype=args.attention_dtype)
    shape, params0, eng_cls, P = W.shape, W.params0, W.eng_cls, W.P
    params, state, aux, ipw, pool = W.params, W.state, W.aux, W.ipw, W.pool
    npool = len(pool)
    eng = W.eng
    engs = {"main": eng}

    def step(i, e=None):
        f, nd, ids, y, _ = pool[i % npool]
        return (e or eng).train_step(params, state, f, nd, ids, y, aux=aux, ipw_table=ipw)

    _flag = torch.zeros(1, device=device)

    def barrier():
        # all ranks + the device: a one-element all-reduce ENQUEUED behind the queued steps (it completes only when every rank
        # has reached it), then one host synchronisation - torch.distributed.barrier() would cost a second host round trip
        # inside the timed region (forced data-parallel run at 20 steps: 62.5 -> 60.9 us/step)
        if pg is not None:
            torch.distributed.all_reduce(_flag, group=pg)
        torch.cuda.synchronize()

    # ---- spin-up: the shader clock needs tens of milliseconds of load to reach its sustained level; the driver's form
    # (--warmup 5 --steps 20) is 1.3 ms of work in all, so without this a short timed region measures the clock ramp
    # (round 5, same box: 51.5 us/step at --steps 20 against 48.3 at --steps 2000).  Untimed, in front of the W warm-up steps,
    # reported on the line as `spinup_ms`.
    spinup_ms = (300.0 if light else 0.0) if args.spinup_ms is None else args.spinup_ms
    spun = 0
    if spinup_ms > 0:
        t_sp = time.perf_counter()
        while 1e3 * (time.perf_counter() - t_sp) < spinup_ms:
            for i in range(32):
                step(spun + i)
            eng.read_loss()
            spun += 32
        barrier()
    # ---- warm-up (untimed) + pick the dominant kernel with all timers armed ------------------------
    for i in range(args.warmup):
        step(i)
    barrier()
    tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
    dnn = cfg["model"] == "dnn"
    dom = None
    # Timers inside the timed region: ONLY the dominant kernel, on every stride-th step, by the start / stop timestamps of its
    # own dispatch packet (what rocprofv3 --kernel-trace reports).  A timed launch cost ~5 us of stream time with default events (tools/
    # short_region.py: all kernels timed at stride 8 cost 2.3 us per step of the headline, 20 us per timed step; round 5: events
    # without a system-scope release and no shadow samples - five timed launches now cost ~0.3 us per step of a 20-step run), so the other
    # kernels' averages come from the calibration pass in front of the timed region (`kernel_us`, all timers armed; the
    # dominant kernel's figure there and inside the region agree to 1 %).
    # at least five timed launches inside the timed region whatever its length (two were thin)
    stride = args.timer_stride or max(1, min((8 if args.steps < 256 else 32) if light else 4, args.steps // 5))
    cal_us = [0.0] * 8
    cal_cnt = [0] * 8
    if dnn:
        ncal = 50 if light else 10
        _lib.check(lib.ultr_prof_enable(0xBF, 8 * ncal), "ultr_prof_enable")
        for i in range(ncal):
            step(i)
        torch.cuda.synchronize()
        _lib.check(lib.ultr_prof_collect(tot, cnt), "ultr_prof_collect")
        cal_us = [1e3 * tot[k] / max(cnt[k], 1) for k in range(8)]
        cal_cnt = [int(cnt[k]) for k in range(8)]
        dom = int(np.argmax(cal_us))
        # reading back 400 event pairs leaves the GPU idle for a millisecond or two (clocks drop): a few more untimed steps
        # so that a SHORT timed region (the driver runs --steps 20 = 1.2 ms) starts on a warm device like a long one does
        for i in range(16):
            step(i)
        _lib.check(lib.ultr_prof_set_stride(stride), "ultr_prof_set_stride")
        _lib.check(lib.ultr_prof_enable(1 << dom, 4 * (args.steps // stride + 2)), "ultr_prof_enable")  # x4: uncounted shadow samples
    # ---- data parallel: is the one-kernel exchange trustworthy on THIS node? --------------------------------------
    # (it has its own start-up self-test; here the product step's own vector is checked against the RCCL all-reduce of the same
    # local gradients, before anything is timed - a mismatch demotes the peer path and `value` is measured with RCCL)
    dp_checks, peer_ok = {}, eng.comm is not None
    if pg is not None and eng.comm is not None:
        a_args = eng._args
        a_args.skip_update = 1  # backward only: grads = this rank's local vector
        step(0)
        a_args.skip_update = 0
        eng.comm.step -= 1  # that call counted an exchange it did not run (epochs and slot parity must not skip)
        local = eng.grads.clone()
        ref = local.clone()
        torch.distributed.all_reduce(ref, group=pg)
        out = torch.empty_like(local)
        eng.comm.allreduce(local, eng.P + eng.tail, eng.P, out, eng.bwd_ws)
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        err = float((out - ref).abs().max())
        ok = torch.tensor([1 if (err <= 1e-5 * max(scale, 1e-30) and eng.comm.status() == 0) else 0], device=device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=pg)
        peer_ok = bool(int(ok.item()))
        dp_checks["peer_exchange_matches_rccl_allreduce"] = peer_ok
        dp_checks["peer_vs_rccl_max_abs_err_over_max"] = err / max(scale, 1e-30)
        if not peer_ok:
            print("WARNING: the hipIpc exchange kernel disagrees with the RCCL all-reduce on this node - using RCCL", file=sys.stderr)
            eng = engs["main"] = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP,
                                         process_group=pg, no_peer_comm=True)
            for i in range(args.warmup):
                step(i)
    # ---- data parallel: the exchange protocol under SKEW, before anything is timed: 1 000 steps on private
    # copies of the replicas, every rank delaying every step by its own 0 - 50 us (a busy-wait kernel in front of the step: the
    # publish of this rank's vector and its flags arrive that much later than its peers'), then: replicas bit-identical, every
    # status word zero.  The first run on real peers yields a verdict on the protocol, not just a throughput number.
    dp_selftest = None
    if pg is not None and not args.no_dp_selftest:
        n_self = args.dp_selftest_steps
        sp, ss = params.clone(), None if state is None else state.clone()
        sa = None if aux is None else aux.clone()
        e_self = engs["selftest"] = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP, process_group=pg,
                                            no_peer_comm=eng.comm is None)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        torch.cuda._sleep(2000000)
        ev1.record()
        torch.cuda.synchronize()
        cyc_per_us = 2000000.0 / (1e3 * ev0.elapsed_time(ev1))
        rs = np.random.RandomState(4242 + 7919 * rank)
        delays = rs.randint(0, 51, size=n_self)
        err = None
        barrier()
        ta = time.perf_counter()
        try:
            for i in range(n_self):
                if delays[i] > 0:
                    torch.cuda._sleep(int(delays[i] * cyc_per_us))
                f, nd, ids, y, _ = pool[i % npool]
                e_self.train_step(sp, ss, f, nd, ids, y, aux=sa, ipw_table=ipw)
                if (i & 15) == 15:
                    e_self.read_scalars()  # raises on a raised status word (exchange timeout on any rank)
            e_self.read_scalars()
        except Exception as ex:  # the verdict goes on the line; the run continues on whatever path still works
            err = repr(ex)
        barrier()
        t_self = time.perf_counter() - ta
        chk = torch.stack([sp.double().sum(), (sp.double() * torch.arange(P, device=device, dtype=torch.float64)).sum()])
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=pg)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=pg)
        same = bool(torch.equal(lo, hi))
        status = 0 if e_self.comm is None else int(e_self.comm.status())
        st_t = torch.tensor([status if err is None else max(status, 1)], device=device)
        torch.distributed.all_reduce(st_t, op=torch.distributed.ReduceOp.MAX, group=pg)
        dp_selftest = {"steps": n_self, "delay_us": "0-50 per rank and step (seeded), a busy-wait kernel in front of the step",
                       "exchange": "peer kernel" if e_self.comm is not None else "process-group all-reduce",
                       "replicas_bit_identical": same, "status_words_zero": int(st_t.item()) == 0, "error": err,
                       "passed": bool(same and int(st_t.item()) == 0 and err is None), "ms_per_step": 1e3 * t_self / n_self}
        if not dp_selftest["passed"]:
            print("WARNING: the data-parallel self-test FAILED: %r" % (dp_selftest,), file=sys.stderr)
        e_self.close()
        del engs["selftest"]
    # ---- the timed region: EXACTLY K steps, each followed by the host's read of its loss (the reference's loss.item()) ----
    def timed_region():
        barrier()
        ta = time.perf_counter()
        for i in range(args.steps):
            step(i)
            eng.read_loss()
        barrier()
        tb = time.perf_counter()
        ds, dn = None, 0
        if dnn:
            _lib.check(lib.ultr_prof_collect(tot, cnt), "ultr_prof_collect")
            lib.ultr_prof_enable(0, 0)
            ds = 1e-3 * tot[dom] / max(cnt[dom], 1)
            dn = int(cnt[dom])
            lib.ultr_prof_set_stride(1)
        return ta, tb, ds, dn

    # A 20-step region is ~1 ms of wall time: ONE descheduling of this host thread (seen on the round's shared boxes: 0.9 and 3.4 ms,
    # twice in ~30 runs, the timed kernel's own duration unchanged) multiplies the figure.  On one GPU a region that took more than
    # 1.6 x (K x the calibration pass's kernel sum) is measured AGAIN - the same K steps, the same brackets - and the discarded
    # figure is reported next to the kept one (`timed_region_discarded_ms_per_step`).
    discarded = []
    while True:
        t0, t1, dom_s, dom_samples = timed_region()
        expected = 1e-6 * sum(cal_us[k] for k in KSLOTS if cal_cnt[k] > 0) * args.steps if dnn else 0.0
        if not (dnn and world == 1 and expected > 0 and (t1 - t0) > 1.6 * expected and len(discarded) < 2):
            break
        discarded.append(1e3 * (t1 - t0) / args.steps)
        for i in range(8):
            step(i)
        _lib.check(lib.ultr_prof_set_stride(stride), "ultr_prof_set_stride")
        _lib.check(lib.ultr_prof_enable(1 << dom, 4 * (args.steps // stride + 2)), "ultr_prof_enable")
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if pg is not None:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    final_loss = eng.read_loss()
    assert np.isfinite(final_loss), "training diverged"
    comm_status = 0 if getattr(eng, "comm", None) is None else eng.comm.status()
    assert comm_status == 0, "a peer wait of the gradient exchange timed out"

    def timed_loop(e, n, sync_each):
        for i in range(min(20, n)):
            step(i, e)
        barrier()
        ta = time.perf_counter()
        for i in range(n):
            step(i, e)
            if sync_each:
                e.read_loss()
        barrier()
        tb = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=device)
        if pg is not None:
            torch.distributed.all_reduce(tb, op=torch.distributed.ReduceOp.MAX)
        return float(tb.item()) / n

    def exchange_alone(e, nrep=200):
        # the exchange alone (all ranks in lockstep): what one step pays for data parallelism on top of the 1-GPU step
        e.grads.zero_()  # repeated sums of a live gradient would overflow; zeros stay zeros, the traffic is the same
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            e.dp_reduce()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        return 1e3 * e0.elapsed_time(e1) / nrep

    # the same loop without the per-step read of the loss (the host runs ahead of the GPU)
    dom_stream_us = None
    if dnn and (not args.no_extras or world > 1):
        # the dominant kernel's dispatch timestamps in THIS loop too: back-to-back launches (what a rocprofv3 average over the
        # whole process is dominated by) - the figure inside the timed region is higher because every step there starts on a
        # GPU that idled while the host read the previous loss
        _lib.check(lib.ultr_prof_set_stride(stride), "ultr_prof_set_stride")
        _lib.check(lib.ultr_prof_enable(1 << dom, 4 * (max(100, min(args.steps, 2000)) // stride + 4)), "ultr_prof_enable")
    nosync = timed_loop(eng, max(100, min(args.steps, 2000)), False) if not args.no_extras or world > 1 else None
    if dnn and nosync is not None:
        tot2, cnt2 = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
        _lib.check(lib.ultr_prof_collect(tot2, cnt2), "ultr_prof_collect")
        lib.ultr_prof_enable(0, 0)
        lib.ultr_prof_set_stride(1)
        if cnt2[dom] > 0:
            dom_stream_us = 1e3 * tot2[dom] / cnt2[dom]

    # the same synced loop with the wide layers' products on the fp32 matrix cores (round 2's arithmetic, bit for bit): the
    # split-half fp16 products are fp32-accurate, this line is for a reader who wants the all-fp32-MFMA figure
    fp32_mfma = None
    if dnn and world == 1 and not args.no_extras and h3_products_on():
        keep = {k: os.environ.get(k) for k in ("ULTR_FB_H3", "ULTR_FWD_H3", "ULTR_BWD_H3", "ULTR_WG_H3")}
        try:
            for k in keep:
                os.environ[k] = "0"
            e32 = engs["fp32_mfma"] = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP)  # re-reads the knobs
            t32 = timed_loop(e32, max(100, min(args.steps, 1000)), True)
            fp32_mfma = {"queries_per_sec": B / t32, "ms_per_step": 1e3 * t32,
                         "what": "ULTR_FB_H3=0 ULTR_FWD_H3=0 ULTR_BWD_H3=0 ULTR_WG_H3=0: every product on v_mfma_f32_16x16x4_f32; loss read on the host every step"}
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            lib.ultr_config_reload()

    allreduce_us, dp_exchanges, rccl_ranks = None, None, None
    NAME_PEER = "ultr_comm_allreduce (one kernel, hipIpc peer reads over xGMI)"
    NAME_RCCL = "process-group all-reduce (RCCL) + ultr_grad_sumsq"
    if pg is not None:
        # replicas must be bit-identical after the timed region (every rank applied the same summed vector).  If they are not
        # and the exchange was the peer kernel, that path is broken on this node: say so loudly and let `value` come from the
        # RCCL loop below instead of losing the line (throughput does not depend on the parameter values)
        def replicas_identical():
            chk = torch.stack([params.double().sum(), (params.double() * torch.arange(P, device=device, dtype=torch.float64)).sum()])
            lo, hi = chk.clone(), chk.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=pg)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=pg)
            return bool(torch.equal(lo, hi))

        main_ok = replicas_identical()
        dp_checks["replicas_bit_identical_after_timed_region"] = main_ok
        if not main_ok:
            print("WARNING: data-parallel replicas DIVERGED with %s" % ("the hipIpc exchange kernel" if eng.comm is not None else "RCCL"),
                  file=sys.stderr)
            assert eng.comm is not None, "replicas diverged with the RCCL all-reduce"
            with torch.no_grad():  # put the replicas back on one state for the RCCL loop: rank 0's parameters
                torch.distributed.broadcast(params, src=0, group=pg)
                if state is not None:
                    torch.distributed.broadcast(state, src=0, group=pg)
        # BOTH exchanges in this run: `value` above was measured with the default one; the other is timed here the same way
        main_is_peer = eng.comm is not None
        ms_main = (t1 - t0) / args.steps
        dp_exchanges = {}
        n_alt = max(100, min(args.steps, 1000))
        rec_main = {"ms_per_step": 1e3 * elapsed / args.steps, "queries_per_sec": world * B * args.steps / elapsed,
                    "ms_per_step_no_host_sync": None if nosync is None else 1e3 * nosync,
                    "exchange_alone_us": exchange_alone(eng), "measured_as": "value (the timed region)"}
        dp_exchanges[NAME_PEER if main_is_peer else NAME_RCCL] = rec_main
        allreduce_us = rec_main["exchange_alone_us"]
        if main_is_peer:
            alt = engs["rccl"] = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP,
                                         process_group=pg, no_peer_comm=True)
            t_alt = timed_loop(alt, n_alt, True)
            t_alt_ns = timed_loop(alt, n_alt, False)
            dp_exchanges[NAME_RCCL] = {"ms_per_step": 1e3 * t_alt, "queries_per_sec": world * B / t_alt,
                                       "ms_per_step_no_host_sync": 1e3 * t_alt_ns, "exchange_alone_us": exchange_alone(alt),
                                       "measured_as": "%d steps, same loop as the timed region, right after it" % n_alt}
            rccl_ranks = torch.distributed.get_world_size(pg)  # a communicator that all-reduced `grads` in this run
            if not main_ok:  # the peer path is broken here: the line's headline figures come from the RCCL loop
                rec_main["measured_as"] = "the timed region - REPLICAS DIVERGED, not used for `value`"
                dp_exchanges[NAME_RCCL]["measured_as"] += " - used for `value` (the peer exchange diverged)"
                elapsed, nosync = t_alt * args.steps, t_alt_ns
                eng = alt
        else:
            rccl_ranks = torch.distributed.get_world_size(pg)
            if not peer_ok or os.environ.get("ULTR_DP_COMM", "peer") != "peer":
                dp_exchanges[NAME_PEER] = {"skipped": "not available / failed verification on this node"
                                           if os.environ.get("ULTR_DP_COMM", "peer") == "peer" else "ULTR_DP_COMM=pg"}
        dp_checks["replicas_bit_identical_after_run"] = replicas_identical()
        assert dp_checks["replicas_bit_identical_after_run"] or not main_ok, "data-parallel replicas diverged"
        # the SAME step without the exchange, in this run, on every rank at once (its own copies of the parameters: the replicas are
        # not disturbed): what one GPU does alone on this node right now - the denominator of the weak-scaling efficiency
        solo_params, solo_state = params.clone(), None if state is None else state.clone()
        solo_aux = None if aux is None else aux.clone()
        e1 = engs["solo"] = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP)

        def solo_step(i):
            f, nd, ids, y, _ = pool[i % npool]
            return e1.train_step(solo_params, solo_state, f, nd, ids, y, aux=solo_aux, ipw_table=ipw)
        for i in range(20):
            solo_step(i)
        barrier()
        n_solo = max(100, min(args.steps, 1000))
        ta = time.perf_counter()
        for i in range(n_solo):
            solo_step(i)
            e1.read_loss()
        torch.cuda.synchronize()
        tsolo = torch.tensor([(time.perf_counter() - ta) / n_solo], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tsolo, op=torch.distributed.ReduceOp.MAX)
        solo_ms = 1e3 * float(tsolo.item())

    e2e, plugin = None, None
    if world == 1 and not args.no_extras and light:
        # end-to-end variant (the reference's own step_time definition, main.py:153-156: get_batch + train): the
        # dataset is resident in HBM and clicks are simulated on the device (ultr_click_batch)
        nq = 20000
        rng = np.random.RandomState(99)
        rfeat = torch.from_numpy(rng.uniform(-1, 1, size=(nq * L, F)).astype(np.float32)).to(device)
        rlists = torch.arange(nq * L, dtype=torch.int32, device=device).view(nq, L).contiguous()
        rel = rng.randint(0, 5, size=(nq, L)).astype(np.float32)
        rel[:, 0] = np.maximum(rel[:, 0], 1)
        rlab = torch.from_numpy(rel).to(device)
        exam_np, cp_np = synthetic.load_pbm()
        exam = torch.tensor(exam_np, dtype=torch.float32, device=device)
        cprob = torch.tensor(cp_np, dtype=torch.float32, device=device)
        dids = torch.empty(L, B, dtype=torch.int32, device=device)
        dclk = torch.empty(L, B, dtype=torch.float32, device=device)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())

        def e2e_step(i):
            _lib.check(lib.ultr_click_batch(vp(rlists), vp(rlab), nq, L, nq * L, vp(exam), int(exam.numel()), vp(cprob),
                                            int(cprob.numel()), 0, 1234, i, B, L, 100, vp(dids), vp(dclk), None,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ultr_click_batch")
            return eng.train_step(params, state, rfeat, nq * L, dids, dclk, ipw_table=ipw)

        for i in range(50):
            e2e_step(i)
        barrier()
        n3 = max(200, min(args.steps, 1000))
        t4 = time.perf_counter()
        for i in range(n3):
            e2e_step(50 + i)
        barrier()
        e2e = B * n3 / (time.perf_counter() - t4)
        plugin = plugin_rates(cfg, pool, device, args.steps)
    trb = None
    if world == 1 and not args.no_extras and not args.no_cpu_baseline:
        try:
            trb = torch_rocm_baseline(cfg, pool, params0, device)
        except Exception as ex:  # a baseline must never take the headline down
            trb = {"value": None, "error": repr(ex)}

    other, evalleg = None, None
    if world == 1 and light and not args.no_other_configs:
        # the other BASELINE configs and the validation path on the SAME line: short synced runs
        evalleg = eval_leg(cfg, device, lib, params0)
        other = {}
        for key, (n_st, n_wu) in (("3", (60, 30)), ("4pair", (40, 20)), ("4lambda", (40, 20)), ("5", (20, 4))):
            try:
                other[key] = short_config_run(key, device, lib, n_st, n_wu)
                other[key]["fp32_mfma_products"] = short_config_run(key, device, lib, max(10, n_st // 2), max(4, n_wu // 2), fp32=True)
            except Exception as ex:  # a secondary figure must never take the headline down
                other.setdefault(key, {})["error"] = repr(ex)
    if rank == 0:
        flops = step_flops(cfg)
        ms_step = 1e3 * elapsed / args.steps
        if dnn:
            bound, amount = algorithmic_work(cfg, P)[dom]
            kname = KNAMES[dom]
        else:  # SetRank: ~80 launches per step, no single dominant kernel - the whole step against the matrix peak
            bound, amount, kname, dom_s, dom_samples = "mfma", flops, "whole step (all launches)", 1e-3 * ms_step, args.steps
        # `bound` names the PEAK the kernel is priced against (its work is a dense contraction -> the fp32 matrix cores);
        # `limited_by` says what the measurements show actually limits it
        limited_by = None
        if not dnn:
            limited_by = ("HBM traffic + instruction issue, not the matrix pipes: ~8.5 GB cross HBM per step (profiles/r05_cfg5_pmc.md; 9.9 GB "
                          "before round 5 fused the forward's Linear / LayerNorm launches into sr_embed_fwd_kernel / sr_block_fwd_kernel), the "
                          "remaining split-half GEMMs (the backward's dgrads) sit at 7-11 % matrix-core occupancy, the fused block kernels are "
                          "chains of ~10k-cycle phases per workgroup, the attention backward is VALU-bound")
        if dnn and light and dom == 7:
            limited_by = ("a per-workgroup LATENCY CHAIN, nothing is at a bandwidth limit: one 10-document list per compute unit in a 16-row MFMA "
                          "tile (8 waves); 55 % of the kernel is dependent row-wise phases (LayerNorms, loss, backward row passes), 45 % the "
                          "three products, whose weight stream (l2_stream_frac below) runs ~2x off the L2's rate and whose matrix-core work "
                          "(mfma_pipe_frac) is a few per cent of the launch")
        if bound == "mfma":
            achieved, peak, unit = amount / dom_s / 1e12, PEAK_FP32_MFMA_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = amount / dom_s / 1e9, PEAK_HBM_GBS, "GB/s"
        # machine-readable honesty about the matrix-core dtype: what is issued, against the peak of what is issued
        roofline_extras = {}
        if dnn and bound == "mfma":
            f16_issued, f32_issued = issued_matrix_work(cfg, dom)
            pipe_s = f16_issued / (PEAK_F16_MFMA_TFLOPS * 1e12) + f32_issued / (PEAK_FP32_MFMA_TFLOPS * 1e12)
            roofline_extras = {
                "mfma_dtype": ("f16 x3 (split hi/lo operands, f32 accumulate)" if f32_issued == 0 else "f16 x3 (split hi/lo) + f32") if f16_issued > 0 else "f32",
                "issued_f16_flops_per_launch": f16_issued, "issued_f32_flops_per_launch": f32_issued,
                "frac_of_issued_dtype_peak": pipe_s / dom_s,
                "frac_of_issued_dtype_peak_note": "time the matrix pipes need for what is ISSUED at their dense peaks (f16 2500, f32 157.3 TFLOP/s) "
                                                  "/ launch duration; `frac` above prices the ALGORITHMIC fp32 flops against the fp32 peak instead "
                                                  "(a comparison with a perfect fp32 implementation, not a ceiling of this kernel)"}
            if dom == 7:
                # every workgroup of the fused kernel streams every weight copy it multiplies with through its XCD's L2: 4 bytes per
                # weight (fp32, or hi + lo halves) x [forward layers + dgrad layers >= 1] x workgroups; duration measured
                dims = dnn_dims(cfg)
                wbytes = 4.0 * (sum(k * m for k, m in dims[:-1]) + sum(k * m for k, m in dims[1:-1]))
                lpb = max(1, 16 // cfg["L"])
                nwg = (cfg["B"] + lpb - 1) // lpb
                roofline_extras.update({"weight_stream_bytes_per_launch": wbytes * nwg, "weight_stream_tbps": wbytes * nwg / dom_s / 1e12,
                                        "l2_stream_frac": wbytes * nwg / dom_s / 1e12 / L2_STREAM_PEAK_TBS,
                                        "l2_stream_note": "workgroups x weight-copy bytes (computed) / measured launch duration, against "
                                                          "8 L2s x 2 KB/clk x 2.4 GHz; the products are ~45 % of the launch, so inside them "
                                                          "the stream runs at about twice this fraction"})
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes/launch from rocprofv3 PMC passes
        if os.path.exists(tfile) and (light or not dnn):
            tj = json.load(open(tfile))
            traffic = tj.get(kname if dnn else "setrank_whole_step")
            traffic_source = "file profiles/traffic.json (%s): rocprofv3 FETCH_SIZE/WRITE_SIZE passes of an earlier run of this " \
                             "command, NOT measured in this process" % tj.get("_source" if dnn else "_source_setrank", "tools/profile_round.sh")
        dp_head = {}
        if pg is not None:  # in front of the long strings: a record that keeps only the head of the line still shows what exchanged
            dp_head = {"dp_exchange": NAME_PEER if eng.comm is not None else NAME_RCCL,  # the one `value` was measured with
                       "rccl_ranks": rccl_ranks,  # world size of the RCCL communicator that all-reduced `grads` in THIS run
                       "weak_scaling_efficiency": (world * B * args.steps / elapsed) / (world * B / (1e-3 * solo_ms)),
                       "exchange_exposed_us": 1e3 * (1e3 * elapsed / args.steps - solo_ms),
                       "dp_selftest": dp_selftest}
        out = {
            "metric": "queries/sec (training step)", "value": world * B * args.steps / elapsed, "unit": "queries/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, **dp_head,
            "dtype": dtype_label(cfg, args.attention_dtype),
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "baseline_config": args.config, "global_batch": world * B, "list_size": L,
                       "feature_size": F, "hidden": HIDDEN, "parallelism": "dp%d" % world, "params": P},
            "roofline": {"kernel": kname, "bound": bound, "limited_by": limited_by, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "peak_note": ("fp32 MFMA dense peak; the kernel's products are counted as fp32 multiply-adds although "
                                                                "they run as three f16 MFMAs on split operands") if (dnn and bound == "mfma" and h3_products_on()) else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         **roofline_extras,
                         "avg_launch_us": 1e6 * dom_s, "launches_timed": dom_samples, "algorithmic_per_launch": amount,
                         # measurement hygiene of `value` (here, inside an object the driver stores whole): a timed region slower than
                         # 1.6x the calibration sum is measured again (attempts > 1, the discarded figures listed); untimed spin-up
                         "timed_region_attempts": 1 + len(discarded), "timed_region_discarded_ms_per_step": discarded,
                         "spinup_ms": spinup_ms, "spinup_steps": spun,
                         "avg_launch_us_back_to_back": dom_stream_us,
                         "avg_launch_us_note": ("avg_launch_us (and `achieved`) is measured INSIDE the timed region, where every step "
                                                "starts on a GPU that idled while the host read the previous loss; "
                                                "avg_launch_us_back_to_back = the same kernel in the loop without host reads, "
                                                "which is what a rocprofv3 average over the whole process mostly sees") if dnn else None,
                         "stage_note": ("the forward / backward slots are STAGES (`stage_kernels` names what they launch for this shape): "
                                        "the wide-tile kernels dnn_fwdw_kernel / dnn_bwdw_kernel (round 5: configs 3 and 4), the 16-row tiles "
                                        "dnn_fwd_kernel / dnn_bwd2_kernel, or the per-layer launches of ultr_dnn_big.hip (one sample = first "
                                        "launch's start to last launch's stop)") if dnn else None},
            "step_tflops": flops / (1e-3 * ms_step) / 1e12, "step_frac_of_fp32_mfma_peak": flops / (1e-3 * ms_step) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_us": {KNAMES[k]: round(cal_us[k], 3) for k in KSLOTS if cal_cnt[k] > 0} if dnn else {},
            "stage_kernels": stage_kernels(lib, cfg),
            "kernel_us_source": "calibration pass in front of the timed region (all kernel timers armed); roofline.avg_launch_us is "
                                "the dominant kernel INSIDE the timed region",
            "final_loss": final_loss,
        }
        out["value_definition"] = "every step followed by the host's read of its loss (the reference's loss.item(), SURVEY 8d)"
        out["timed_region_attempts"] = 1 + len(discarded)
        out["timed_region_discarded_ms_per_step"] = discarded
        out["spinup_ms"] = spinup_ms
        out["spinup_steps"] = spun
        out["spinup_note"] = ("untimed steps in front of the W warm-up steps so that the shader clock is at its sustained level when a "
                              "SHORT timed region starts (--steps 20 is 1 ms of GPU work); --spinup-ms 0 switches it off")
        if other is not None:
            out["other_configs"] = other
        if evalleg is not None:
            out["eval"] = evalleg
        if fp32_mfma is not None:
            out["fp32_mfma_products"] = fp32_mfma
        if nosync is not None:
            out["queries_per_sec_no_host_sync"] = B * world / nosync
            out["ms_per_step_no_host_sync"] = 1e3 * nosync
        if pg is not None:
            out["dp_exchanges"] = dp_exchanges
            out["allreduce_us"] = allreduce_us
            out["dp_checks"] = dp_checks
            out["single_gpu_in_run"] = {"ms_per_step": solo_ms, "queries_per_sec": B / (1e-3 * solo_ms),
                                        "what": "the same synced step WITHOUT the exchange on every rank at once, in this run (slowest rank)"}
        if trb is not None:
            out["torch_rocm_baseline"] = trb
        if e2e is not None:
            out["end_to_end_queries_per_sec_device_feed"] = e2e  # batch construction (click simulation) + train step
        if plugin is not None:
            out["plugin_queries_per_sec"] = plugin[0]  # IPWrank.train(host numpy feed): marshal + PCIe + .item() + print
            out["plugin_device_feed_queries_per_sec"] = plugin[1]  # IPWrank.train(DeviceClickFeed batch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, pool, params0)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)  # whatever the teardown prints (RCCL's banner arrives at communicator destruction) must not follow the line
    for e in engs.values():
        e.close()
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
