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

Prints ONE JSON line on rank 0 (see the contract in the task statement); extra keys: `roofline` (dominant kernel,
timed inside the timed region from its own dispatch packets), `cpu_baseline` (the oracle = a torch-CPU port of the
reference's step, timed on this box's host cores on the same workload), `kernel_us` (per-kernel average, calibration pass),
`plugin_*` (the same step through the reference-shaped plugin API), `dp_exchange` / `rccl_ranks` / `allreduce_us`.
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
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
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


def cpu_baseline(cfg, pool, params0, budget_s=10.0):
    """The oracle's step (vectorised torch-CPU port of the reference's train()) on this box's host cores.
    The thread count is chosen by a short probe (small GEMMs do not scale to every core of a big host, and an
    oversubscribed baseline would flatter the GPU); `cores` reports the threads actually used."""
    ncpu = os.cpu_count() or 1
    B = cfg["B"]
    step = oracle_stepper(cfg)
    npool = len(pool)

    def run(nsteps, st, i0):
        t = 0.0
        for i in range(i0, i0 + nsteps):
            t0 = time.perf_counter()
            st = step(st, pool[i % npool][4])
            t += time.perf_counter() - t0
        return t, st

    heavy = cfg["model"] == "setrank" or cfg["F"] * cfg["L"] * B > 2e6
    best, probe = None, {}
    for th in sorted({1, 4, 8, 16, 32, 64, ncpu} if not heavy else {8, 32, 64}):
        if th > ncpu:
            continue
        torch.set_num_threads(th)
        st = cpu_state0(cfg, params0)
        _, st = run(1, st, 0)
        t, st = run(1 if heavy else 5, st, 1)
        probe[th] = t / (1 if heavy else 5)
        if best is None or probe[th] < probe[best]:
            best = th
    torch.set_num_threads(best)
    st = cpu_state0(cfg, params0)
    _, st = run(1 if heavy else 5, st, 0)
    n, t_used = 0, 0.0
    chunk = 1 if heavy else 10
    while t_used < budget_s or n < (2 if heavy else 20):
        t, st = run(chunk, st, 5 + n)
        n += chunk
        t_used += t
    out = {"value": B * n / t_used, "unit": "queries/sec", "cores": best, "kind": "port",
           "sample": "%d steps of the same workload after warm-up, oracle/ultr_oracle (vectorised torch-CPU restatement), "
                     "%d threads picked by probe %s (host has %d), %.2f ms/step"
                     % (n, best, {k: round(1e3 * v, 2) for k, v in probe.items()}, ncpu, 1e3 * t_used / n)}
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
        for i in range(20):
            algo.train(feed_obj.get_batch(ds)[0])
        n = max(100, min(steps, 500))
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
    ap.add_argument("--sync-every-step", action="store_true", help="also read loss.item() every step (API-faithful)")
    args = ap.parse_args()
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
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from ultra_pytorch_amd import parallel
    _, _, _, pg = parallel.init_process_group_from_env(backend="nccl")

    from ultra_pytorch_amd import _lib, engine, hip_ops, synthetic
    lib = _lib.load()
    if cfg["model"] == "setrank":
        from ultra_pytorch_amd.ranking_model.SetRank import init_setrank_params
        shape = hip_ops.SetRankShape(F, 256, 8, 2, 64, attention_dtype=args.attention_dtype)
        params0 = init_setrank_params(shape, seed=0).numpy()
        eng_cls = engine.SetRankStepEngine
    else:
        from ultra_pytorch_amd.ranking_model import init_flat_params
        shape = hip_ops.DnnShape(F, HIDDEN, "elu")
        params0 = init_flat_params(shape, seed=0).numpy()
        eng_cls = engine.StepEngine
    P = shape.n_params
    params = torch.from_numpy(params0.copy()).to(device)  # identical replicas on every rank
    state = None if cfg["algo"] == "dla" else torch.zeros_like(params)
    aux = None
    if cfg["algo"] == "dla":
        aux = torch.zeros(L + 1, device=device)
    elif cfg["algo"] in ("pairdebias", "lambdarank"):
        aux = torch.ones(2 * L, device=device)
    ipw = torch.tensor(synthetic.load_ipw(), dtype=torch.float32, device=device) if cfg["algo"] == "softmax" else None
    pool = make_pool(cfg, np.random.RandomState(1234 + rank), device)
    npool = len(pool)
    eng = eng_cls(shape, B, L, device, algo=cfg["algo"], learning_rate=LR, max_gradient_norm=CLIP, process_group=pg)

    def step(i):
        f, nd, ids, y, _ = pool[i % npool]
        return eng.train_step(params, state, f, nd, ids, y, aux=aux, ipw_table=ipw)

    _flag = torch.zeros(1, device=device)

    def barrier():
        # all ranks + the device: a one-element all-reduce ENQUEUED behind the queued steps (it completes only when every rank
        # has reached it), then one host synchronisation - torch.distributed.barrier() would cost a second host round trip
        # inside the timed region (forced data-parallel run at 20 steps: 62.5 -> 60.9 us/step)
        if pg is not None:
            torch.distributed.all_reduce(_flag, group=pg)
        torch.cuda.synchronize()

    # ---- warm-up (untimed) + pick the dominant kernel with all timers armed ------------------------
    for i in range(args.warmup):
        step(i)
    barrier()
    tot, cnt = (ctypes.c_double * 8)(), (ctypes.c_int64 * 8)()
    dnn = cfg["model"] == "dnn"
    dom = None
    # Timers inside the timed region: ONLY the dominant kernel, on every stride-th step, by the start / stop timestamps of its
    # own dispatch packet (what rocprofv3 --kernel-trace reports).  A timed launch costs ~5 us of stream time (tools/
    # short_region.py: all kernels timed at stride 8 cost 2.3 us per step of the headline, 20 us per timed step), so the other
    # kernels' averages come from the calibration pass in front of the timed region (`kernel_us`, all timers armed; the
    # dominant kernel's figure there and inside the region agree to 1 %).
    stride = (8 if args.steps < 256 else 32) if light else 4
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
        _lib.check(lib.ultr_prof_enable(1 << dom, args.steps // stride + 2), "ultr_prof_enable")
    # ---- the timed region: EXACTLY K steps ----------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    t1 = time.perf_counter()
    dom_s, dom_samples = None, 0
    if dnn:
        _lib.check(lib.ultr_prof_collect(tot, cnt), "ultr_prof_collect")
        lib.ultr_prof_enable(0, 0)
        dom_s = 1e-3 * tot[dom] / max(cnt[dom], 1)
        dom_samples = int(cnt[dom])
        lib.ultr_prof_set_stride(1)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if pg is not None:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    final_loss = float(eng.scalars[0].item())
    assert np.isfinite(final_loss), "training diverged"
    comm_status = 0 if getattr(eng, "comm", None) is None else eng.comm.status()
    assert comm_status == 0, "a peer wait of the gradient exchange timed out"

    allreduce_us = None
    if pg is not None:
        # the exchange alone (all ranks in lockstep): what one step pays for data parallelism on top of the 1-GPU step
        nrep = 200
        eng.grads.zero_()  # repeated sums of a live gradient would overflow; zeros stay zeros, the traffic is the same
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            eng.dp_reduce()
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = 1e3 * e0.elapsed_time(e1) / nrep
        barrier()

    synced = None
    if args.sync_every_step or (world == 1 and not args.no_extras):
        n2 = max(100, min(args.steps, 500))  # secondary figures keep their own minimum loop length
        barrier()
        t2 = time.perf_counter()
        for i in range(n2):
            step(i)[0].item()  # the reference's loss.item() each step
        t3 = time.perf_counter()
        synced = (t3 - t2) / n2

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
                                            int(cprob.numel()), 1234, i, B, L, 100, vp(dids), vp(dclk), None,
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

    if rank == 0:
        flops = step_flops(cfg)
        ms_step = 1e3 * elapsed / args.steps
        if dnn:
            bound, amount = algorithmic_work(cfg, P)[dom]
            kname = KNAMES[dom]
        else:  # SetRank: ~80 launches per step, no single dominant kernel - the whole step against the matrix peak
            bound, amount, kname, dom_s, dom_samples = "mfma", flops, "whole step (all launches)", 1e-3 * ms_step, args.steps
        if bound == "mfma":
            achieved, peak, unit = amount / dom_s / 1e12, PEAK_FP32_MFMA_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = amount / dom_s / 1e9, PEAK_HBM_GBS, "GB/s"
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes/launch from rocprofv3 PMC passes
        if os.path.exists(tfile) and light:
            tj = json.load(open(tfile))
            traffic = tj.get(kname)
            traffic_source = "file profiles/traffic.json (%s): rocprofv3 FETCH_SIZE/WRITE_SIZE passes of an earlier run of this " \
                             "command, NOT measured in this process" % tj.get("_source", "tools/profile_round.sh")
        out = {
            "metric": "queries/sec (training step)", "value": world * B * args.steps / elapsed, "unit": "queries/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if (dnn or args.attention_dtype == "fp32") else "f32 (self-attention operands f16, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "baseline_config": args.config, "global_batch": world * B, "list_size": L,
                       "feature_size": F, "hidden": HIDDEN, "parallelism": "dp%d" % world, "params": P},
            "roofline": {"kernel": kname, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_us": 1e6 * dom_s, "launches_timed": dom_samples, "algorithmic_per_launch": amount,
                         "stage_note": ("the forward / backward slots are STAGES: dnn_fwd_kernel / dnn_bwd2_kernel, or - where the "
                                        "launcher's measured rule sends the shape (config 4: both) - the per-layer launches of "
                                        "ultr_dnn_big.hip (statistics passes + tiled GEMMs), one sample = first launch's start to last "
                                        "launch's stop") if dnn else None},
            "step_tflops": flops / (1e-3 * ms_step) / 1e12, "step_frac_of_fp32_mfma_peak": flops / (1e-3 * ms_step) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_us": {KNAMES[k]: round(cal_us[k], 3) for k in KSLOTS if cal_cnt[k] > 0} if dnn else {},
            "kernel_us_source": "calibration pass in front of the timed region (all kernel timers armed); roofline.avg_launch_us is "
                                "the dominant kernel INSIDE the timed region",
            "final_loss": final_loss,
        }
        if pg is not None:
            out["rccl_ranks"] = world
            out["dp_exchange"] = "ultr_comm_allreduce (one kernel, hipIpc peer reads over xGMI)" if eng.comm is not None \
                else "process-group all-reduce (RCCL) + ultr_grad_sumsq"
            out["allreduce_us"] = allreduce_us
        if synced is not None:
            out["queries_per_sec_with_loss_item_each_step"] = B * world / synced
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
    if getattr(eng, "comm", None) is not None:
        eng.comm.close()
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
