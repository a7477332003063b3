"""The PRODUCT data-parallel path at world size 2, on one GPU: two processes share cuda:0, each takes its
shard of ONE global batch (parallel.shard_input_feed; uneven shards on purpose), runs StepEngine(process_group=...)
.train_step - i.e. ultr_train_step(skip_update) -> the gradient exchange -> ultr_apply_update - and must end with the
parameters / Adagrad state / EM state of the single-process step on the whole batch and of the oracle.

Both exchange paths run: "peer" = ultr_comm_allreduce (hipIpc-mapped exchange buffers, one kernel; the two processes
map each other's buffer exactly as two GPUs of a node would) and "pg" = the process group's all-reduce + ultr_grad_sumsq
(gloo here, RCCL in bench.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, HIDDEN, B, L = 24, [32, 16], 7, 6  # 7 lists -> shards of 4 and 3
ALGOS = ["softmax", "pairdebias", "lambdarank", "dla", "regem"]


def make_global(seed):
    rng = np.random.RandomState(seed)
    n_docs = B * L - 3
    feats = rng.uniform(-1, 1, size=(n_docs, F)).astype(np.float32)
    ids = rng.permutation(B * L)
    ids = np.where(ids >= n_docs, n_docs, ids).reshape(L, B)  # three PAD documents
    clicks = (rng.uniform(size=(L, B)) < 0.4).astype(np.float32)
    # every list has a click; position 0 is clicked in some lists and not in others (PairDebias normalises its EM ratios by
    # the position-0 sums: 0/0 = NaN when no pair involves position 0 on both sides - the reference does the same)
    clicks[0, :] = np.arange(B) % 2
    clicks[1, :] = 1.0 - clicks[0, :]
    rel = rng.randint(0, 5, size=(L, B)).astype(np.float32)
    names_d = ["docid_input%d" % l for l in range(L)]
    names_l = ["label%d" % l for l in range(L)]
    feed = {"letor_features": feats}
    for l in range(L):
        feed[names_d[l]] = ids[l].astype(np.float32)
        feed[names_l[l]] = clicks[l]
    return feed, names_d, names_l, rel


def feed_arrays(feed, names_d, names_l, labels=None):
    ids = np.stack([np.asarray(feed[n]) for n in names_d]).astype(np.int32)
    y = np.stack([np.asarray(feed[n]) for n in names_l]).astype(np.float32) if labels is None else labels
    return np.asarray(feed["letor_features"], np.float32), ids, y


def initial_state(algo):
    sys.path.insert(0, ROOT)
    from oracle import ultr_oracle as O
    params = O.init_params(F, HIDDEN, seed=5)
    rng = np.random.RandomState(11)
    state = None if algo == "dla" else (0.01 * rng.uniform(size=params.shape)).astype(np.float32)
    if algo == "dla":
        aux = (0.1 * rng.randn(L + 1)).astype(np.float32)
    elif algo in ("pairdebias", "lambdarank"):
        aux = np.linspace(0.9, 1.2, 2 * L).astype(np.float32)
    elif algo == "regem":
        aux = np.linspace(0.9, 0.3, L).astype(np.float32)
    else:
        aux = None
    uniforms = rng.uniform(size=(B, L)).astype(np.float32)
    return params, state, aux, uniforms


def n_steps_of(algo):
    # DLA's per-step optimizers make its update sign-like, p -= lr * g / (|g| + 1e-10) (dla.py:141-177): an element whose
    # gradient is ~0 moves by +-lr depending on rounding, so DLA is compared after ONE step and only where |g| is not ~0
    return 1 if algo == "dla" else 2


def run_engine(algo, feats, ids, y, pg, uniforms):
    """n_steps train steps of the product engine on (feats, ids, y); returns numpy post-state."""
    import torch
    from ultra_pytorch_amd import engine, hip_ops
    dev = torch.device("cuda", torch.cuda.current_device())
    params0, state0, aux0, _ = initial_state(algo)
    shape = hip_ops.DnnShape(F, HIDDEN, "elu")
    Bl = ids.shape[1]
    eng = engine.StepEngine(shape, Bl, L, dev, algo=algo, learning_rate=0.005 if algo == "pairdebias" else 0.05,
                            process_group=pg)
    p = torch.tensor(params0, device=dev)
    st = None if state0 is None else torch.tensor(state0, device=dev)
    aux = None if aux0 is None else torch.tensor(aux0, device=dev)
    ipw = torch.linspace(1.0, 3.0, 4, device=dev) if algo == "softmax" else None  # shorter than L: saturates
    f = torch.tensor(feats, device=dev) if feats.shape[0] > 0 else None
    i, yy = torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
    u = torch.tensor(uniforms, device=dev) if algo == "regem" else None
    losses = []
    for _ in range(n_steps_of(algo)):
        sc = eng.train_step(p, st, f, feats.shape[0], i, yy, aux=aux, ipw_table=ipw, uniforms=u)
        torch.cuda.synchronize()
        losses.append(float(sc[0]))
    status = 0 if eng.comm is None else eng.comm.status()
    out = dict(params=p.cpu().numpy(), grads=eng.grads[:shape.n_params].cpu().numpy(), state=None if st is None else st.cpu().numpy(),
               aux=None if aux is None else aux.cpu().numpy(), losses=losses, norm=float(eng.scalars[1]),
               peer=eng.comm is not None, status=status, batch_total=eng.batch_total, rng_seed=eng.rng_seed)
    if eng.comm is not None:
        eng.comm.close()
    return out


def worker(rank, world, port, mode, algo, q, extra_env=None, multi_gpu=False):
    """multi_gpu: one PHYSICAL GPU per rank (LOCAL_RANK = rank) and the RCCL process group - what bench.py --gpus N and a
    real data-parallel job run; otherwise every rank shares cuda:0 and the group is gloo (the 1-GPU box of the -m gpu tier)."""
    sys.path.insert(0, ROOT)
    local = rank if multi_gpu else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), ULTR_DP_COMM=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.update(extra_env or {})
    import torch
    import torch.distributed as dist
    from ultra_pytorch_amd import parallel
    torch.cuda.set_device(local)
    r, w, _, pg = parallel.init_process_group_from_env(backend="nccl" if multi_gpu else "gloo")
    assert (r, w) == (rank, world) and pg is not None
    feed, names_d, names_l, rel = make_global(3)
    if algo == "lambdarank":  # relevance labels
        for l in range(L):
            feed[names_l[l]] = rel[l]
    local = parallel.shard_input_feed(feed, "letor_features", names_d, names_l, L, rank, world)
    feats, ids, y = feed_arrays(local, names_d, names_l)
    lo, hi = parallel.shard_bounds(B, rank, world)
    assert ids.shape == (L, hi - lo)
    _, _, _, uniforms = initial_state(algo)
    res = run_engine(algo, feats, ids, y, pg, uniforms[lo:hi])
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def single_process(algo):
    feed, names_d, names_l, rel = make_global(3)
    feats, ids, y = feed_arrays(feed, names_d, names_l, rel if algo == "lambdarank" else None)
    _, _, _, uniforms = initial_state(algo)
    return run_engine(algo, feats, ids, y, None, uniforms), (feats, ids, y)


def oracle_two_steps(algo, feats, ids, y):
    from oracle import ultr_oracle as O
    params, state, aux, uniforms = initial_state(algo)
    ids64 = ids.astype(np.int64)
    for _ in range(n_steps_of(algo)):
        if algo == "softmax":
            r = O.train_step_softmax(params, state, F, HIDDEN, feats, ids64, y, ipw_list=np.linspace(1.0, 3.0, 4))
            params, state = r["params"], r["state"]
        elif algo == "pairdebias":
            r = O.pairdebias_step(params, state, aux[:L], aux[L:], F, HIDDEN, feats, ids64, y)
            params, state, aux = r["params"], r["state"], np.concatenate([r["t_plus"].ravel(), r["t_minus"].ravel()])
        elif algo == "lambdarank":
            r = O.lambdarank_step(params, state, aux[:L], aux[L:], F, HIDDEN, feats, ids64, y)
            params, state, aux = r["params"], r["state"], np.concatenate([r["t_plus"].ravel(), r["t_minus"].ravel()])
        elif algo == "dla":
            r = O.dla_step(params, aux, F, HIDDEN, feats, ids64, y)
            params, aux = r["params"], r["prop_params"]
        else:
            r = O.regression_em_step(params, state, aux, uniforms, F, HIDDEN, feats, ids64, y)
            params, state, aux = r["params"], r["state"], r["propensity"].ravel()
    return dict(params=params, state=state, aux=aux, loss=r["loss"])


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("mode", ["peer", "pg"])
@pytest.mark.parametrize("algo", ALGOS)
def test_two_rank_step_equals_single_process(mode, algo):
    two_rank_case(mode, algo, multi_gpu=False)


@pytest.mark.parametrize("mode", ["peer", "pg"])
@pytest.mark.parametrize("algo", ALGOS)
def test_two_physical_gpus_step_equals_single_process(mode, algo):
    """The same assertions with one PHYSICAL GPU per rank and the RCCL process group: the exchange kernel's publish / flag /
    peer-read protocol then really crosses xGMI (on a shared device "peer" memory is local HBM).  Skips on a 1-GPU box."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % _n_gpus())
    two_rank_case(mode, algo, multi_gpu=True)


def two_rank_case(mode, algo, multi_gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + ALGOS.index(algo) * 2 + (0 if mode == "peer" else 1) + (20 if multi_gpu else 0)
    procs = [ctx.Process(target=worker, args=(r, 2, port, mode, algo, q, None, multi_gpu)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=300) for _ in range(2))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    one, (feats, ids, y) = single_process(algo)
    ref = oracle_two_steps(algo, feats, ids, y)
    for rank in (0, 1):
        res = got[rank]
        assert res["status"] == 0, "a peer wait timed out"
        assert res["batch_total"] == B
        if mode == "peer":
            assert res["peer"], "the hipIpc exchange path was not taken (PeerComm.create fell back)"
        else:
            assert not res["peer"]
        # the ranks are replicas: bitwise equal to each other
        assert np.array_equal(res["params"], got[0]["params"])
        if res["aux"] is not None:
            assert np.array_equal(res["aux"], got[0]["aux"])
        # equal to the single-process step up to the summation order of the shards
        gmax = np.abs(one["params"]).max()
        ok = np.ones(one["params"].shape, bool)
        if algo == "dla":
            ok = np.abs(one["grads"]) > 1e-4 * np.abs(one["grads"]).max()
            assert ok.mean() > 0.9
        np.testing.assert_allclose(res["params"][ok], one["params"][ok], rtol=2e-5, atol=2e-6 * gmax, err_msg="vs single process")
        np.testing.assert_allclose(res["losses"], one["losses"], rtol=1e-5, atol=1e-6)
        if res["state"] is not None:
            np.testing.assert_allclose(res["state"], one["state"], rtol=2e-4, atol=1e-9)
        if res["aux"] is not None:
            np.testing.assert_allclose(res["aux"], one["aux"], atol=2e-6)
    assert got[0]["rng_seed"] != got[1]["rng_seed"]  # RegressionEM's device draw is keyed per rank
    # and to the oracle (the first step is teacher-forced identical; the second starts from slightly different params)
    res = got[0]
    assert abs(res["losses"][-1] - ref["loss"]) <= 2e-5 * max(1.0, abs(ref["loss"]))
    sel = np.abs(ref["params"] - initial_state(algo)[0]) > 1e-4  # parameters that actually moved
    if algo == "dla":
        sel &= np.abs(one["grads"]) > 1e-4 * np.abs(one["grads"]).max()
    np.testing.assert_allclose(res["params"][sel], ref["params"][sel], rtol=2e-3, atol=2e-4)
    if res["aux"] is not None:
        np.testing.assert_allclose(res["aux"], ref["aux"], atol=1e-5)


@pytest.mark.parametrize("world,algo", [(4, "softmax"), (8, "softmax"), (8, "pairdebias")])
def test_more_physical_gpus_peer_exchange(world, algo):
    """4 / 8 ranks, one physical GPU each (the driver's scaling run goes to 8).  Skips when the node has fewer GPUs."""
    if _n_gpus() < world:
        pytest.skip("needs >= %d GPUs (found %d)" % (world, _n_gpus()))
    more_ranks_case(min(world, B), algo, multi_gpu=True)


@pytest.mark.parametrize("world,algo", [(4, "softmax"), (4, "pairdebias"), (7, "softmax")])
def test_more_ranks_peer_exchange(world, algo):
    more_ranks_case(world, algo, multi_gpu=False)


def more_ranks_case(world, algo, multi_gpu):
    """The exchange kernel is compiled per world size (2 .. 8 ranks: flag matrix [slice][rank], rank-ordered sums): 4 ranks
    (shards of 2, 2, 2, 1 lists) and 7 ranks (one list each - the scaling run goes to 8) on the one GPU, against the
    single-process step; PairDebias also carries the agreed global batch through uneven shards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29760 + world + (10 if algo == "pairdebias" else 0) + (20 if multi_gpu else 0)
    procs = [ctx.Process(target=worker, args=(r, world, port, "peer", algo, q, None, multi_gpu)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=600) for _ in range(world))
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    one, _ = single_process(algo)
    gmax = np.abs(one["params"]).max()
    for rank in range(world):
        res = got[rank]
        assert res["status"] == 0 and res["peer"] and res["batch_total"] == B
        assert np.array_equal(res["params"], got[0]["params"])
        np.testing.assert_allclose(res["params"], one["params"], rtol=2e-5, atol=2e-6 * gmax)
        np.testing.assert_allclose(res["losses"], one["losses"], rtol=1e-5, atol=1e-6)
        if res["aux"] is not None:
            np.testing.assert_allclose(res["aux"], one["aux"], atol=2e-6)


@pytest.mark.parametrize("mode", ["peer", "pg"])
def test_forced_data_parallel_world1_equals_plain_step(mode):
    """ULTR_FORCE_DP=1 with one rank (what profiles/r02_cfg2_forced_dp_world1_bench.json measures): the data-parallel step -
    ONE C call that queues backward -> exchange kernel -> update (peer), or backward | all-reduce | sumsq | update (pg) - must
    leave exactly the parameters of the plain single-GPU step (the exchange with W = 1 is a copy; same kernels otherwise)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=worker, args=(0, 1, 29790 + (0 if mode == "peer" else 1), mode, "softmax", q, {"ULTR_FORCE_DP": "1"}))
    proc.start()
    rank, res = q.get(timeout=300)
    proc.join(120)
    assert proc.exitcode == 0 and rank == 0
    one, _ = single_process("softmax")
    assert res["status"] == 0 and res["peer"] == (mode == "peer") and res["batch_total"] == B
    np.testing.assert_allclose(res["losses"], one["losses"], rtol=1e-6, atol=1e-7)
    gmax = np.abs(one["params"]).max()
    np.testing.assert_allclose(res["params"], one["params"], rtol=2e-6, atol=2e-7 * gmax)
    np.testing.assert_allclose(res["state"], one["state"], rtol=2e-5, atol=1e-10)


def test_peer_comm_world1_matches_grad_sumsq():
    """world 1 through the exchange kernel (ULTR_FORCE_DP): copy + partials == ultr_grad_sumsq, BIT FOR BIT (ranks of one step may
    take different exchange kernels; a one-ulp difference in a partial would be a different clip coefficient per rank)."""
    import ctypes
    import torch
    from ultra_pytorch_amd import _lib, hip_ops
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    P, Ls = 5000, 10
    tail = hip_ops.tail_floats(Ls)
    g = torch.randn(P + tail, device=dev)
    h = ctypes.c_void_p()
    _lib.check(lib.ultr_comm_create(0, 1, P + tail, ctypes.byref(h)), "create")
    out = torch.empty_like(g)
    nsq = (P + tail + 63) // 64
    ws = torch.zeros(nsq + 8, device=dev)
    ws2 = torch.zeros(nsq + 8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ultr_comm_allreduce(h, 0, ctypes.c_void_p(g.data_ptr()), P + tail, P, ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(ws.data_ptr()), nsq, st), "allreduce")
    hip_ops.grad_sumsq(g, P, Ls, ws2)
    assert lib.ultr_comm_status(h, st) == 0
    assert torch.equal(out, g)
    assert torch.equal(ws[:nsq], ws2[:nsq])
    lib.ultr_comm_destroy(h)


def timeout_worker(rank, port, q):
    """rank 0 runs a data-parallel step, rank 1 never publishes: rank 0's peer wait must time out (bounded, 3 s), leave the
    parameters untouched (the update behind the exchange is guarded by the status word), make read_loss() raise, and raise the
    status word on rank 1 as well."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                      ULTR_DP_COMM="peer", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from ultra_pytorch_amd import _lib, engine, hip_ops, parallel
    torch.cuda.set_device(0)
    _, _, _, pg = parallel.init_process_group_from_env(backend="gloo")
    dev = torch.device("cuda", 0)
    feed, names_d, names_l, _ = make_global(3)
    local = parallel.shard_input_feed(feed, "letor_features", names_d, names_l, L, rank, 2)
    feats, ids, y = feed_arrays(local, names_d, names_l)
    params0, state0, _, _ = initial_state("softmax")
    shape = hip_ops.DnnShape(F, HIDDEN, "elu")
    eng = engine.StepEngine(shape, ids.shape[1], L, dev, algo="softmax", process_group=pg)
    assert eng.comm is not None
    p, st = torch.tensor(params0, device=dev), torch.tensor(state0, device=dev)
    res = {"rank": rank}
    dist.barrier()
    if rank == 0:
        eng.train_step(p, st, torch.tensor(feats, device=dev), feats.shape[0], torch.tensor(ids, device=dev),
                       torch.tensor(y, device=dev), ipw_table=torch.linspace(1.0, 3.0, 4, device=dev))
        try:
            eng.read_loss()
            res["raised"] = False
        except _lib.UltrHipError as ex:
            res["raised"] = "timed out" in str(ex)
        torch.cuda.synchronize()
        res["params_unchanged"] = bool(np.array_equal(p.cpu().numpy(), params0))
        res["state_unchanged"] = bool(np.array_equal(st.cpu().numpy(), state0))
        res["status"] = eng.comm.status()
    dist.barrier()  # rank 1 waits here while rank 0 times out
    if rank == 1:
        res["status"] = eng.comm.status()  # raised remotely by rank 0
    q.put(res)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_comm_timeout_freezes_the_update_and_raises():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=timeout_worker, args=(r, 29795, q)) for r in range(2)]
    [p.start() for p in procs]
    got = {r["rank"]: r for r in (q.get(timeout=300) for _ in range(2))}
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[0]["raised"] is True and got[0]["params_unchanged"] and got[0]["state_unchanged"]
    assert got[0]["status"] == -4 and got[1]["status"] == -4  # ULTR_E_COMM_TIMEOUT on BOTH ranks


# ---- round 6 -----------------------------------------------------------------------------------------------------------------------
def _skew_worker(rank, world, port, q, shards, n_steps, delays, multi_gpu=False):
    """n_steps product steps on this rank's shard of a fixed global batch, every step delayed by a busy-wait kernel of this rank's
    own (seeded) length in front of it: the rank's publish arrives that much later than its peers'."""
    sys.path.insert(0, ROOT)
    local = rank if multi_gpu else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local),
                      ULTR_DP_COMM="peer", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from ultra_pytorch_amd import engine, hip_ops, parallel, synthetic
    from oracle import ultr_oracle as O
    torch.cuda.set_device(local)
    _, _, _, pg = parallel.init_process_group_from_env(backend="nccl" if multi_gpu else "gloo")
    dev = torch.device("cuda", local)
    Ls = 10
    Bl = shards[rank]
    rng = np.random.RandomState(100 + rank)
    feats, ids, y = synthetic.make_batch(rng, Bl, Ls, F)
    shape = hip_ops.DnnShape(F, HIDDEN, "elu")
    eng = engine.StepEngine(shape, Bl, Ls, dev, algo="softmax", process_group=pg, batch_total=int(sum(shards)))
    p = torch.tensor(O.init_params(F, HIDDEN, seed=5), device=dev)
    st = torch.zeros_like(p)
    f, i_, yy = torch.tensor(feats, device=dev), torch.tensor(ids, device=dev), torch.tensor(y, device=dev)
    ipw = torch.linspace(1.0, 3.0, 10, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    torch.cuda._sleep(1000000)
    ev1.record()
    torch.cuda.synchronize()
    cyc_per_us = 1000000.0 / (1e3 * ev0.elapsed_time(ev1))
    rs = np.random.RandomState(9000 + rank)
    err = None
    try:
        for k in range(n_steps):
            d = int(rs.randint(0, 51)) if delays else 0
            if d > 0:
                torch.cuda._sleep(int(d * cyc_per_us))
            eng.train_step(p, st, f, feats.shape[0], i_, yy, ipw_table=ipw)
            if (k & 7) == 7:
                eng.read_scalars()
        eng.read_scalars()
    except Exception as ex:
        err = repr(ex)
    torch.cuda.synchronize()
    status = -1 if eng.comm is None else int(eng.comm.status())
    q.put((rank, dict(params=p.cpu().numpy(), state=st.cpu().numpy(), status=status, err=err, peer=eng.comm is not None)))
    if eng.comm is not None:
        eng.comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _skew_case(world, shards, n_steps, delays, port, multi_gpu=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_skew_worker, args=(r, world, port, q, shards, n_steps, delays, multi_gpu)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=600) for _ in range(world))
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r in range(world):
        assert got[r]["err"] is None, got[r]["err"]
        assert got[r]["peer"] and got[r]["status"] == 0, "rank %d: status %r" % (r, got[r]["status"])
        assert np.isfinite(got[r]["params"]).all()
        assert np.array_equal(got[r]["params"], got[0]["params"]), "rank %d's replica differs from rank 0's" % r
        assert np.array_equal(got[r]["state"], got[0]["state"])
    return got


@pytest.mark.parametrize("world", [2, 4, 7])
def test_exchange_protocol_under_skew(world):
    """The publish / flag / peer-read protocol of the one-kernel exchange with every rank late by its own 0 - 50 us per step (a
    busy-wait kernel in front of the step), 300 steps, ranks sharing the one GPU: replicas stay bit-identical, no status word is
    raised.  (bench.py --gpus N runs the same self-test at config 2 in front of its timed region: `dp_selftest` on its line.)"""
    _skew_case(world, [3 + (r % 2) for r in range(world)], 300, True, 29820 + world)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_protocol_under_skew_physical_gpus(world):
    if _n_gpus() < world:
        pytest.skip("needs >= %d GPUs (found %d)" % (world, _n_gpus()))
    _skew_case(world, [3 + (r % 2) for r in range(world)], 300, True, 29840 + world, multi_gpu=True)


def test_mixed_exchange_paths_keep_replicas_bitwise_equal():
    """with unequal local batches one rank's slab reduction runs the exchange itself and hands level-2
    sum-of-squares partials to the update, while the other - more than 1024 loss partials: a two-level loss fold - takes the
    stand-alone exchange and sums level-1 partials.  Both sums now associate identically (groups of four in order, then strided),
    so the gradient norm, the clip coefficient and the replicas keep the same bits: 6 000 lists against 40, five steps."""
    _skew_case(2, [6000, 40], 5, False, 29860)
