"""world_size-2 tests of the data-parallel glue on CPU (gloo, 127.0.0.1): key all-gather order + replicated queue
(golden set G7: the reference run chunk-wise, tests/golden/g7_dp.npz), cross-rank key shuffle, bucketed gradient all-reduce."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vince_oracle as vo


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return dict(ret)


def _keys_case(rank, world):
    from vince_amd import dp
    from vince_amd.utils.queue_index import enqueue_segments
    K, D, B = 96, 8, 20
    queue = np.full((K, D), -1.0, np.float32)
    tail, full = 0, False
    tails = []
    for step in range(4):
        local = torch.full((B, D), float(step * 1000 + rank * B)) + torch.arange(B, dtype=torch.float32)[:, None]
        gathered = dp.gather_keys(local)
        assert gathered.shape == (world * B, D)
        segs, tail, wrapped = enqueue_segments(tail, gathered.shape[0], K)
        for dst, src, ln in segs:
            queue[dst:dst + ln] = gathered[src:src + ln].numpy()
        full = full or wrapped
        tails.append(tail)
    return queue, tails, full


def test_replicated_queue_identical_and_matches_oracle():
    out = run2(_keys_case)
    q0, t0, f0 = out[0]
    q1, t1, f1 = out[1]
    np.testing.assert_array_equal(q0, q1)
    assert t0 == t1 and f0 == f1
    # single-process emulation (G7): keys concatenated in rank order into the oracle queue
    oq = vo.OracleQueue(96, 8, init=-np.ones((96, 8), np.float32))
    for step in range(4):
        blocks = [np.full((20, 8), float(step * 1000 + r * 20), np.float32) + np.arange(20, dtype=np.float32)[:, None]
                  for r in range(2)]
        oq.enqueue(np.concatenate(blocks))
    np.testing.assert_array_equal(q0, oq.vectors)
    assert t0[-1] == oq.current_tail and f0 == oq.full


def _g7_case(rank, world):
    """Each rank contributes ITS rows of the reference's key block; the gathered block, the ring arithmetic and the
    replicated queue must reproduce what the reference's StorageQueue held after enqueueing the rank-ordered block."""
    from vince_amd import dp
    from vince_amd.utils.queue_index import enqueue_segments
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_dp.npz"))
    p = "w%d_" % world
    keys = g[p + "keys"]
    b = keys.shape[0] // world
    K = g[p + "queue_before"].shape[0]
    queue = g[p + "queue_before"].copy()
    gathered = dp.gather_keys(torch.from_numpy(keys[rank * b:(rank + 1) * b].copy()))
    segs, tail, wrapped = enqueue_segments(K - 5, gathered.shape[0], K)
    for dst, src, ln in segs:
        queue[dst:dst + ln] = gathered[src:src + ln].numpy()
    return queue, tail, bool(wrapped), gathered.numpy()


@pytest.mark.parametrize("world", [2, 8])
def test_g7_reference_chunkwise_emulation_keys_and_queue(world):
    """Golden set G7: the REFERENCE run chunk-wise to emulate `world` ranks (oracle/make_golden_full.py)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_dp.npz"))
    out = run2(_g7_case, world=world)
    p = "w%d_" % world
    for r in range(world):
        queue, tail, wrapped, gathered = out[r]
        np.testing.assert_array_equal(gathered, g[p + "keys"])                    # rank order
        np.testing.assert_array_equal(queue, g[p + "queue_after"])                # bit-exact: a copy, no arithmetic
        assert tail == int(g[p + "tail"]) and wrapped == bool(g[p + "full"])


def _shuffle_case(rank, world):
    from vince_amd import dp
    B = 6
    local = (torch.arange(B, dtype=torch.float32) + rank * B)[:, None].repeat(1, 3)   # global row id in every column
    perm = dp.global_permutation(world * B, step=7, seed=1)
    mine = dp.exchange_rows(local, perm)
    want = perm[rank * B:(rank + 1) * B].float()
    assert torch.equal(mine[:, 0], want), (mine[:, 0], want)
    # "encode" = identity; gather and un-permute -> natural global order on every rank
    nat = dp.unpermute_gathered(dp.gather_keys(mine), perm)
    assert torch.equal(nat[:, 0], torch.arange(world * B, dtype=torch.float32))
    return perm.tolist()


def test_cross_rank_key_shuffle_roundtrip():
    out = run2(_shuffle_case)
    assert out[0] == out[1]           # same permutation on every rank
    assert sorted(out[0]) == list(range(12))


def _reduce_case(rank, world):
    from vince_amd import dp
    n = 1000
    model = types.SimpleNamespace(_flat=torch.zeros(n), _flat_grad=torch.arange(n, dtype=torch.float32) * (rank + 1),
                                  _n_train=n, _stage_offsets={"layer1": 100, "layer2": 300, "layer3": 500, "layer4": 800},
                                  _bucket_events=None)
    red = dp.GradientReducer(model, (2, 2, 2, 2))
    red.reduce_after_backward()
    return model._flat_grad.clone()


def test_bucketed_gradient_allreduce_sums_every_element_once():
    out = run2(_reduce_case)
    want = torch.arange(1000, dtype=torch.float32) * 3
    assert torch.equal(out[0], want) and torch.equal(out[1], want)


def _reduce_bf16_case(rank, world):
    from vince_amd import dp
    n = 1000
    g = torch.Generator().manual_seed(5 + rank)
    grad = torch.randn(n, generator=g)
    model = types.SimpleNamespace(_flat=torch.zeros(n), _flat_grad=grad.clone(), _n_train=n,
                                  _stage_offsets={"layer1": 100, "layer2": 300, "layer3": 500, "layer4": 800}, _bucket_events=None)
    red = dp.GradientReducer(model, (2, 2, 2, 2), payload="bf16")
    red.reduce_after_backward()
    return grad, model._flat_grad.clone()


def test_bucketed_gradient_allreduce_bf16_payload_opt_in():
    """`dp_grad_payload="bf16"`: buckets travel as bfloat16 and land back in the fp32 buffer; every element
    summed once, to bf16 accuracy, identically on both ranks."""
    out = run2(_reduce_bf16_case)
    want = out[0][0] + out[1][0]
    assert torch.equal(out[0][1], out[1][1])
    assert float((out[0][1] - want).abs().max()) <= 2e-2 * float(want.abs().max())
    assert float((out[0][1] - want).abs().max()) > 0          # (it really went through bf16)


def _coin_case(rank, world):
    """with one process per GPU every rank must pick the same jigsaw side each step."""
    import random
    from vince_amd.solvers.vince_solver import VinceSolver
    random.seed(1234 + rank)   # the per-process RNG differs between ranks on purpose
    stub = types.SimpleNamespace(args=types.SimpleNamespace(jigsaw=True), _jigsaw_rng=None)
    return [VinceSolver._jigsaw_coin(stub) < 0.5 for _ in range(64)]


def test_jigsaw_side_is_drawn_in_lockstep_across_ranks():
    out = run2(_coin_case)
    assert out[0] == out[1]
    assert 8 < sum(out[0]) < 56   # and it is still a coin


def _save_case(rank, world):
    import tempfile
    from vince_amd.solvers.vince_solver import VinceSolver
    d = os.path.join(tempfile.gettempdir(), "vince_save_case_%s" % os.environ["MASTER_PORT"])
    calls = []
    model = types.SimpleNamespace(save=lambda it, keep: calls.append((it, keep)))
    stub = types.SimpleNamespace(model=model, iteration=512, check_loss_latch=lambda **kw: None)
    VinceSolver.save(stub, 5)
    return calls


def _final_save_case(rank, world):
    """One rank on the exception path of a driver's `finally: solver.save()`, its peer still inside a collective: the final save must
    not hold a barrier.  Rank 1 'fails' and saves at once; rank 0 first finishes an all-reduce that rank 1 also joins
    AFTER its save -- with a barrier inside save() the two collectives would be mismatched and this would hang until the timeout."""
    import torch
    import torch.distributed as dist
    from vince_amd.solvers.vince_solver import VinceSolver
    calls = []
    model = types.SimpleNamespace(save=lambda it, keep: calls.append((it, keep)))
    stub = types.SimpleNamespace(model=model, iteration=7, check_loss_latch=lambda **kw: None)
    t = torch.ones(4)
    if rank == 1:
        VinceSolver.save(stub)          # the failing rank's `finally`
        dist.all_reduce(t)
    else:
        dist.all_reduce(t)              # the healthy rank is in its gradient all-reduce
        VinceSolver.save(stub)
    return calls, float(t[0])


def test_final_save_holds_no_collective():
    out = run2(_final_save_case)
    assert out[0] == ([(7, -1)], 2.0) and out[1] == ([], 2.0)


def test_only_rank_zero_writes_checkpoints():
    out = run2(_save_case)
    assert out[0] == [(512, 5)] and out[1] == []


def _latched_save_case(rank, world):
    """the periodic (sync) save reads the finite-loss latch BEFORE writing, and shares the verdict: a NaN seen on ONE rank must
    raise on every rank -- none left waiting in the save's barrier -- and nothing is written."""
    import torch
    from vince_amd.solvers.vince_solver import VinceSolver
    calls = []
    model = types.SimpleNamespace(save=lambda it, keep: calls.append((it, keep)), device=torch.device("cpu"))

    # the solver's own latch check on a device-side latch only rank 1 has tripped (first offending iteration 3, stored + 1)
    stub = types.SimpleNamespace(model=model, iteration=64, _loss_latch=torch.tensor([1 if rank == 1 else 0, 4 if rank == 1 else 0]))
    stub.check_loss_latch = lambda *a, **kw: VinceSolver.check_loss_latch(stub, *a, **kw)
    try:
        VinceSolver.save(stub, 5, sync=True)
        raised = None
    except AssertionError as e:
        raised = str(e)
    # the same shared verdict at the other two points every rank reaches together (log iterations, the epoch boundary)
    try:
        stub.check_loss_latch({"nce_loss": 1.0}, shared=True)
        raised_log = None
    except AssertionError as e:
        raised_log = str(e)
    return calls, raised, raised_log


def test_periodic_save_refuses_a_model_that_took_nan_steps_on_every_rank():
    out = run2(_latched_save_case)
    assert out[0][0] == [] and out[1][0] == []
    assert out[0][1] is not None and "another rank" in out[0][1]
    assert out[1][1] is not None and "first at iteration 3" in out[1][1]
    assert out[0][2] is not None and "another rank" in out[0][2]
    assert out[1][2] is not None and "first at iteration 3" in out[1][2]


def _samples_case(rank, world):
    from vince_amd.solvers.vince_solver import VinceSolver
    stub = types.SimpleNamespace(args=types.SimpleNamespace(batch_size=16))
    return VinceSolver.samples_per_step.fget(stub)


def test_iteration_counts_the_samples_of_all_ranks():
    """`iteration` counts SAMPLES (vince_solver.py:514, where batch_size is the batch of ALL GPUs behind
    nn.DataParallel); with a per-rank batch of 16 on two ranks a step consumes 32 -- so a data-parallel run and a single process at the
    same global batch agree on the sample counter, the epoch a checkpoint resumes into and the schedule position."""
    assert run2(_samples_case) == {0: 32, 1: 32}
