"""The N>1 data-parallel path on CPU: 2 gloo ranks, flat gradient bucket, mean all-reduce."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, init_process_group_from_env
    import torch.distributed as dist
    r, w, _ = init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must make them equal
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    model[0].weight.requires_grad = False  # a frozen part, like the encoders
    ddp = ClipDataParallel(model)
    g = torch.Generator().manual_seed(7)
    xs, ys = torch.randn(8, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    idx = ddp.shard(list(range(8)), rank)
    opt = torch.optim.SGD(ddp.params, momentum=0.9, nesterov=True, weight_decay=1e-4)
    for _ in range(2):
        ddp.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(xs[idx]), ys[idx])
        loss.backward()
        ddp.all_reduce_gradients()
        lo, hi = ddp.flat.data_ptr(), ddp.flat.data_ptr() + 4 * ddp.flat.numel()
        assert all(lo <= p.grad.data_ptr() < hi for p in ddp.params)  # views of the bucket (gathered there before the exchange)
        opt.step()
    out[rank] = {k: v.clone() for k, v in model.state_dict().items() if "running" not in k and "num_batches" not in k}
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_stay_in_lockstep_and_average_gradients():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        a, b = out[0], out[1]
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_single_process_bucket_views_and_zero_grad():
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel
    model = torch.nn.Linear(4, 2)
    ddp = ClipDataParallel(model, world_size=1)
    model(torch.ones(3, 4)).sum().backward()
    assert ddp.flat.abs().sum() > 0
    assert model.weight.grad.data_ptr() == ddp.flat.data_ptr()       # first step: the views set up by the constructor
    # zero_grad drops the gradients (autograd then adopts what backward returns: no per-parameter add kernels) ...
    ddp.zero_grad()
    assert model.weight.grad is None and model.bias.grad is None
    (model(torch.ones(3, 4))[:, 0] * 0 + model.weight.sum()).sum().backward()     # a step in which the bias gets NO gradient
    assert model.weight.grad.data_ptr() != ddp.flat.data_ptr()
    # ... and gather_gradients() brings them into the bucket (zeros for a parameter without gradient, stale values gone)
    ddp.all_reduce_gradients()
    assert model.weight.grad.data_ptr() == ddp.flat.data_ptr() and torch.equal(model.weight.grad, torch.full((2, 4), 3.0))
    assert model.bias.grad.abs().sum() == 0 and ddp.flat[8:].abs().sum() == 0
    ddp.gather_gradients()                                            # idempotent
    assert torch.equal(model.weight.grad, torch.full((2, 4), 3.0))


def _worker_overlap(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, init_process_group_from_env
    import torch.distributed as dist
    init_process_group_from_env(backend="gloo")
    res = {}
    for overlap in (False, True):
        torch.manual_seed(5)
        model = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                                    torch.nn.Linear(64, 32), torch.nn.Linear(32, 3), torch.nn.Linear(3, 3))
        model[6].weight.requires_grad = False        # frozen in the middle of a slice
        for p in model[4].parameters():              # a whole slice that never receives a gradient (used under no_grad below)
            p.requires_grad = True
        # 8 KiB slices: the 12 trainable tensors fall into several buckets; the last ones fill first during backward
        ddp = ClipDataParallel(model, overlap=overlap, bucket_mb=8 / 1024)
        if overlap:
            assert len(ddp.buckets) >= 3 and ddp.buckets[0][0] == 0 and ddp.buckets[-1][1] == sum(p.numel() for p in ddp.params)
            assert all(a[1] == b[0] for a, b in zip(ddp.buckets, ddp.buckets[1:]))
        g = torch.Generator().manual_seed(11 + rank)
        flats = []
        for step in range(2):
            ddp.zero_grad()
            x = torch.randn(6, 40, generator=g)
            h = model[3](model[2](model[1](model[0](x))))
            with torch.no_grad():
                skip = model[4](h)                       # model[4] is outside the graph: its slice gets no gradient
            y = model[6](model[5](skip + 0 * h[:, :32]))
            y.square().mean().backward()
            ddp.all_reduce_gradients()
            flats.append(ddp.flat.clone())
        res[overlap] = flats
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_exchange_equals_the_single_all_reduce():
    """ClipDataParallel(overlap=True): slices of the flat bucket are all-reduced from autograd hooks as their last gradient
    arrives (reverse order of the parameters), slices without any gradient at all_reduce_gradients(); the result is the same
    mean as ONE all-reduce of the whole bucket, and both ranks hold it."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
        for rank in (0, 1):
            for a, b in zip(out[rank][False], out[rank][True]):
                assert torch.allclose(a, b, rtol=0, atol=1e-7) and a.abs().sum() > 0
        for a, b in zip(out[0][True], out[1][True]):
            assert torch.equal(a, b)


def _worker_rank_dependent_graph(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, init_process_group_from_env
    import torch.distributed as dist
    init_process_group_from_env(backend="gloo")
    res = {}
    for overlap in (False, True):
        torch.manual_seed(9)
        model = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 48),
                                    torch.nn.Linear(48, 16), torch.nn.Linear(16, 3))
        ddp = ClipDataParallel(model, overlap=overlap, bucket_mb=8 / 1024)
        if overlap:
            assert len(ddp.buckets) >= 4
        g = torch.Generator().manual_seed(3 + rank)
        ddp.zero_grad()
        x = torch.randn(6, 40, generator=g)
        h = model[1](model[0](x))
        if rank == 1:                     # a data-dependent branch: rank 1 skips a layer in the MIDDLE of the model this step
            h = h[:, :48]
        else:
            h = model[2](h)
        model[4](model[3](h)).square().mean().backward()
        ddp.all_reduce_gradients()
        res[overlap] = ddp.flat.clone()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_keeps_one_collective_order_when_a_rank_skips_a_layer():
    """ADVICE (round 2): a slice whose parameters get no gradient on ONE rank must not change the order in which that rank
    issues its collectives (differently sized all-reduces would pair up across ranks: a gloo size error, a silent
    mis-reduction or a hang on RCCL).  Slices go out strictly last-to-first; the skipped one and everything below it wait for
    all_reduce_gradients() on the rank that skipped.  Result == the single all-reduce, identical on both ranks."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_rank_dependent_graph, args=(world, port, out), nprocs=world, join=True)
        for rank in (0, 1):
            assert torch.allclose(out[rank][False], out[rank][True], rtol=0, atol=1e-7) and out[rank][True].abs().sum() > 0
        assert torch.equal(out[0][True], out[1][True])
