"""(a) Data-parallel equivalence on the REAL model: 2 freshly spawned ranks (gloo, both on the one GPU) with the HIP LFAN,
``ClipDataParallel`` and ``FlatNesterovSGD`` -- 2 ranks x B/2 clips must equal 1 rank x B clips on gradients and post-step
weights when the BatchNorms use running statistics ("N ranks x B/N == 1 rank x B with eval-mode BN").
(b) The reference trainer's AMP wrapper exactly as trainer.py:341,365-391 writes it -- ``GradScaler(enabled=True)``,
``autocast``, ``zero_grad(set_to_none=True)``, ``torch.optim.SGD`` -- around the HIP model."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

MODS = ["video", "vggish", "bert"]
B, L, HW = 4, 8, 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed=0, conditioned=False):
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=HW // 8, seed=seed, conditioned=conditioned)
    model = LFAN(backbone_settings={}, output_dim=7, task="CLASSIFICATION", modality=MODS, example_length=L, kernel_size=5,
                 tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cuda", head_hw=HW // 8)
    model.init(load_backbone=False)
    model.load_state_dict(sd, strict=True)
    return model.cuda()


def _steps(model, ddp, opt, x, labels, n_steps=2, train=False):
    """eval-mode BatchNorm / no dropout, but a real backward: every clip's gradient is independent of its batch mates.
    ``train``: model.train() with dropout off -- batch-statistics BatchNorm, local to the rank."""
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    model.eval()
    if train:
        model.train()
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        for net in model.temporal.values():
            net.dropout = 0.0
    grads = None
    for _ in range(n_steps):
        ddp.zero_grad()
        out = model({k: v for k, v in x.items()})
        loss = cross_entropy_loss(out, labels)
        loss.backward()
        ddp.all_reduce_gradients()
        if grads is None:
            grads = ddp.flat.clone()
        opt.step()
    return grads, ddp.flat_param.clone()


def _worker_train(rank, world, port, out):
    _worker(rank, world, port, out, train=True)


def _worker(rank, world, port, out, train=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.modules.setdefault("triton", None)
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD, init_process_group_from_env
    import torch.distributed as dist
    init_process_group_from_env(backend="gloo")
    torch.cuda.set_device(0)
    model = _model(seed=rank)          # different weights per rank: broadcast_state must make them rank 0's
    # the overlapped exchange: 2 MB slices of the 20 MB bucket, all-reduced from autograd hooks during the backward
    ddp = ClipDataParallel(model, overlap=True, bucket_mb=2.0)
    assert len(ddp.buckets) >= 5
    opt = FlatNesterovSGD(ddp, lr=1e-3)
    x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=55)
    idx = ddp.shard(list(range(B)), rank)
    xs = {k: v[idx].cuda() for k, v in x.items()}
    g, w = _steps(model, ddp, opt, xs, labels[idx].cuda(), n_steps=1 if train else 2, train=train)
    out[rank] = (g.cpu(), w.cpu())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_half_batches_equal_one_rank_on_the_full_batch():
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        (g0, w0), (g1, w1) = out[0], out[1]
    assert torch.equal(g0, g1) and torch.equal(w0, w1)      # the ranks hold the same reduced gradient and stay in lockstep
    model = _model(seed=0)
    ddp = ClipDataParallel(model, world_size=1)
    opt = FlatNesterovSGD(ddp, lr=1e-3)
    x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=55)
    g, w = _steps(model, ddp, opt, {k: v.cuda() for k, v in x.items()}, labels.cuda())
    gerr = (g.cpu() - g0).abs().max().item() / g.abs().max().item()
    werr = (w.cpu() - w0).abs().max().item()
    print(f"\n[dp] 2 ranks x {B // 2} clips vs 1 rank x {B} clips: relative gradient difference {gerr:.2e}, weight difference {werr:.2e}")
    assert gerr < 2e-5          # mean of two half-batch means vs one full-batch mean: fp32 summation order only
    assert werr < 1e-7


def test_train_mode_ranks_apply_the_mean_of_their_shards_gradients():
    """model.train() (batch-statistics BatchNorm in the encoder and the tail): the statistics are local to a rank -- each rank
    IS the single-process reference run on its shard (torch DDP without SyncBN) -- and the applied gradient is the mean over
    ranks.  So the all-reduced gradient of 2 ranks has to equal the mean of two single-process gradients, one per shard
    (NOT the gradient of one process on the full batch, whose BatchNorms would see other statistics)."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_train, args=(world, port, out), nprocs=world, join=True)
        (g0, w0), (g1, w1) = out[0], out[1]
    assert torch.equal(g0, g1) and torch.equal(w0, w1)
    x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=55)
    singles = []
    for r in range(world):
        model = _model(seed=0)
        ddp = ClipDataParallel(model, world_size=1)
        opt = FlatNesterovSGD(ddp, lr=1e-3)
        idx = list(range(B))[r::world]
        g, _ = _steps(model, ddp, opt, {k: v[idx].cuda() for k, v in x.items()}, labels[idx].cuda(), n_steps=1, train=True)
        singles.append(g.cpu())
    mean = (singles[0] + singles[1]) / 2
    err = (mean - g0).abs().max().item() / mean.abs().max().item()
    print(f"\n[dp train] all-reduced gradient vs the mean of the two single-shard gradients: {err:.2e}")
    assert err < 1e-6


def test_sync_buffers_and_broadcast_move_version_counters():
    """Caches keyed on (data_ptr, _version) must see the collectives' writes (single process: world 1 is a no-op for the
    collectives, so this checks the copy path through the tensors themselves)."""
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD
    model = _model()
    ddp = ClipDataParallel(model, world_size=1)
    opt = FlatNesterovSGD(ddp, lr=1e-3)
    v0 = [p._version for p in ddp.params]
    opt.step()
    assert all(p._version > a for p, a in zip(ddp.params, v0))


def _reference_amp_steps(model, x, labels, amp, n_steps=2):
    """trainer.py:341 (GradScaler), :365 (zero_grad(set_to_none=True)), :367-383 (autocast forward + CE on .long() labels),
    :389-391 (scale / step / update); optimizer as instantiators.py:74-79 builds it."""
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params=params, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True)
    scaler = torch.cuda.amp.GradScaler(enabled=amp)
    model.train()
    losses = []
    for _ in range(n_steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            outputs = model({k: v for k, v in x.items()})
            bsz, nfms, _ = labels.shape
            loss = cross_entropy_loss(outputs.contiguous().view(bsz * nfms, -1), labels.contiguous().view(bsz * nfms).long())
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    return losses, torch.cat([p.detach().reshape(-1) for p in params]).cpu()


def _flat_trainable(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad]).cpu()


@pytest.mark.parametrize("conditioned", [True, False])
def test_reference_amp_wrapper_drives_the_hip_model(conditioned):
    """Under autocast the encoder follows the reference onto fp16 arithmetic (narrow kernels); GradScaler's loss scaling
    passes through the hand-written backward (a power of two: exact) and the step equals the un-scaled step of the same
    model pinned to precision = "fp16".

    fp16 storage against the 2^-15-per-product encoder: on the well-conditioned problem (``conditioned=True``: video-TCN
    biases at the scale of the signal, synth.lfan_state_dict) the two-step parameter UPDATE agrees to < 10 % (storage-only
    emulation on the CPU: 2.5 %).  On the default draw this batch seed sits on a LeakyReLU switch of the video temporal net
    and the same 2e-3 relative embedding difference moves the update by ~40 % -- reproduced WITHOUT any kernel by
    tests/test_conditioning_cpu.py (storage-only emulation: 0.43; another batch seed, same weights: 0.009), so for that
    case the update is bounded by the emulation's value instead (it is a property of the problem, and it is the fp32
    reference's sensitivity too)."""
    from feature_vs_text_compound_emotion_amd import synth
    x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=56)
    xd, ld = {k: v.cuda() for k, v in x.items()}, labels.cuda()

    def fresh(precision):
        m = _model(conditioned=conditioned)
        m.spatial["visual"].backbone.precision = precision
        for mod in m.modules():                 # same dropout-free step on both sides
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        for net in m.temporal.values():
            net.dropout = 0.0
        return m
    w_init = _flat_trainable(fresh("bf16x3"))
    l_amp, w_amp = _reference_amp_steps(fresh("bf16x3"), xd, ld, amp=True)      # autocast switches the encoder to fp16
    l_f16, w_f16 = _reference_amp_steps(fresh("fp16"), xd, ld, amp=False)       # the same arithmetic without the wrapper
    l_ref, w_ref = _reference_amp_steps(fresh("bf16x3"), xd, ld, amp=False)     # full-precision step
    print(f"\n[amp conditioned={conditioned}] losses amp {l_amp} | fp16 no wrapper {l_f16} | bf16x3 {l_ref}")
    assert max(abs(a - b) for a, b in zip(l_amp, l_f16)) < 1e-6
    assert (w_amp - w_f16).abs().max().item() < 1e-7
    upd_amp, upd_ref = w_amp - w_init, w_ref - w_init
    rel = ((upd_amp - upd_ref).norm() / upd_ref.norm()).item()
    print(f"[amp conditioned={conditioned}] relative difference of the two-step update, fp16 autocast vs bf16x3: {rel:.3e}")
    assert rel < (0.1 if conditioned else 0.6)
    assert max(abs(a - b) for a, b in zip(l_amp, l_ref)) < 5e-3
    assert torch.isfinite(w_amp).all()
