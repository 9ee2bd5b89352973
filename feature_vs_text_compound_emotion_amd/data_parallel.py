"""Data parallelism over clips: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code at all; clips are
independent, so the path shards over clips with ONE exchange per step: the sum
of the trainable-parameter gradients (2.2-5.0 M floats = 9-20 MB, latency
bound on the xGMI mesh).  All gradients live in a single flat fp32 bucket:
``p.grad`` of every trainable parameter is a view into it, autograd accumulates
into the views in place, one ``all_reduce`` (RCCL picks direct/tree for this
size) covers the whole model, and the optimiser reads the same views -- no
flatten / unflatten copies.  The frozen encoders are replicated and exchange
nothing.

Semantics: each rank equals the single-process reference run on its shard of
the global batch, and the applied gradient is the mean over ranks
(== the gradient of the mean loss over the global batch when every rank holds
the same number of frames).  BatchNorm statistics stay local to the rank, like
torch DDP without SyncBN.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None, single_rank_group=False):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from the launcher (torch.distributed.run).  ``single_rank_group``:
    create the process group for WORLD_SIZE == 1 as well (a one-rank RCCL communicator is legal; the tests use it to drive
    the collectives' stream semantics on one GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class ClipDataParallel:
    """Gradient bucket + collectives for a model whose trainable part is small."""

    def __init__(self, model, world_size=None, broadcast=True, overlap=False, bucket_mb=25.0):
        """``overlap``: cut the flat bucket into slices of ``bucket_mb`` (in parameter order) and start the all-reduce of a
        slice from an autograd hook as soon as the last gradient of the slice has been accumulated -- backward produces
        the gradients from the top of the model down, so with released encoder units (28-43 M parameters, 112-172 MB) the
        exchange of the tail's and the upper units' gradients runs under the backward of the units below.  Needs ONE
        ``backward()`` per ``zero_grad()`` (the reference's training loop); off by default."""
        self.model = model
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("model has no trainable parameter")
        dev = self.params[0].device
        total = (sum(p.numel() for p in self.params) + 3) // 4 * 4  # the fused optimiser works on float4
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        if broadcast and self.world > 1:
            self.broadcast_state()
        # independent dropout streams per rank: the models draw their masks from (dropout_seed, counter), and every rank
        # starts dropout_seed at 0 -- fold the rank in (2^20 steps apart)
        if self.world > 1 and dist.is_initialized() and hasattr(model, "dropout_seed"):
            model.dropout_seed += dist.get_rank() << 20
        # ``overlap="force"`` keeps the hook-driven slices on with ONE rank too (a single-rank RCCL communicator is legal):
        # the path the first multi-GPU run will take, exercised on one GPU (tests/test_rccl_single_rank_gpu.py)
        self.overlap = bool(overlap) and dist.is_initialized() and (self.world > 1 or overlap == "force")
        self._gathered = True          # p.grad are bucket views right now
        self.buckets, self._works = [], []
        if self.overlap:
            self._build_buckets(bucket_mb)

    # ------------------------------------------------------------------ bucketed, overlapped exchange
    def _build_buckets(self, bucket_mb):
        cap = max(1, int(bucket_mb * (1 << 20)) // 4)
        start = off = 0
        count = 0
        index_of = []
        for p in self.params:
            if count and off + p.numel() - start > cap:
                self.buckets.append([start, off, count])
                start, count = off, 0
            index_of.append(len(self.buckets))
            off += p.numel()
            count += 1
        self.buckets.append([start, off, count])
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        for p, b in zip(self.params, index_of):
            p.register_post_accumulate_grad_hook(self._make_hook(b))

    def _make_hook(self, b):
        def hook(_param):
            self._pending[b] -= 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        """Collectives are issued STRICTLY in reverse slice order (last slice first, what a top-down backward completes
        first anyway): slice b goes out only when every slice above it has.  The issue order is then the same on every rank
        even when a parameter gets no gradient on one of them (a data-dependent branch, a modality absent from a shard) --
        that rank's slice, and everything below it, simply waits for ``all_reduce_gradients`` -- instead of following each
        rank's own autograd firing order, which would pair differently sized collectives across ranks."""
        while self._next >= 0 and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next -= 1

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def broadcast_state(self, src=0):
        """Replicate rank ``src``'s parameters and buffers (one flat message each).  The copies go through the tensors
        themselves (not ``.data``), so their version counters move and every packed-weight cache keyed on
        (data_ptr, _version) -- IR50.pack*, the VGGish / BERT caches -- is rebuilt on the next forward."""
        for tensors in (list(self.model.parameters()), [b for b in self.model.buffers() if b.dtype == torch.float32]):
            if not tensors:
                continue
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            dist.broadcast(flat, src)
            off = 0
            with torch.no_grad():
                for t in tensors:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def sync_buffers(self, mode="mean", force=False):
        """BatchNorm running statistics are rank-local during training (each rank == the reference on its shard).  Before
        a checkpoint or an evaluation make them one set again: ``mean`` averages the float buffers over the ranks (every
        shard's statistics count), ``broadcast`` takes rank 0's.  Integer buffers (num_batches_tracked) are equal already."""
        if self.world == 1 and not (force and dist.is_initialized()):   # ``force``: run the collective with one rank too (tests)
            return
        bufs = [b for b in self.model.buffers() if b.dtype == torch.float32]
        if not bufs:
            return
        flat = torch.cat([b.detach().reshape(-1) for b in bufs])
        if mode == "mean":
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.world)
        elif mode == "broadcast":
            dist.broadcast(flat, 0)
        else:
            raise ValueError(mode)
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()

    def zero_grad(self):
        """Overlapped exchange: the bucket is zeroed and every ``p.grad`` is (again) a view into it, so autograd accumulates in
        place and the slice hooks see the gradients land.  Otherwise (round 3): ``p.grad = None`` -- autograd then ADOPTS the
        gradient tensor a backward function returns instead of launching one ``grad += g`` kernel per parameter (102 launches
        of ~2 us work each per step for the tri-modal LFAN tail), and ``gather_gradients()`` moves them into the bucket with a
        few multi-tensor copies before the all-reduce / the fused optimiser reads it."""
        if not self.overlap:
            for p in self.params:
                p.grad = None
            self._gathered = False
            return
        self.flat.zero_()
        self._pending = [b[2] for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._works = []
        self._gathered = True
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def gather_gradients(self):
        """Bring every parameter's gradient into the flat bucket (a no-op when they already live there) and make ``p.grad`` the
        bucket views again; parameters that received no gradient read as zeros.  Idempotent per ``zero_grad()``."""
        if getattr(self, "_gathered", True):
            return
        base = self.flat.data_ptr()
        dsts, srcs, zeros, off = [], [], [], 0
        views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            views.append(v)
            g = p.grad
            if g is None:
                zeros.append(v)
            elif g.data_ptr() != base + 4 * off:
                dsts.append(v)
                srcs.append(g if g.dtype == torch.float32 else g.float())
            off += p.numel()
        if off < self.flat.numel():
            zeros.append(self.flat[off:])
        with torch.no_grad():
            if dsts:
                torch._foreach_copy_(dsts, srcs)
            if zeros:
                torch._foreach_zero_(zeros)
        for p, v in zip(self.params, views):
            p.grad = v
        self._gathered = True

    def all_reduce_gradients(self):
        """Mean of the gradients over ranks, in place in the bucket."""
        self.gather_gradients()
        if self.world > 1 or self.overlap:
            if self.overlap:
                while self._next >= 0:   # slices at / below one whose parameters got no gradient this step, same order
                    self._launch(self._next)
                    self._next -= 1
                for w in self._works:
                    w.wait()
                self._works = []
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)

    def flatten_parameters(self):
        """Re-home every trainable parameter in ONE flat fp32 buffer (same order and padding as the gradient bucket) so
        that the optimiser update is a single launch.  ``p.data`` becomes a view; values are preserved."""
        if getattr(self, "flat_param", None) is None:
            self.flat_param = torch.empty_like(self.flat)
            off = 0
            for p in self.params:
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += p.numel()
            self.flat_param[off:].zero_()
        return self.flat_param

    def shard(self, global_batch_indices, rank):
        """Distributed-sampler rule: rank r takes clips r::world of every global batch."""
        return global_batch_indices[rank::self.world]


class FlatNesterovSGD:
    """torch.optim.SGD(momentum, nesterov, weight_decay) of the reference (instantiators.py:74-92) as ONE HIP launch over
    the flat parameter / gradient / momentum buffers of a ``ClipDataParallel`` (bit-identical arithmetic, tested against
    torch.optim.SGD).  ``param_groups[0]['lr']`` is what the reference's scheduler mutates (base/scheduler.py:167-197)."""

    def __init__(self, ddp, lr=1e-3, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True):
        self.ddp = ddp
        self.param_groups = [{"params": ddp.params, "lr": lr, "momentum": momentum, "dampening": dampening,
                              "weight_decay": weight_decay, "nesterov": nesterov}]
        self.flat_param = ddp.flatten_parameters()
        self.buf = torch.zeros_like(self.flat_param)
        self.steps = 0

    def zero_grad(self, set_to_none=False):
        self.ddp.zero_grad()

    def step(self):
        from . import ops
        self.ddp.gather_gradients()
        g = self.param_groups[0]
        ops.sgd_nesterov_flat(self.flat_param, self.ddp.flat, self.buf, g["lr"], g["momentum"], g["dampening"],
                              g["weight_decay"], g["nesterov"], first_step=self.steps == 0)
        self.steps += 1
        # the kernel wrote through raw pointers: move the version counters like an in-place torch op would, so caches keyed
        # on (data_ptr, _version) see the update.  Semantics note: every trainable parameter always has a (dense) gradient
        # view in the bucket, so weight decay and momentum apply to all of them every step -- torch.optim.SGD skips a
        # parameter whose grad is None after zero_grad(set_to_none=True); the two only differ for a parameter that
        # receives no gradient at all in a step, which the models here do not have.
        torch.autograd.graph.increment_version(self.ddp.params)

    def state_dict(self):
        return {"momentum_buffer": self.buf, "steps": self.steps, "param_groups": [{k: v for k, v in g.items() if k != "params"}
                                                                                   for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.buf.copy_(sd["momentum_buffer"])
        self.steps = int(sd["steps"])
        for g, h in zip(self.param_groups, sd["param_groups"]):
            g.update(h)
