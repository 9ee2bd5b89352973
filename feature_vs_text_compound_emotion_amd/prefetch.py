"""Pinned staging + prefetch for the on-GPU input pipeline.

The reference feeds its trainer with ``DataLoader(num_workers=6, pin_memory=True)`` (base/experiment.py:205-211): six worker
PROCESSES run PIL resize / crop / flip per frame and the main process copies the float batch to the device synchronously
(trainer.py:351-352).  Here the per-frame work already runs on the GPU (``frames.FrameTransform``, bit-identical to PIL), so
what the host has to do per batch is: read the window's raw rows from the memory-mapped ``.npy`` files, pack them into
PINNED staging buffers, and start the transfer.  ``DevicePrefetcher`` does that ``depth`` batches ahead:

    worker threads (numpy releases the GIL in its copies)  ->  one pinned slot per in-flight batch (allocated once, reused)
    ->  ``copy_(non_blocking=True)`` on a dedicated HIP stream  ->  the frame-transform kernel on that same stream
    ->  an event the consumer's stream waits on (no host synchronisation anywhere).

A slot is handed back to the producer only after the consumer has asked for the NEXT batch and the transfer out of it has
completed (event), so the model never sees a buffer that is being refilled.  uint8 frames cross PCIe (3 B per pixel of the
256x256 source, 0.2 MB per frame) instead of the reference's float32 crops after the CPU transform -- the 4-byte crops are
smaller (19 KB per 40x40 frame), but producing them costs the reference ~0.8 ms of PIL work per frame and process.
"""
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class DevicePrefetcher:
    """Iterate ``(inputs, trials, lengths, indices)`` batches of a ``TrialDataset``-like dataset, resident on ``device``.

    ``batches``: list of index lists (the sampler's output for one epoch).  ``frame_transform``: a ``FrameTransform`` (its
    crop / flip draws are made here, in batch order) or None.  The draws come from the prefetcher's OWN ``random.Random``
    stream, seeded at construction on the caller's thread from Python's global generator (``seed`` overrides): the producer
    thread runs ``depth`` batches ahead of the main thread, so drawing from the shared global stream would interleave with
    the main thread's own uses of ``random`` in a timing-dependent order and a seeded run would not reproduce its crops.
    """

    def __init__(self, dataset, batches, device="cuda", frame_transform=None, num_workers=6, depth=2, seed=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePrefetcher stages batches for the GPU path: no CUDA/HIP device")
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dataset, self.batches, self.device = dataset, [list(b) for b in batches], dev
        self.transform, self.depth = frame_transform, max(1, int(depth))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(num_workers)))
        self.stream = torch.cuda.Stream(device=self.device)
        self.ready = queue.Queue()
        self.free = queue.Queue()
        for s in range(self.depth):
            self.free.put({"id": s, "pinned": {}, "released": None})
        self.error = None
        self._last = None
        self.rng = random.Random(random.getrandbits(64) if seed is None else seed)
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    # ------------------------------------------------------------------ producer side
    def _stage(self, slot, key, arrays):
        """Pack the per-clip arrays of one modality into the slot's pinned buffer (grown on demand) -> pinned view."""
        first = arrays[0]
        shape = (len(arrays),) + tuple(first.shape)
        dtype = first.dtype if torch.is_tensor(first) else torch.from_numpy(np.asarray(first)).dtype
        n = int(np.prod(shape))
        buf = slot["pinned"].get(key)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty((n,), dtype=dtype, pin_memory=True)
            slot["pinned"][key] = buf
        view = buf[:n].view(shape)
        for i, a in enumerate(arrays):
            view[i].copy_(a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)))
        return view

    def _produce(self):
        try:
            torch.cuda.set_device(self.device)
            for idxs in self.batches:
                slot = self.free.get()
                if slot is None:
                    return
                if slot["released"] is not None:
                    slot["released"].synchronize()           # the previous transfer out of this slot has completed
                items = list(self.pool.map(self.dataset.__getitem__, idxs))
                examples = [it[0] for it in items]
                crop = self.transform.draw(len(items), self.rng) if (self.transform is not None and "video" in examples[0]) else None
                out = {}
                with torch.cuda.stream(self.stream):
                    for k in examples[0]:
                        host = self._stage(slot, k, [e[k] for e in examples])
                        out[k] = host.to(self.device, non_blocking=True)
                    if crop is not None:
                        out["video"] = self.transform(out["video"], crop_xyf=crop)
                    done = torch.cuda.Event()
                    done.record(self.stream)
                slot["released"] = done
                meta = ([it[1] for it in items], torch.tensor([it[2] for it in items]),
                        torch.from_numpy(np.stack([np.asarray(it[3]) for it in items])))
                self.ready.put((slot, out, done, meta))
            self.ready.put(None)
        except BaseException as err:  # surfaces in the consumer
            self.error = err
            self.ready.put(None)

    # ------------------------------------------------------------------ consumer side
    def __iter__(self):
        return self

    def __next__(self):
        if self._last is not None:           # the consumer is done with the previous batch's slot
            self.free.put(self._last)
            self._last = None
        item = self.ready.get()
        if item is None:
            if self.error is not None:
                raise self.error
            raise StopIteration
        slot, out, done, (trials, lengths, indices) = item
        torch.cuda.current_stream(self.device).wait_event(done)   # stream-side wait: the host does not block
        for t in out.values():
            t.record_stream(torch.cuda.current_stream(self.device))
        self._last = slot
        return out, trials, lengths, indices

    def close(self):
        self.free.put(None)
        self.pool.shutdown(wait=False)
