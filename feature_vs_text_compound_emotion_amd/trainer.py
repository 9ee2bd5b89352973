"""Host loop around the hot path: one optimisation step, whole-video inference, window stitching.

Mirror of the reference's ``Trainer`` for the parts that call the model (trainer.py:315-434
``train_one_epoch``, :436-523 ``inference``, :611-770 ``optimize``, :788-892 ``window_input`` /
``inference_forward_windows``, :894-912 ``windowing``) plus the dataset-side window rule
(base/dataset.py:434-453, which uses ``>`` where the trainer uses ``>=``).

Two ways in:

* ``Trainer(**trainer_kwards)`` takes the keyword dictionary ``experiment.py:153-175`` builds (``models``, ``device``,
  ``model_name``, ``criterion``, ``train_batch_size`` ... -- unknown keys are kept as attributes like
  ``base/trainer.py:21-45,96-115`` does) and answers the calls ``experiment.py:178-214`` makes next: ``set_args``,
  ``post_set_args``, ``set_number_classes``, ``init_optimizer_and_scheduler``, ``optimize``; ``inference(dataloader)`` and
  ``inference_forward_windows(data)`` have the reference's signatures and read the window rule from ``self.args``.
* ``DeviceEvalMixin`` carries only the evaluation side (window stitching and confusion counts on the device, SURVEY.md
  section 8 f3): ``class MyTrainer(DeviceEvalMixin, reference_trainer.Trainer)`` keeps the reference's own fit loop, logging
  and checkpoints and replaces its two evaluation methods.

Logging (dllogger), pickled outputs, PerfTracker reports and best-model files stay with the reference's trainer: they are
its control plane, not this path.
"""
from collections import Counter
from types import SimpleNamespace

import numpy as np
import torch

from . import metrics
from .lfan import cross_entropy_loss

VIDEO, VGGISH, BERT, LOGMEL = "video", "vggish", "bert", "logmel"
EXPR = "EXPR_continuous_label"
TRAINSET, VALIDSET, TESTSET = "train", "valid", "test"   # constants.py: TRAINSET / VALIDSET / TESTSET


def windowing(x, window_length, hop_length):
    """trainer.py:894-912 (inclusive ``>=``): inference-time windows of a whole video."""
    n = len(x)
    if n >= window_length:
        steps = (n - window_length) // hop_length + 1
        out = [x[i * hop_length:i * hop_length + window_length] for i in range(steps)]
        if out[-1][-1] < n - 1:
            out.append(x[-window_length:])
        return out
    return [x]


def dataset_windowing(x, window_length, hop_length):
    """base/dataset.py:434-453 (strict ``>``): training-time windows of a trial."""
    n = len(x)
    if n > window_length:
        steps = (n - window_length) // hop_length + 1
        out = [x[i * hop_length:i * hop_length + window_length] for i in range(steps)]
        if out[-1][-1] < n - 1:
            out.append(x[-window_length:])
        return out
    return [x]


def _num_frames(modality, t):
    if modality == VIDEO:
        return t.shape[1]
    if modality in (VGGISH, BERT, LOGMEL):
        return t.shape[2]
    raise NotImplementedError(modality)


def _take(modality, t, wd):
    return t[:, wd, ...] if modality == VIDEO else t[:, :, wd, ...]


def _is_gpu(device):
    return torch.device(device).type == "cuda"


class DeviceEvalMixin:
    """``inference`` / ``inference_forward_windows`` / ``window_input`` with the reference's signatures
    (trainer.py:436,788,832) on top of whatever trainer provides ``self.model``, ``self.device``, ``self.number_classes``,
    ``self.train_batch_size`` and ``self.args`` (``window_length``, ``hop_length``, ``model_name``; optional ``amp``).

    ``self.eval_aggregate``: "device" (default on a GPU) keeps logits on the card, stitches overlapping windows with one
    kernel and folds every video into device-side confusion counts; "host" is the reference's own sequence (one forward
    per window, indexed adds, numpy scores) and the checker of the device path.  ``self.eval_frame_budget`` bounds the frames
    one forward may carry when windows are batched (default: ``train_batch_size x window_length``, the training
    footprint), so a long video never needs more activation memory than a training step."""

    eval_aggregate = None          # None -> "device" on a GPU, "host" otherwise
    eval_frame_budget = None
    eval_keep_logits = True        # the reference always returns (and pickles) the per-video logits (trainer.py:500-523)
    ignore_classes = (None,)

    # ---- small accessors so the mixin works on the reference trainer (args namespace) and on ours
    def _arg(self, name, default=None):
        args = getattr(self, "args", None)
        if args is not None and hasattr(args, name):
            return getattr(args, name)
        return getattr(self, name, default)

    def _aggregate(self, aggregate=None):
        aggregate = aggregate or self.eval_aggregate or ("device" if _is_gpu(self.device) else "host")
        if aggregate not in ("device", "host"):
            raise ValueError(aggregate)
        if aggregate == "device" and not _is_gpu(self.device):
            raise ValueError("aggregate='device' needs a GPU trainer (the stitch / confusion kernels take device pointers)")
        return aggregate

    def window_input(self, data):
        sizes = [[t.shape[0], _num_frames(m, t)] for m, t in data.items()]
        for s in sizes:
            assert s == sizes[0], f"{s} | {sizes[0]}"
        windows = windowing(np.arange(sizes[0][1]), self._arg("window_length"), self._arg("hop_length"))
        return [[{m: _take(m, t, wd) for m, t in data.items()}, wd] for wd in windows]

    def inference_forward_windows(self, data, aggregate=None):
        """Forward a video longer than the model's window (trainer.py:832-892).  Device path: the windows go through the
        model in groups of at most ``eval_frame_budget`` frames (eval mode: clips are independent), then ONE kernel
        scatter-adds them in window order and divides by the overlap counts (``eval_device.stitch_windows``).  Host path:
        the reference's own sequence.  A batch of several videos (bsz > 1; the reference asserts bsz == 1 in ``inference``
        but not here) always takes the host path, whose indexed adds carry the batch dimension."""
        total = _num_frames(*next(iter(data.items())))
        chunks = self.window_input(data)
        last = int(chunks[-1][1][-1])
        assert total == last + 1, f"{total} | {last + 1}"
        aggregate = self._aggregate(aggregate)
        bsz = next(iter(data.values())).shape[0]
        if aggregate == "device" and bsz == 1:
            from .eval_device import stitch_windows
            wlen = len(chunks[0][1])
            budget = self.eval_frame_budget or max(1, int(self._arg("train_batch_size", 1))) * int(self._arg("window_length"))
            group = max(1, budget // max(wlen, 1))
            outs = []
            for g0 in range(0, len(chunks), group):
                part = chunks[g0:g0 + group]
                batch = {m: torch.cat([c[m] for c, _ in part], dim=0).contiguous() for m in part[0][0]}
                out = self.model(batch)                               # [n_windows_in_group, window_length, n_cls]
                assert out.ndim == 3, out.ndim
                outs.append(out.float())
            out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
            return stitch_windows(out, [int(wd[0]) for _, wd in chunks], total).unsqueeze(0)
        results = []
        for chunk, wd in chunks:
            out = self.model({m: t.contiguous() for m, t in chunk.items()})
            assert out.ndim == 3, out.ndim
            results.append((out, wd))
        final = torch.zeros((results[-1][0].shape[0], total, results[-1][0].shape[2]), device=results[-1][0].device,
                            dtype=results[-1][0].dtype)
        idx = []
        for out, wd in results:
            final[:, wd, ...] = final[:, wd, ...] + out
            idx += wd.tolist()
        counts = sorted(Counter(idx).items())
        freqs = torch.tensor([c for _, c in counts], dtype=final.dtype, device=final.device).view(1, -1, 1)
        where = np.asarray([i for i, _ in counts], dtype=np.int64)
        final[:, where, ...] = final[:, where, ...] / freqs
        return final

    @torch.no_grad()
    def inference(self, dataloader, keep_logits=None, aggregate=None):
        """Trainer.inference (trainer.py:436-523): returns ``(current_perf, per_video_frame_logits)``.  Device path: each
        video is folded into device-side confusion counts (``DeviceEvalAccumulator``) and the scores come from one small
        copy at the end; the per-video ``{labels, logits}`` dictionary the reference returns is filled when
        ``keep_logits`` (default ``self.eval_keep_logits`` = True, as the reference; False skips the per-video copies)."""
        aggregate = self._aggregate(aggregate)
        keep_logits = self.eval_keep_logits if keep_logits is None else keep_logits
        self.model.eval()
        per_video = {}
        acc = None
        if aggregate == "device":
            from .eval_device import DeviceEvalAccumulator
            acc = DeviceEvalAccumulator(self.number_classes, self.ignore_classes, device=self.device)
        amp = bool(self._arg("amp", False)) and _is_gpu(self.device)
        for X, trials, lengths, indices in dataloader:
            inputs = {k: v.to(self.device) for k, v in X.items()}
            labels = inputs.pop("continuous_label", None)
            if labels is None:
                labels = inputs.pop(EXPR, None)
            nframes = 0
            for m, t in inputs.items():
                assert t.shape[0] == 1, f"{t.shape[0]} | {m}"
                nframes = _num_frames(m, t)
            if labels.numel() == self.train_batch_size:     # the reference's "todo : fix this." label hack (:468-472)
                labels = torch.zeros((self.train_batch_size, len(indices[0]), 1), dtype=torch.float32, device=self.device)
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                if nframes > self._arg("window_length") and self._arg("model_name") == "LFAN":
                    outputs = self.inference_forward_windows(inputs, aggregate)
                else:
                    outputs = self.model(inputs)
            outputs = outputs.detach().float()
            bsz, nfms, d = labels.shape
            assert d == 1 and tuple(outputs.shape) == (bsz, nfms, self.number_classes), tuple(outputs.shape)
            if acc is not None:
                acc.add(outputs.contiguous().view(bsz * nfms, -1), labels.contiguous().view(bsz * nfms))
            if keep_logits or acc is None:
                per_video[trials[0]] = {"labels": labels.contiguous().view(bsz * nfms).long().cpu().numpy().flatten(),
                                        "logits": outputs.contiguous().view(bsz * nfms, -1).cpu().numpy()}
        if acc is not None:
            return acc.compute(), per_video
        return metrics.compute_perf(per_video, self.ignore_classes), per_video


_REFERENCE_KWARGS = ("device", "emotion", "model_name", "models", "save_path", "fold", "min_epoch", "max_epoch",
                     "early_stopping", "learning_rate", "min_learning_rate", "patience", "train_batch_size",
                     "eval_batch_size", "criterion", "factor", "verbose", "milestone", "metrics",
                     "load_best_at_each_epoch", "save_plot")   # experiment.py:153-175


class Trainer(DeviceEvalMixin):
    """``Trainer(**trainer_kwards)`` as ``experiment.py:153-178`` builds it, or the short form
    ``Trainer(model, optimizer=..., window_length=..., ...)`` the benches and parity tests use."""

    def __init__(self, model=None, optimizer=None, criterion=None, device="cuda", window_length=300, hop_length=200,
                 model_name="LFAN", train_batch_size=2, number_classes=None, data_parallel=None, ignore_classes=(None,),
                 **kwargs):
        if model is None:
            if "models" not in kwargs:
                raise TypeError("Trainer needs the model: Trainer(model, ...) or Trainer(models=model, ...) as experiment.py does")
            model = kwargs.pop("models")
        for k in _REFERENCE_KWARGS:                      # base/trainer.py:21-45,96-115 keep them as attributes
            if k in kwargs:
                setattr(self, k, kwargs.pop(k))
        if kwargs:
            raise TypeError(f"unexpected Trainer arguments: {sorted(kwargs)}")
        self.device = device
        self.model = model.to(device) if hasattr(model, "to") else model     # base/trainer.py:25
        self.optimizer, self.scheduler = optimizer, None
        self.criterion = criterion if criterion is not None else cross_entropy_loss
        self.model_name, self.train_batch_size = model_name, train_batch_size
        self.number_classes = number_classes if number_classes is not None else 7
        self.ddp, self.ignore_classes = data_parallel, tuple(ignore_classes)
        # trainer.py:436-523,832 read the window rule and model name from the argparse namespace
        self.args = SimpleNamespace(window_length=window_length, hop_length=hop_length, model_name=model_name, amp=False)
        self.epoch, self.counter, self.seed = 0, 0, 0
        self.dataloaders = None
        self.fit_finished = False
        self.start_epoch = 0
        self.cl_to_int, self.int_to_cl = {}, {}
        self.max_epoch = getattr(self, "max_epoch", 0)

    # the short form's attribute names stay readable
    window_length = property(lambda self: self.args.window_length)
    hop_length = property(lambda self: self.args.hop_length)

    # ------------------------------------------------------------------ the calls experiment.py:178-182 makes
    def set_args(self, args):
        """trainer.py:92-93.  ``args`` is the argparse namespace (or any object / dict with its fields)."""
        if isinstance(args, dict):
            args = SimpleNamespace(**args)
        for k, v in vars(self.args).items():            # keep the short form's window rule unless args overrides it
            if not hasattr(args, k):
                setattr(args, k, v)
        self.args = args
        if getattr(args, "model_name", None):
            self.model_name = args.model_name

    def post_set_args(self, class_id=None):
        """trainer.py:95-105 loads ``<folds_dir>/split-<fold>/class_id.yaml``; the file belongs to the dataset folds, which
        are not on this path -- pass the mapping directly (``{class name: int}``) or leave it empty."""
        assert self.args is not None
        self.cl_to_int = dict(class_id or {})
        self.int_to_cl = {v: k for k, v in self.cl_to_int.items()}
        assert len(self.int_to_cl) == len(self.cl_to_int), "more than 1 key with same value. wrong."

    def set_number_classes(self, ncls):
        assert isinstance(ncls, int), type(ncls)
        assert ncls > 0, ncls
        self.number_classes = ncls

    def get_parameters(self):
        """base/trainer.py:83-93"""
        return [p for _, p in self.model.named_parameters() if p.requires_grad]

    def init_optimizer_and_scheduler(self, epoch=0):
        """trainer.py:127-134 -> instantiators.py:62-140: ``opt__``-prefixed hyper-parameters; SGD is built WITHOUT ``lr``
        (instantiators.py:74-79: torch's default 1e-3 applies until a scheduler changes it).  With ``data_parallel`` the
        same update runs as ONE fused launch over the flat bucket (``FlatNesterovSGD``, bit-identical to torch.optim.SGD)."""
        a = {k.split("__", 1)[1] if k.startswith("opt") and "__" in k else k: v for k, v in vars(self.args).items()}
        name = a.get("name_optimizer", "sgd")
        params = self.get_parameters()
        if name == "sgd":
            hp = dict(momentum=a.get("momentum", 0.9), dampening=a.get("dampening", 0.0),
                      weight_decay=a.get("weight_decay", 1e-4), nesterov=a.get("nesterov", True))
            if self.ddp is not None and hp["nesterov"] and hp["dampening"] == 0.0:
                from .data_parallel import FlatNesterovSGD
                self.optimizer = FlatNesterovSGD(self.ddp, lr=1e-3, momentum=hp["momentum"], weight_decay=hp["weight_decay"])
            else:
                self.optimizer = torch.optim.SGD(params=params, **hp)
        elif name == "adam":
            self.optimizer = torch.optim.Adam(params=params, betas=(a.get("beta1", 0.9), a.get("beta2", 0.999)),
                                              eps=a.get("eps_adam", 1e-8), weight_decay=a.get("weight_decay", 0.0),
                                              amsgrad=a.get("amsgrad", False))
        else:
            raise ValueError(f"Unsupported optimizer `{name}`")
        self.scheduler = None
        if a.get("lr_scheduler") and isinstance(self.optimizer, torch.optim.Optimizer):
            sched = a.get("name_lr_scheduler")
            if sched in ("step", "mystep"):
                self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=a.get("step_size", 1),
                                                                 gamma=a.get("gamma", 0.1))
            elif sched == "cosine":
                self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=a.get("t_max", 1),
                                                                            eta_min=a.get("min_lr", 0.0))
            else:
                raise NotImplementedError(f"lr scheduler {sched!r} (base/scheduler.py) stays with the reference's trainer")

    # ------------------------------------------------------------------ training
    def _split(self, X):
        inputs = {k: v.to(self.device) for k, v in X.items()}
        labels = inputs.pop("continuous_label", None)
        if labels is None:
            labels = inputs.pop(EXPR, None)
        return inputs, labels

    def train_step(self, X, indices=None):
        """One iteration of trainer.py:345-391.  Returns the (detached) loss tensor."""
        inputs, labels = self._split(X)
        if labels.numel() == self.train_batch_size:  # the reference's "todo : fix this." label hack (:360-363)
            n = len(indices[0]) if indices is not None else labels.shape[1]
            labels = torch.zeros((self.train_batch_size, n, 1), dtype=torch.float32, device=self.device)
        if self.ddp is not None:
            self.ddp.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        outputs = self.model(inputs)
        bsz, nfms, d = labels.shape
        assert d == 1, d
        assert outputs.ndim == 3 and tuple(outputs.shape) == (bsz, nfms, self.number_classes), tuple(outputs.shape)
        loss = self.criterion(outputs.contiguous().view(bsz * nfms, -1), labels.contiguous().view(bsz * nfms).long())  # trainer.py:380-383
        loss.backward()
        if self.ddp is not None:
            self.ddp.all_reduce_gradients()
        self.optimizer.step()
        return loss.detach()

    def train_one_epoch(self, dataloader=None):
        """trainer.py:315-434 (the reference reads ``self.dataloaders[TRAINSET]``; a loader may also be passed)."""
        if dataloader is None:
            dataloader = self.dataloaders[TRAINSET]
        self.model.train()
        running, count = 0.0, 0
        for X, trials, lengths, indices in dataloader:
            running = running + self.train_step(X, indices)
            count += 1
        self.counter += 1
        return float(running / max(count, 1))

    def optimize(self, dataloader_dict, checkpoint_controller=None, parameter_controller=None):
        """trainer.py:611-770 without its control plane: validate, ``max_epoch`` x (train epoch, scheduler step,
        validate, remember the best weights by frame-level weighted F1), then test the best weights.  Returns
        ``{"valid": [perf per evaluation], "loss": [epoch losses], "test": perf, "best_epoch": i}``."""
        self.dataloaders = dataloader_dict
        if self.optimizer is None:
            self.init_optimizer_and_scheduler(epoch=0)
        history = {"valid": [], "loss": []}

        def master(perf):
            return perf[self.ignore_classes[0]][metrics.W_F1][metrics.FRAME_LEVEL]["master"]
        perf, _ = self.inference(dataloader_dict[VALIDSET], keep_logits=False)
        history["valid"].append(perf)
        best, best_state, best_epoch = master(perf), {k: v.detach().clone() for k, v in self.model.state_dict().items()}, -1
        for epoch in range(int(self.max_epoch)):
            history["loss"].append(self.train_one_epoch())
            if self.scheduler is not None:
                self.scheduler.step()
            if parameter_controller is not None and hasattr(parameter_controller, "step"):
                parameter_controller.step(epoch)
            perf, _ = self.inference(dataloader_dict[VALIDSET], keep_logits=False)
            history["valid"].append(perf)
            if master(perf) > best:
                best, best_epoch = master(perf), epoch
                best_state = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        self.fit_finished = True
        if TESTSET in dataloader_dict:
            self.model.load_state_dict(best_state, strict=True)
            history["test"], history["test_logits"] = self.inference(dataloader_dict[TESTSET])
        history["best_epoch"] = best_epoch
        return history
