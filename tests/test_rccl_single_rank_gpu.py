"""De-risking the first multi-GPU run on ONE GPU (round-2 verdict, task 6): every collective of the data-parallel path has
only ever run on gloo, whose stream semantics differ from RCCL's (gloo works on host copies and blocks; RCCL enqueues on
its own stream and ``wait()`` is a stream dependency).  A single-rank RCCL communicator is legal, so a freshly spawned child
initialises ``backend="nccl"`` (= RCCL) with world_size 1 and drives ``ClipDataParallel`` with the overlapped slices FORCED
on: async all-reduces launched from autograd hooks while the backward is still producing the slices below, ``wait()``,
the 1/world scaling, the fused optimiser.  With one rank the reduction is the identity, so the result has to equal the
non-overlapped step BIT FOR BIT -- any missing stream dependency (a slice reduced before its last gradient landed, the
optimiser reading a slice the collective still owns) shows as a difference.  ``broadcast_state`` and ``sync_buffers`` run
through the same communicator.  N > 1 stays unmeasured."""
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


def _child(rank, port, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.modules.setdefault("triton", None)
    import torch.distributed as dist
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD, init_process_group_from_env
    from feature_vs_text_compound_emotion_amd.lfan import LFAN, cross_entropy_loss
    init_process_group_from_env(backend="nccl", single_rank_group=True)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    def model():
        sd = synth.lfan_state_dict(MODS, n_cls=7, head_hw=HW // 8, seed=0)
        m = LFAN(backbone_settings={}, output_dim=7, task="CLASSIFICATION", modality=MODS, example_length=L, kernel_size=5,
                 tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cuda", head_hw=HW // 8)
        m.init(load_backbone=False)
        m.load_state_dict(sd, strict=True)
        return m.cuda().eval()

    x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=55)
    xd, ld = {k: v.cuda() for k, v in x.items()}, labels.cuda()

    def run(overlap):
        m = model()
        ddp = ClipDataParallel(m, overlap=overlap, bucket_mb=1.0)
        opt = FlatNesterovSGD(ddp, lr=1e-3)
        grads = []
        for _ in range(3):
            ddp.zero_grad()
            loss = cross_entropy_loss(m(dict(xd)), ld)
            loss.backward()
            ddp.all_reduce_gradients()
            grads.append(ddp.flat.clone())
            opt.step()
        return ddp, torch.stack(grads).cpu(), ddp.flat_param.clone().cpu()

    ddp_o, g_o, w_o = run("force")
    assert ddp_o.overlap and len(ddp_o.buckets) >= 10, len(ddp_o.buckets)
    ddp_p, g_p, w_p = run(False)
    assert not ddp_p.overlap
    out["grad_equal"] = bool(torch.equal(g_o, g_p))
    out["weight_equal"] = bool(torch.equal(w_o, w_p))
    out["grad_maxdiff"] = float((g_o - g_p).abs().max())
    out["buckets"] = len(ddp_o.buckets)
    # broadcast_state / sync_buffers through RCCL: identity with one rank, and the version counters move (packed-weight caches)
    before = {k: v.clone() for k, v in ddp_o.model.state_dict().items()}
    versions = [p._version for p in ddp_o.model.parameters()]
    ddp_o.broadcast_state()
    ddp_o.sync_buffers("mean", force=True)
    ddp_o.sync_buffers("broadcast", force=True)
    torch.cuda.synchronize()
    after = ddp_o.model.state_dict()
    out["state_equal"] = all(torch.equal(before[k], after[k]) for k in before)
    out["versions_moved"] = all(p._version > v for p, v in zip(ddp_o.model.parameters(), versions))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_all_reduce_on_a_single_rank_rccl_communicator():
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_child, args=(_free_port(), out), nprocs=1, join=True)   # a fresh child: nothing here re-execs a GPU process
        res = dict(out)
    print(f"\n[rccl x1] {res}")
    assert res["buckets"] >= 10
    assert res["grad_equal"] and res["weight_equal"], res
    assert res["state_equal"] and res["versions_moved"], res
