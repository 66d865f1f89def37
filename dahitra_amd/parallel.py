"""Data parallelism for the change-detection train step: one process per GPU, image pairs sharded by
rank, BatchNorm statistics per replica (what the reference's nn.DataParallel does,
models/networks.py:121-125), and ONE all-reduce per step over the net's flat fp32 gradient arena
(only grad-carrying parameters: 12.0 MB for base_transformer_pos_s4, 16.8 MB for newUNetTrans).

backend "nccl" is RCCL on ROCm; on an 8xMI355X node the ring runs over xGMI.  The 1/world factor of
the mean is folded into the AdamW kernel's grad_scale, so the reduced buffer is consumed as is."""
import os

import torch
import torch.distributed as dist
import torch.utils.data


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DAHITRA_FORCE_DIST=1: build the process group even for one rank, so that the multi-GPU code path (RCCL init,
    # barrier, gradient all-reduce, AdamW with the 1/world factor after the graph replay) can be exercised on one GPU
    force = os.environ.get("DAHITRA_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_batch(global_batch, rank, world):
    """contiguous, equal shards (equal sizes make mean-of-shard-means == global mean of the focal loss)"""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def exchange_enabled(group=None):
    """True when a step must run the gradient exchange: more than one rank, or DAHITRA_FORCE_DIST=1 (one-rank process
    group: the collective is the identity but the same calls are made)"""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DAHITRA_FORCE_DIST", "0") == "1"


def allreduce_sum_(flat, group=None):
    """in-place SUM all-reduce of a flat gradient buffer; returns the factor (1/world) that turns it
    into the mean (pass it to AdamW.step(grad_scale=...))."""
    if not exchange_enabled(group):
        return 1.0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


def allreduce_net_grads_(net, group=None):
    """all-reduce the whole gradient arena of a dahitra_amd CDNet in one collective"""
    _, grad = net.flat_params()
    return allreduce_sum_(grad, group)


def broadcast_params_(net, src=0, group=None):
    """replicas start from rank `src`'s parameters and BN buffers"""
    if not exchange_enabled(group):
        return
    flat = net._arena.flat
    dist.broadcast(flat, src=src, group=group)
    for b in net.buffers():
        dist.broadcast(b, src=src, group=group)


def broadcast_buffers_(net, src=0, group=None):
    """BatchNorm running statistics (and num_batches_tracked) of rank `src` on every rank.  They stay per replica during
    training (the reference's nn.DataParallel keeps device 0's, models/networks.py:121-125); before an evaluation pass that
    is sharded over the ranks, and before a checkpoint, every rank must score / store the SAME model -- rank 0's, the one
    best_ckpt.pt holds."""
    if not exchange_enabled(group):
        return
    for b in net.buffers():
        dist.broadcast(b, src=src, group=group)


class ShardSampler(torch.utils.data.Sampler):
    """indices rank, rank + world, ... of a dataset, in order, WITHOUT the padding of DistributedSampler (which repeats
    samples until every rank has the same count: a summed confusion matrix would count them twice).  Ragged per-rank
    lengths are fine where no collective runs inside the loop (the trainers' evaluation pass)."""

    def __init__(self, n, rank, world):
        self.idx = list(range(int(rank), int(n), int(world)))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


# ---- which form of the data-parallel step: ONE graph + one all-reduce + AdamW ("serial"), or TWO graphs around an all-reduce of
# the arena tail that overlaps the rest of the backward ("overlapped", dahitra_amd/graph.py).  The overlapped form is not free: two
# graph launches, an event hand-over to RCCL's stream and back, and a second graph that may not hold the persistent one-launch
# BatchNorm backward (its device-wide barrier needs every CU; RCCL's kernels hold some) -- measured on ONE rank, where the
# collectives move nothing: 3.52 against 3.34 ms per step of base_transformer_pos_s4, i.e. ~0.18 ms (bench.py `ddp_rehearsal`).
# It pays when the all-reduce it hides takes longer than that.  No multi-GPU node was available to any round of this build, so
# the crossover is a MODEL, stated here and overridable: a ring all-reduce over xGMI (point-to-point links, per-link bound)
#     t(bytes, world) = 10 us + 2 (world - 1) x 5 us  +  2 (world - 1) / world x bytes / (0.6 x 153 GB/s)
# DAHITRA_OVERLAP = auto (default: overlapped iff t(tail bytes, world) > 1.5 x DAHITRA_OVERLAP_OVERHEAD_US [180]) | 1 (always, when
# the net has a split point) | 0 (never; DAHITRA_NO_OVERLAP=1 is the older spelling).  DAHITRA_XGMI_GBS overrides the 92 GB/s.
# The factor 1.5 is deliberate: the model is the PESSIMISTIC end for the collective (one ring, one link per hop; RCCL runs several
# rings over the fully connected xGMI mesh), the overlapped form still leaves the head's all-reduce exposed, and the serial form
# is the one with fewer moving parts -- so the two-graph form is taken only where it is modelled to win clearly: arenas of tens
# of MB (the ResNet-50 variant at N >= 4).  For the 12 MB / 16.8 MB arenas of base_transformer_pos_s4 / newUNetTrans every world
# size up to 8 takes one graph + one all-reduce: 3.36 ms against 3.33 ms single-process on one rank, + the collective itself.
def overlap_mode():
    if os.environ.get("DAHITRA_NO_OVERLAP", "0") == "1":
        return "0"
    m = os.environ.get("DAHITRA_OVERLAP", "auto")
    if m not in ("0", "1", "auto"):
        raise ValueError("DAHITRA_OVERLAP=%s (0, 1 or auto)" % m)
    return m


def allreduce_model_us(nbytes, world):
    """modelled duration of a ring all-reduce of `nbytes` over `world` ranks on xGMI (see above); 0 for one rank"""
    if world <= 1:
        return 0.0
    gbs = float(os.environ.get("DAHITRA_XGMI_GBS", "92"))
    return 10.0 + 2 * (world - 1) * 5.0 + 2.0 * (world - 1) / world * nbytes / (gbs * 1e3)


def split_offset(net, world=None):
    """Arena offset (in floats) at which the overlapped data-parallel step cuts the gradient exchange, or None.
    The backward runs in two parts (Engine.backward_first / backward_second); every gradient the SECOND part writes (BiT nets:
    stem, layer1, layer2; newUNetTrans / xBD: the whole ResNet trunk -- Engine.split_prefixes) must lie below the offset, so
    that the tail [offset, end) is final after the first part and can be all-reduced while the second part computes.  None when
    the form is switched off (overlap_mode), when the net has no such parameters, when less than 1 MB would be overlapped, or --
    DAHITRA_OVERLAP=auto with `world` given -- when the modelled all-reduce of the tail is shorter than what the two-graph form costs."""
    mode = overlap_mode()
    if mode == "0":
        return None
    off = net._arena.offsets
    early = net._engine.split_prefixes()
    second = [k for k in net._active_keys if k.startswith(early)]
    if not second or len(second) == len(net._active_keys):
        return None
    split = max(off[k][0] + off[k][1] for k in second)           # end of the last gradient the second part writes
    if (net._arena.n_active - split) * 4 < (1 << 20):
        return None
    if mode == "auto" and world is not None:
        overhead = float(os.environ.get("DAHITRA_OVERLAP_OVERHEAD_US", "180"))
        if allreduce_model_us((net._arena.n_active - split) * 4, world) <= 1.5 * overhead:
            return None
    return split


def allreduce_counts_(counts, group=None):
    """in-place SUM of an integer count tensor (the trainers' device-side confusion matrix) over the ranks"""
    if exchange_enabled(group):
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def barrier(group=None):
    """no-op without a process group"""
    if dist.is_initialized():
        dist.barrier(group=group)
