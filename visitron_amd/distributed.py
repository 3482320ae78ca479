"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference's only parallelism is DDP with bucketed
gradient all-reduce (tasks/viewpoint_select/pretrain.py:96-102,191) plus seven scalar all-reduces
for logged metrics (:169-189); here the gradients already live in ONE flat slab, so a bucket is a
slice of it -- no flatten/unflatten copies -- and the seven scalars travel as one 7-float message.
"""
import torch
import torch.distributed as dist


def bucket_ranges(n, bucket_elems):
    """[start, end) slices covering n elements, each <= bucket_elems (last one may be short)."""
    if bucket_elems <= 0:
        raise ValueError("bucket_elems must be positive")
    return [(s, min(n, s + bucket_elems)) for s in range(0, n, bucket_elems)]


def all_reduce_flat(flat, bucket_elems, group=None, async_handles=None):
    """In-place SUM all-reduce of a flat tensor in fixed-size buckets (several collectives in flight:
    on xGMI a ring is bound per link, so a few large messages keep every link busy)."""
    works = []
    for s, e in bucket_ranges(flat.numel(), bucket_elems):
        works.append(dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True))
    if async_handles is not None:
        async_handles.extend(works)
        return
    for w in works:
        w.wait()


def all_reduce_ranges(flat, ranges, bucket_elems, group=None, async_handles=None):
    """all_reduce_flat over a list of [start, end) element ranges of `flat` (e.g. the gradient slab
    regions of the encoder layers whose backward has just been enqueued)."""
    for s, e in ranges:
        if e > s:
            all_reduce_flat(flat[s:e], bucket_elems, group=group, async_handles=async_handles)


def complement_ranges(n, ranges):
    """[0, n) minus the given disjoint ranges, as a sorted list of ranges."""
    out, cur = [], 0
    for s, e in sorted(ranges):
        if s > cur:
            out.append((cur, s))
        cur = max(cur, e)
    if cur < n:
        out.append((cur, n))
    return out


def all_reduce_metrics(values, group=None):
    """The reference's 7x (x /= world; all_reduce(SUM)) of pretrain.py:169-189 as one message."""
    world = dist.get_world_size(group)
    t = torch.stack([v if torch.is_tensor(v) else torch.tensor(float(v), device=values[0].device) for v in values])
    t = t.to(torch.float32) / world
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tuple(t.unbind(0))


def shard_batch(batch, rank, world):
    """Rank's contiguous slice of a global batch (what DistributedSampler amounts to for one step)."""
    out = {}
    for k, v in batch.items():
        n = v.shape[0]
        assert n % world == 0, "global batch must divide by world size"
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out
