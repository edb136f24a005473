"""Data-parallel host logic: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The reference only has single-process `nn.DataParallel` (train.py:249-252): per step it
broadcasts all parameters from GPU 0, scatters the batch, and reduce-adds gradients onto GPU 0
(SURVEY.md §2.3 C1-C4).  Here every rank keeps its own replica and its own shard of the batch
(images are independent units: per-image loss normalisation, losses.py:118/161, batch mean :168),
so the only exchange is ONE averaged all-reduce of the flat gradient arena (+ 2 floats of loss
for the `loss == 0` skip decision, train.py:316-317, which must be identical on every rank).
"""
import torch
import torch.distributed as dist


def world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_range(n, rank, size):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(flat, group=None, bucket_bytes=0):
    """Average `flat` (the flat gradient arena) over the group, in place.

    xGMI is point-to-point (7 links/GPU): a ring moves 2*(N-1)/N*S through one link, so a few
    LARGE messages beat many small ones; by default the whole arena (152 MB fp32 for R50) is one
    collective.  `bucket_bytes` > 0 splits it (to overlap with the tail of backward)."""
    _, size = world(group)
    if size == 1:
        return flat
    if bucket_bytes and flat.numel() * flat.element_size() > bucket_bytes:
        step = max(1, bucket_bytes // flat.element_size())
        works = [dist.all_reduce(flat[i:i + step], op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for i in range(0, flat.numel(), step)]
        for w in works:
            w.wait()
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / size)
    return flat


def reduce_losses_(losses, group=None):
    """Mean of the per-rank [cls_loss, reg_loss] (equal shards: mean of shard means == global mean,
    train.py:312-313); every rank then takes the same `loss == 0` skip decision."""
    _, size = world(group)
    if size > 1:
        dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=group)
        losses.mul_(1.0 / size)
    return losses


def syncbn_stats(sum_, sumsq, count, eps, group=None):
    """Cross-replica BatchNorm statistics with the reference's SyncBN arithmetic
    (utils/sync_batchnorm/batchnorm.py:128-145): mean = sum/n, sumvar = sumsq - sum*mean,
    biased var = sumvar/n, unbiased = sumvar/(n-1), inv_std = clamp(biased var, eps)^-1/2
    (a clamp, not `+ eps`).  One all-reduce of the fused [sum | sumsq | count] buffer per layer.
    Returns (mean, inv_std, unbiased_var, n)."""
    _, size = world(group)
    buf = torch.cat([sum_.reshape(-1), sumsq.reshape(-1),
                     torch.tensor([float(count)], dtype=sum_.dtype, device=sum_.device)])
    if size > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    c = sum_.numel()
    s, ss, n = buf[:c], buf[c:2 * c], buf[2 * c]
    mean = s / n
    sumvar = ss - s * mean
    inv_std = (sumvar / n).clamp(min=eps).pow(-0.5)
    return mean, inv_std, sumvar / (n - 1), n
