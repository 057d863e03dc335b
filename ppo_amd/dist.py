"""Data parallelism over envs/samples: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code at all (single process, README.md:23); what must hold is
that W shards reproduce the single-buffer update of PPO.py:200-251:
  * PPO.py:212 normalises over the WHOLE buffer -> the per-shard (count, mean, M2) triples are
    all-gathered and merged in rank order (identical result on every rank);
  * PPO.py:244 differentiates the mean over ALL samples -> every shard scales its gradient by
    1/n_global and the P+2 floats (gradient + the two loss terms) are summed with one all-reduce
    per epoch.  The message is ~37-49 KB, i.e. latency-bound: one collective per epoch, no buckets.
Parameters and Adam state are replicated and stay bit-identical because every rank applies the
same all-reduced gradient.
"""
import os

import torch
import torch.distributed as td


def is_distributed():
    """More than one rank - or PPO_DIST_FORCE=1 on a one-rank group, which sends a single process
    down the data-parallel code path (per-epoch reduce, all-reduce, Adam) for rehearsal and timing."""
    return td.is_available() and td.is_initialized() and \
        (td.get_world_size() > 1 or os.environ.get("PPO_DIST_FORCE") == "1")


def rank():
    return td.get_rank() if is_distributed() else 0


def merge_moments_list(triples):
    """Chan et al. pairwise merge of (count, mean, M2) rows [W, 3] in row order (fp64)."""
    n, mean, m2 = triples[0, 0], triples[0, 1], triples[0, 2]
    for i in range(1, triples.shape[0]):
        nb, mb, m2b = triples[i, 0], triples[i, 1], triples[i, 2]
        tot = n + nb
        delta = mb - mean
        mean = mean + delta * (nb / tot)
        m2 = m2 + m2b + delta * delta * (n * nb / tot)
        n = tot
    return torch.stack([n, mean, m2])


def _stage_through_host(t):
    """gloo (the CPU backend the tests and single-GPU rehearsals use) gets host tensors.  Its
    collectives on device tensors stage through the host anyway, on a side stream ordered behind the
    caller's stream by an event; with two ranks time-slicing one GPU that wait was seen not to end
    when the work in front of it was the fused fwd/bwd kernel (it always ends after a host
    synchronize, and behind ordinary kernels: profiles/r03_gloo_probe.txt).  The blocking copy here
    is that host synchronize.  RCCL ("nccl", one process per GPU) works on the device tensors."""
    return t.is_cuda and td.get_backend() == "gloo"


def merge_moments(moments):
    """moments: fp64 [3] on this rank -> merged over all ranks (same tensor if not distributed)."""
    if not is_distributed():
        return moments
    world = td.get_world_size()
    src = moments.cpu() if _stage_through_host(moments) else moments.contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    td.all_gather(parts, src)
    return merge_moments_list(torch.stack(parts)).contiguous().to(moments.device)


def all_reduce_sum(t):
    if is_distributed():
        if _stage_through_host(t):
            h = t.cpu()
            td.all_reduce(h, op=td.ReduceOp.SUM)
            t.copy_(h)
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM)
    return t
