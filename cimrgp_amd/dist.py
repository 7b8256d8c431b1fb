"""Sharding of the independent (resolution, region) blocks over the GPUs of a node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm,
"gloo" in the CPU tests).  Blocks of one layer are independent; layers form a
short chain through the residual f_bar (Stats.py:126-157).  Communication:
  * fit: ownership is NESTED (plan_layers): from the first layer with at least as many regions as ranks on, a
    rank fits the blocks inside its own row ranges and needs nobody else's predictions; only the coarser layers
    exchange (one all-reduce each of the layer's training-point prediction, N x q, disjoint slices summed), and ONE
    all-reduce after the sweep assembles the local layers' predictions (and the blocks' failure flags) everywhere;
  * predict: ONE all-reduce of the fused [mean | var] buffer (N* x (q+1)) --
    the sum over resolutions of MRGP.py:802-803.
Payloads are a few MiB: latency-bound on xGMI, so a single fused buffer per
exchange and no bucketing.
"""
import numpy as np
import torch
import torch.distributed as td


def world(group=None):
    """(rank, world_size) of ``group``; (0, 1) when torch.distributed is not initialised."""
    if td.is_available() and td.is_initialized():
        return td.get_rank(group), td.get_world_size(group)
    return 0, 1


def assign_blocks(sizes, world_size):
    """Longest-processing-time assignment of one layer's blocks to ranks.

    ``sizes``: block lengths n_l; cost model n^3 (the Cholesky).  Deterministic:
    ties are broken by region id, then by rank id, so every rank computes the
    same map without communication.  Returns an int array owner[l]."""
    sizes = np.asarray(sizes, dtype=np.float64)
    order = sorted(range(len(sizes)), key=lambda l: (-sizes[l], l))
    load = np.zeros(world_size)
    owner = np.zeros(len(sizes), dtype=np.int64)
    for l in order:
        r = int(np.argmin(load))       # first minimum -> lowest rank id on ties
        owner[l] = r
        load[r] += sizes[l] ** 3
    return owner


def plan_layers(bounds, world_size):
    """Ownership of every (resolution, region) block, and the first layer from which the residual chain is LOCAL.

    ``bounds``: per layer an array of (start, stop) rows.  The regions of the reference's uniform index set are
    nested contiguous ranges (IndexSetGenerator.py:51-65).  Let the ANCHOR be the first layer with at least
    ``world_size`` regions: its regions are dealt to the ranks as contiguous runs of equal cost (n^3), and every
    region of a finer layer goes to the rank that owns the anchor region containing it.  A rank then fits, in every
    layer from the anchor on, exactly the blocks inside its own row ranges, whose targets need the coarser layers'
    predictions on those ranges only -- its own: no exchange per layer.  The layers above the anchor
    have fewer blocks than ranks: longest-processing-time assignment, one all-reduce each.

    Returns (owners, first_local): owners[j][l] = rank; layers j >= first_local exchange nothing during the fit
    (first_local = number of layers when the regions do not nest, e.g. a sample count the divider does not divide:
    then every layer is assigned by LPT and exchanged, as before)."""
    n_layers = len(bounds)
    lpt = [assign_blocks([int(b) - int(a) for a, b in layer], world_size) for layer in bounds]
    if world_size <= 1:
        return lpt, n_layers
    anchor = next((j for j in range(n_layers) if len(bounds[j]) >= world_size), None)
    if anchor is None:
        return lpt, n_layers
    ab = [(int(a), int(b)) for a, b in bounds[anchor]]
    # contiguous runs of (nearly) equal cost: region l goes to the rank whose share of the total cost its midpoint falls in
    cost = np.asarray([float(b - a) ** 3 for a, b in ab])
    mid = np.cumsum(cost) - 0.5 * cost
    a_owner = np.minimum((mid / cost.sum() * world_size).astype(np.int64), world_size - 1)
    if len(set(a_owner.tolist())) < world_size:            # degenerate costs: fall back to equal counts
        a_owner = (np.arange(len(ab)) * world_size) // len(ab)
    owners = list(lpt[:anchor]) + [a_owner]
    starts = np.asarray([a for a, _ in ab])
    for j in range(anchor + 1, n_layers):
        own = np.zeros(len(bounds[j]), dtype=np.int64)
        for l, (a, b) in enumerate(bounds[j]):
            k = int(np.searchsorted(starts, int(a), side="right")) - 1
            if k < 0 or int(b) > ab[k][1]:                 # not inside one anchor region: the layers do not nest
                return lpt, n_layers
            own[l] = a_owner[k]
        owners.append(own)
    return owners, anchor


def allreduce_sum_(tensor, group=None, force=False):
    """In-place sum over ranks; a no-op for a single process unless ``force`` (then a group of one is
    reduced through its backend all the same: the only way to exercise RCCL on a one-GPU box)."""
    _, ws = world(group)
    if ws > 1 or (force and td.is_available() and td.is_initialized()):
        td.all_reduce(tensor, op=td.ReduceOp.SUM, group=group)
    return tensor


def allreduce_sum_begin(tensor, group=None, force=False):
    """Start the in-place sum over ranks and return at once: the handle's ``wait()`` makes the CURRENT STREAM
    wait for the collective (it does not block the host).  None when there is nothing to reduce.  A caller that
    next touches ``tensor`` a whole step later (bench.py) waits there and hides the collective behind the step."""
    _, ws = world(group)
    if ws > 1 or (force and td.is_available() and td.is_initialized()):
        return td.all_reduce(tensor, op=td.ReduceOp.SUM, group=group, async_op=True)
    return None


def share_one_gpu():
    """Several ranks SHARE one GPU (rehearsals and the 2-rank tests on a one-GPU box; production runs one
    process per GPU).  Each process must then stay within the runtime's four hardware queues: one queue for
    the carried rows (cimrgp_set_rows_queues(1)), no stream pool for the independent blocks of a layer, one
    look-ahead context.  Measured: two such processes with five or more streams each stall for hundreds of
    milliseconds between steps."""
    from . import _lib, Posteriors
    _lib.set_rows_queues(1)
    Posteriors.MAX_BLOCK_STREAMS = 1


class RemoteRankError(RuntimeError):
    """Raised on the ranks that did NOT fail when another rank of the group did."""


def raise_together(error, group=None, device=None):
    """Make a rank-local failure collective.

    ``error``: the exception this rank caught in its local phase, or None.  Every
    rank of the group calls this at the same point; if any rank holds an error,
    EVERY rank raises (the failing ones their own exception, the others
    ``RemoteRankError``) instead of the healthy ranks blocking for ever in the
    next collective.  Costs one 1-element all-reduce; a no-op for one process."""
    _, ws = world(group)
    if ws > 1:
        flag = torch.tensor([0.0 if error is None else 1.0], dtype=torch.float32,
                            device=device if device is not None else "cpu")
        td.all_reduce(flag, op=td.ReduceOp.MAX, group=group)
        if error is None and float(flag.item()) > 0:
            raise RemoteRankError("another rank of the process group failed in this phase")
    if error is not None:
        raise error
