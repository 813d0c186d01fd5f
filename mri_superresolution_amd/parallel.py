"""Data parallelism for the U-Net training step: one process per GPU, RCCL over xGMI.

The reference has no distributed code; this adds the capability BASELINE.json asks
for.  Semantics = single-process large batch: GroupNorm is per-sample, the loss is a batch mean, so
equal shards + gradient MEAN reproduce the global batch up to summation order.

The gradient lives in ONE flat fp32 buffer laid out in forward (registration) order; backward
completes it from the end towards the start, so buckets are contiguous suffix slices that are
all-reduced (sum) on RCCL's stream as soon as the kernels writing them have been queued -
overlapping the remaining backward convolutions.  The 1/world_size factor is folded into the fused
Adam kernel (``FusedAdam.dp_grad_scale``).
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
    """Rank-local sample indices, DistributedSampler-style: one permutation shared by all ranks
    (seed + epoch), padded by wrap-around to a multiple of world, strided by rank."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g).tolist()
    else:
        order = list(range(n_items))
    total = (n_items + world - 1) // world * world
    order += order[: total - n_items]
    return order[rank:total:world]


class GradBucketer:
    """Splits a flat gradient buffer into suffix buckets and all-reduces each when complete."""

    def __init__(self, flat_grads: torch.Tensor, offsets, bucket_bytes: int = 8 << 20, group=None):
        self.flat = flat_grads
        self.group = group
        self.offsets = dict(offsets)              # param name -> (offset, numel)
        self.total = flat_grads.numel()
        # bucket boundaries (element offsets), built from the end; boundaries fall on parameter starts
        starts = sorted({off for off, _ in self.offsets.values()}, reverse=True)
        cap = max(1, bucket_bytes // 4)
        self.bounds: List[int] = [self.total]
        for s in starts:
            if self.bounds[-1] - s >= cap:
                self.bounds.append(s)
        if self.bounds[-1] != 0:
            self.bounds.append(0)
        self.reset()

    def reset(self):
        self.next_bucket = 0                      # index into bounds: bucket i = [bounds[i+1], bounds[i])
        self.handles = []

    def layer_start(self, layer_name: str) -> int:
        """Lowest flat offset among the parameters whose gradients are complete once `layer_name`
        (reverse execution order) has finished its weight-gradient kernel."""
        offs = [off for k, (off, _) in self.offsets.items() if k.startswith(layer_name + ".") or k == layer_name]
        if not offs:
            raise KeyError(layer_name)
        return min(offs)

    def on_layer_done(self, layer_name: str):
        self.flush_down_to(self.layer_start(layer_name))

    def flush_down_to(self, offset: int):
        while self.next_bucket + 1 < len(self.bounds) and self.bounds[self.next_bucket + 1] >= offset:
            lo, hi = self.bounds[self.next_bucket + 1], self.bounds[self.next_bucket]
            self._launch(lo, hi)
            self.next_bucket += 1

    def _launch(self, lo: int, hi: int):
        if hi > lo:
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launches whatever is left (alpha and the stem sit at the front) and waits for all."""
        self.flush_down_to(0)
        for h in self.handles:
            h.wait()
        self.reset()


class DataParallel:
    """Wraps a UNetSuperRes replica: broadcast of the initial weights, bucketed overlapped gradient
    all-reduce, and scalar metric averaging.  ``optimizer.dp_grad_scale`` must be 1/world_size."""

    def __init__(self, model, group=None, bucket_bytes: int = 8 << 20):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        dist.broadcast(model.flat_params, src=0, group=group)
        model.mark_weights_changed()
        self.bucketer = GradBucketer(model.flat_grads, model._offsets, bucket_bytes, group)
        self._sync = True
        self._reduced = False          # flat_grads currently holds an all-reduced sum
        model.grad_ready_hook = self._on_layer_done
        model.backward_start_hook = self._on_backward_start

    # ---- hooks called by UNetSuperRes._run_backward
    def _on_backward_start(self, fresh: bool):
        """``fresh``: this backward starts from zeroed gradients.  Accumulating a local micro-batch gradient on top of a
        buffer that was already summed over ranks and reducing it again would count the earlier micro-batches
        world_size times."""
        if fresh:
            self._reduced = False
        elif self._reduced:
            raise RuntimeError("backward() accumulates onto gradients that were already all-reduced: run all but the "
                               "last micro-batch under `with dp.no_sync():` (or zero_grad() between steps)")

    def _on_layer_done(self, layer_name: str):
        if self._sync:
            self.bucketer.on_layer_done(layer_name)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate locally; the first backward
        outside it all-reduces the accumulated sum (same contract as torch DDP.no_sync)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def finish_gradients(self):
        """Call after loss.backward(): all buckets reduced (summed) when this returns (stream-wise)."""
        if not self._sync:
            return
        self.bucketer.finish()
        self._reduced = True

    def sum_scalars(self, t: torch.Tensor) -> torch.Tensor:
        """Sum over ranks of a small tensor (e.g. (sum of per-batch losses, batch count) pairs: dividing the reduced
        sums once weights every batch equally even when the ranks' shards are uneven or empty)."""
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def average_scalars(self, t: torch.Tensor) -> torch.Tensor:
        """Mean over ranks of a small tensor of metrics (loss / SSIM logging, validation loss that
        drives ReduceLROnPlateau identically on all ranks)."""
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world
