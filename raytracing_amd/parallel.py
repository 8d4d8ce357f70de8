"""Partitioning of one render across the ranks of a node, and the one collective that merges it.

Replaces the reference's tile thread pool (src/parallel.rs:5-62, src/integrators/integrator.rs:39-105):
every (pixel, sample) path is independent, the scene is replicated on every GPU, and the only
exchange is the final merge of the per-pixel XYZ sums (W*H*3 floats), done with torch.distributed
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

  samples : rank r renders samples [r*spp/world, (r+1)*spp/world) of every pixel; merge = SUM.
  tiles   : rank r renders the 16x16 tiles k with k % world == r, all samples; the buffers are
            disjoint (zeros elsewhere), so the same SUM merges them exactly and the result is
            bit-identical to a single-GPU render.
"""
import torch
import torch.distributed as dist

SAMPLES, TILES = "samples", "tiles"


def sample_range(spp, rank, world):
    """Contiguous, balanced split of [0, spp) (sizes differ by at most one)."""
    base, extra = divmod(spp, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def rank_share(kind, spp, rank, world):
    """Keyword arguments for api.render_params describing this rank's share of the render."""
    if world <= 1:
        return {}
    if kind == SAMPLES:
        return {"sample_range": sample_range(spp, rank, world)}
    if kind == TILES:
        return {"tile_rank": rank, "tile_world": world}
    raise ValueError("unknown partition %r" % (kind,))


def merge_sums(xyz_sum, dst=0, group=None):
    """Sums every rank's per-pixel XYZ buffer onto rank `dst` (one reduce of W*H*3 floats)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if xyz_sum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo has no reduce for device tensors (a rehearsal with several ranks on one GPU): through the host
            host = xyz_sum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:
                xyz_sum.copy_(host)
        else:
            dist.reduce(xyz_sum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return xyz_sum


def render_distributed(accumulate, tonemap, n_values, spp, kind=SAMPLES, device="cpu", group=None):
    """One render over all ranks of the default process group.

    accumulate(xyz_sum_tensor, share_kwargs) adds this rank's share into the zeroed float32 buffer;
    tonemap(xyz_sum_tensor, spp) turns the merged sums into the 8-bit image (called on rank 0 only).
    Returns the image on rank 0 and None elsewhere.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    acc = torch.zeros(n_values, dtype=torch.float32, device=device)
    accumulate(acc, rank_share(kind, spp, rank, world))
    merge_sums(acc, 0, group)
    return tonemap(acc, spp) if rank == 0 else None
