"""Row-sharded rendering across the GPUs of one node: one process per GPU, torch.distributed.

The reference parallelises render_scene by giving each OS thread a contiguous band of rows and
funnelling finished pixels through an mpsc channel into one Screen (src/world.rs:1198-1244).  Here
each RANK owns the rows {j : (j // block_rows) % world_size == rank} (interleaved rather than
contiguous: the top third of Book-1 is sky and costs a third of the ground rows), renders them into
its own HBM through the C ABI, and the tone-mapped shards meet on rank 0 in ONE gather (RCCL over
xGMI when the backend is nccl; every peer has its own link to the root, so a gather -- not a ring
all-reduce -- is the right collective).  Sample streams are keyed by the GLOBAL pixel index, so the
assembled image is bit-identical for every world size.
"""
import numpy as np


def shard_for(rank, world_size, block_rows=1):
    return (rank, world_size, block_rows)


def shard_row_indices(height, shard):
    r, n, b = shard
    return [j for j in range(height) if (j // b) % n == r]


def max_shard_rows(height, world_size, block_rows=1):
    return max(len(shard_row_indices(height, (r, world_size, block_rows))) for r in range(world_size))


def assemble(parts, height, width, world_size, block_rows=1, channels=3):
    """Interleave per-rank shard buffers (flat or [rows, w, c]) back into a [height, width, c] image."""
    first = np.asarray(parts[0])
    img = np.zeros((height, width, channels), dtype=first.dtype)
    for r in range(world_size):
        rows = shard_row_indices(height, (r, world_size, block_rows))
        flat = np.asarray(parts[r]).reshape(-1)[: len(rows) * width * channels]
        img[rows] = flat.reshape(len(rows), width, channels)
    return img


def gather_shards(local, dst=0, group=None):
    """One gather of equally sized shard tensors to `dst`.  Returns the list on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gather_list = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, gather_list=gather_list, dst=dst, group=group)
    return gather_list


def render_scene_distributed(scene, cam, cfg, block_rows=1, want_accum=False, device="cuda"):
    """render_scene across all ranks of the default process group; rank 0 returns (rgb8[h,w,3], accum or None)."""
    import torch
    import torch.distributed as dist
    from . import api
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    h, w = api.image_height(cfg), cfg.image_width
    shard = shard_for(rank, world, block_rows)
    rows_max = max_shard_rows(h, world, block_rows)
    d_rgb8 = torch.zeros(rows_max * w * 3, dtype=torch.uint8, device=device)
    d_accum = torch.zeros(rows_max * w * 3, dtype=torch.float64, device=device) if want_accum else None
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(cam, cfg, shard=shard, d_accum=d_accum.data_ptr() if want_accum else 0,
                        d_rgb8=d_rgb8.data_ptr(), stream=stream)
    parts = gather_shards(d_rgb8)
    parts_a = gather_shards(d_accum) if want_accum else None
    if rank != 0:
        return None, None
    rgb8 = assemble([p.cpu().numpy() for p in parts], h, w, world, block_rows)
    accum = assemble([p.cpu().numpy() for p in parts_a], h, w, world, block_rows) if want_accum else None
    return rgb8, accum
