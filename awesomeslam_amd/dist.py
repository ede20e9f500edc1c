"""Multi-GPU layout: one process per GPU, trajectories sharded by rank, poses gathered once at the end.

Independent filters ("trajectories") are the only axis the path shards on -- callbacks of one filter are a
strict recurrence (ekf.cpp:297,310) -- so there is no data-path collective while filtering:
rank r owns trajectories [r*B_local, (r+1)*B_local) and the only exchange is one all_gather of the pose
streams when the replay is done (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
A single filter does not shard: "replicas only".
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    Returns (rank, world, local_rank).  world == 1 needs no group."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # ASLAM_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks (the ranks then
            # share devices and CUDA tensors are staged through host memory for the collectives)
            backend = os.environ.get("ASLAM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_local, rank):
    """Global trajectory indices owned by `rank` under weak scaling (n_local per rank)."""
    return range(rank * n_local, (rank + 1) * n_local)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def _coll_device(device):
    """gloo moves CPU tensors; nccl (RCCL) needs them on the rank's GPU"""
    return "cpu" if dist.get_backend() == "gloo" else device


def max_over_ranks(x, device="cpu"):
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device="cpu"):
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_poses(poses):
    """poses: [B_local, T, 3] on this rank -> [world*B_local, T, 3] on every rank, in global trajectory order."""
    if not dist.is_initialized():
        return poses
    world = dist.get_world_size()
    src = poses.contiguous()
    if dist.get_backend() == "gloo" and src.is_cuda:
        src = src.cpu()  # rehearsal: staged through host memory
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)  # concatenation along dim 0, in rank order
    return out.to(poses.device)


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
