"""Process-group helpers with the reference's names (mmf/utils/distributed.py:67-448): one process per
GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm), env:// rendezvous."""
import os

import torch
import torch.distributed as dist


def is_dist_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_initialized() else 0


def is_master():
    return get_rank() == 0


def synchronize():
    """Barrier (distributed.py:67-82)."""
    if is_dist_initialized() and get_world_size() > 1:
        dist.barrier()


def distributed_init_from_env(backend=None):
    """distributed_init (distributed.py:359-391) for a torchrun-style launch: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; binds this process to its GPU and warms
    the communicator with a 1-element all-reduce (C2 in SURVEY.md §2.3)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or is_dist_initialized():
        return get_rank(), get_world_size()
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    use_gpu = torch.cuda.is_available()
    if backend is None:
        # MMF_AMD_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses duplicate devices): used to exercise
        # the multi-rank code path on a single-GPU box
        backend = os.environ.get("MMF_AMD_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if use_gpu:
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    t = torch.zeros(1, device="cuda" if use_gpu else "cpu")
    dist.all_reduce(t)
    return rank, world


def reduce_dict(dictionary):
    """Average a dict of scalar tensors onto rank 0 (distributed.py:219-240, used by Meter)."""
    world = get_world_size()
    if world < 2:
        return dictionary
    with torch.no_grad():
        if len(dictionary) == 0:
            return dictionary
        keys, values = zip(*sorted(dictionary.items()))
        values = torch.stack([v.reshape(()).float() for v in values], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0:
            values /= world
        return {k: v for k, v in zip(keys, values)}


def broadcast_scalar(scalar, src=0, device="cpu"):
    """distributed.py:145-152."""
    if get_world_size() < 2:
        return scalar
    t = torch.tensor(scalar).long().to(device)
    dist.broadcast(t, src)
    return t.item()
