"""Multi-GPU layout of the env stepper: envs are independent, so the path
shards with no data-path collective.  One process per GPU (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests); rank g owns global env ids
[g*N/G, (g+1)*N/G) and seeds them seed0 + global id, so trajectories do not depend on G.
The only exchange is an all-gather of per-env episode returns, once per rollout."""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_total, world_size, rank):
    """Contiguous block of global env ids owned by `rank` (first_env_id, count)."""
    if n_total % world_size:
        raise ValueError("total env count {} must be divisible by world size {}".format(n_total, world_size))
    per = n_total // world_size
    return rank * per, per


def init_process_group(backend, local_rank=0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        return
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)


def _via_cpu():
    import torch.distributed as dist
    return dist.get_backend() == "gloo"


def gather_episode_returns(local_returns, out=None):
    """All-gather of float32[n_local] -> float32[world * n_local], rank-major == global env id order.
    128 KiB at 8 x 4096 envs: latency bound, so it is issued once per rollout, never per step."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_returns if out is None else out.copy_(local_returns)
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local_returns.numel(),), dtype=local_returns.dtype, device=local_returns.device)
    if _via_cpu() and local_returns.is_cuda:       # CPU-side collective (tests / single-GPU dry runs of the N>1 path)
        tmp = torch.empty((world * local_returns.numel(),), dtype=local_returns.dtype)
        dist.all_gather_into_tensor(tmp, local_returns.detach().cpu().contiguous())
        out.copy_(tmp)
        return out
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


def max_over_ranks(seconds, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=None if _via_cpu() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def per_rank(seconds, device=None):
    """Every rank's value of `seconds`, in rank order (one all-gather of a float64): bench.py reports the per-rank step times next to
    their maximum so that a multi-GPU record shows stragglers."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(seconds)]
    t = torch.tensor([seconds], dtype=torch.float64, device=None if _via_cpu() else device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
