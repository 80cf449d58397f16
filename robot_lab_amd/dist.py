"""One process per GPU: what a rank is, which device it owns, and the learner-side collectives of a multi-GPU run.

The reference's multi-GPU entry is `python -m torch.distributed.run --nnodes=1 --nproc_per_node=N scripts/reinforcement_learning/rsl_rl/
train.py --task=<id> --headless --distributed` (`README.md:323-337`): `train.py:143-150` puts rank r's environments on
`cuda:{app_launcher.local_rank}` with seed `agent_cfg.seed + app_launcher.local_rank`, and rsl_rl's runner keeps the N learners in step
(parameters broadcast from rank 0 before the first iteration, the flattened gradient all-reduced SUM / world after every backward, the
KL statistic of the adaptive schedule averaged over the ranks, logs and checkpoints from rank 0 only - SURVEY.md B10).  The environments
themselves shard with no data-path collective; everything in this file is host-side plumbing around `step()`.

`RL_SHARE_GPU=1` is a self-test aid for boxes with fewer GPUs than ranks (a `gpurun` box has ONE): logical `cuda:k` resolves to physical
`cuda:(k mod device_count)` and the collectives run over gloo (RCCL refuses two ranks on one device).  Never set by a production launch.
"""
from __future__ import annotations

import os

import torch


def rank_info() -> tuple[int, int, int]:
    """(global rank, local rank, world size) as `torch.distributed.run` exports them; (0, 0, 1) outside a launch."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def share_gpu() -> bool:
    return os.environ.get("RL_SHARE_GPU") == "1"


def physical_device(device: str | torch.device) -> torch.device:
    """The device a logical `cuda:k` lives on.  Identity, except under RL_SHARE_GPU=1 where ranks beyond the visible devices wrap around."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is not None and share_gpu() and torch.cuda.is_available():
        return torch.device("cuda", dev.index % torch.cuda.device_count())
    return dev


class LearnerGroup:
    """rsl_rl's multi-GPU contract for a PPO learner (`OnPolicyRunner._configure_multi_gpu`, `PPO.broadcast_parameters`,
    `PPO.reduce_parameters` of rsl-rl-lib 3.0.1 - third-party, absent here, restated from SURVEY.md B10).  With a world of one every
    method is a no-op, so the single-GPU path carries no collective."""

    def __init__(self, device: str):
        self.rank, self.local_rank, self.world_size = rank_info()
        self.enabled = self.world_size > 1
        self.device = str(device)
        if not self.enabled:
            return
        # rsl_rl raises the same way: a rank whose learner does not sit on its own GPU would all-reduce with somebody else's device
        if self.device != f"cuda:{self.local_rank}":
            raise ValueError(f"multi-GPU training: rank {self.rank} (local rank {self.local_rank}) was given device '{self.device}', expected "
                             f"'cuda:{self.local_rank}' - launch the reference's train.py with --distributed (train.py:143-150)")
        import torch.distributed as dist

        if not dist.is_initialized():
            if share_gpu() or not torch.cuda.is_available():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world_size)
            else:
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world_size,  # "nccl" IS RCCL on ROCm
                                        device_id=torch.device("cuda", self.local_rank))
        self.backend = dist.get_backend()

    @property
    def is_main(self) -> bool:
        """rank 0 logs and writes checkpoints; the others train silently (rsl_rl `disable_logs`)."""
        return not self.enabled or self.rank == 0

    def broadcast_parameters(self, module: torch.nn.Module) -> None:
        """every learner starts from rank 0's initialisation (their seeds differ: train.py:148)"""
        if not self.enabled:
            return
        import torch.distributed as dist

        with torch.no_grad():
            params = list(module.parameters())
            flat = torch.cat([p.reshape(-1) for p in params])
            dist.broadcast(flat, src=0)
            off = 0
            for p in params:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def reduce_gradients(self, module: torch.nn.Module) -> None:
        """ONE all-reduce of the flattened gradient per mini-batch (1.2 MB for the A1 networks - a single ring pass over xGMI, not a
        bucket per layer), SUM then / world: every rank applies the same Adam step, so the replicas never drift."""
        if not self.enabled:
            return
        import torch.distributed as dist

        grads = [p.grad for p in module.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= self.world_size
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def mean(self, value: torch.Tensor) -> torch.Tensor:
        """a scalar statistic (the KL of the adaptive schedule) averaged over the ranks: the same number everywhere, hence the same
        learning-rate decision on every rank"""
        if not self.enabled:
            return value
        import torch.distributed as dist

        t = value.detach().clone().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return (t / self.world_size).reshape(())
