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


# ---- the one collective of the env path (SURVEY.md 8(e) "Collective") ------------------------------------------------------------
# Every quantity of `step()` is indexed by env and the envs shard over the ranks with nothing shared, so the only numbers that cross
# GPUs are the LOGGING means: `Episode_Reward/*`, `Episode_Termination/*`, `Metrics/*`, `Curriculum/terrain_levels`.  The reference
# logs them per rank (rsl_rl: rank 0 alone writes, i.e. the means of ITS 4096 envs stand for the job); here they can be the job's:
# one SUM all-reduce of a packed <= 64-float vector - per-term episode sums over the envs that were reset, the reset count, the
# termination counts, the metric sums, the sum of the terrain levels and the env count - issued on a SIDE stream, so that 256 bytes
# of pure latency (10 - 20 us over xGMI) never sit on the step's critical path; the means are taken after the reduction.
LOG_SLOT_TERRAIN_SUM, LOG_SLOT_NUM_ENVS = 6, 7  # spare words of a log slot (csrc/env_tables.h: 0 count, 1 - 3 terminations, 4 - 5 metrics, 8.. term sums)
LOG_SLOT_CMD_LIN, LOG_SLOT_CMD_ANG, LOG_SLOT_RANKS, LOG_SLOT_FRESH = 60, 61, 62, 63  # (8 + MAX_T = 48 <= 60; 63 = LOG_FRESH of the kernel)
_side_streams: dict = {}


class EpisodeLogFuture:
    """Handle of a `reduce_episode_log` in flight.  `result()` orders the caller's stream behind the collective (no host sync on the RCCL
    path) and returns the dict the reference calls `extras["log"]`, with the means taken over every rank's envs; `vector()` the reduced
    packed vector itself."""

    def __init__(self, env, vec, work, side):
        self._env, self._vec, self._work, self._side, self._out = env, vec, work, side, None

    def vector(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._side is not None:
            torch.cuda.current_stream(self._vec.device).wait_stream(self._side)
            self._side = None
        return self._vec

    def result(self) -> dict:
        if self._out is not None:
            return self._out
        e, s = self._env, self.vector()
        cnt = torch.clamp(s[0], min=1.0)
        out = {"Episode_Reward/" + name: s[8 + i] / cnt / e.max_episode_length_s for i, name in enumerate(e.desc.reward_names)}
        out["Metrics/base_velocity/error_vel_xy"] = s[4] / cnt
        out["Metrics/base_velocity/error_vel_yaw"] = s[5] / cnt
        out["Episode_Termination/time_out"] = s[1]
        out["Episode_Termination/terrain_out_of_bounds"] = s[2]
        if e.desc.task.term_illegal_contact:
            out["Episode_Termination/illegal_contact"] = s[3]
        if e.desc.terrain.curriculum and not e.desc.terrain.is_plane:
            out["Curriculum/terrain_levels"] = s[LOG_SLOT_TERRAIN_SUM] / torch.clamp(s[LOG_SLOT_NUM_ENVS], min=1.0)
        ranks = torch.clamp(s[LOG_SLOT_RANKS], min=1.0)
        if e.desc.task.cur_cmd_lin:  # the ranks' live upper bounds (each rank runs its own curriculum decision): their mean
            out["Curriculum/command_levels_lin_vel"] = s[LOG_SLOT_CMD_LIN] / ranks
        if e.desc.task.cur_cmd_ang:
            out["Curriculum/command_levels_ang_vel"] = s[LOG_SLOT_CMD_ANG] / ranks
        out["episodes"], out["num_envs"] = s[0], s[LOG_SLOT_NUM_ENVS]  # (not reference keys: how many episodes / envs stand behind the means)
        self._out = out
        return out


def pack_episode_log(env, steps: int = 1) -> torch.Tensor:
    """This rank's packed vector.  `steps` = 1: the log slot of the most recent step that reset an env (csrc/env_terms.h step_front: the
    device-side ring resolves an empty slot to its predecessor).  `steps` > 1: the SUM over the last `steps` steps of the slots whose step
    itself reset somebody (their LOG_FRESH word; an inherited slot repeats its predecessor and is skipped) - every episode that ended
    inside the window counts once, which is what rsl_rl's logger gets by averaging the `ep_infos` of all steps of a rollout; a
    window without a single reset falls back to the `steps` = 1 form.  Spare words: the sum of the terrain levels, the env count, the
    live command-range upper bounds of the command_levels_* curricula and a 1 per rank (so that the reduced vector knows the world size)."""
    e = env.unwrapped if hasattr(env, "unwrapped") else env
    k = e._native.log_slot()
    log = e._bufs["LOG"]
    R = log.shape[0]
    cur, prev = log[k].sum(0), log[(k - 1) % R].sum(0)  # (a slot is RL_LOG_PARTS partial rows)
    vec = torch.where(cur[0] > 0, cur, prev).clone()
    steps = max(1, min(int(steps), R - 4, int(e.common_step_counter)))  # (the device keeps the last R - 3 steps)
    if steps > 1:
        idx = torch.tensor([(k - i) % R for i in range(steps)], device=log.device)
        rows = log[idx].sum(1)                                   # [steps, LOG_SIZE]
        fresh = (rows[:, LOG_SLOT_FRESH] > 0).to(rows.dtype)[:, None]
        window = (rows * fresh).sum(0)
        vec = torch.where(window[0] > 0, window, vec)
    if e.desc.terrain.curriculum and not e.desc.terrain.is_plane:
        vec[LOG_SLOT_TERRAIN_SUM] = e.terrain_levels.float().sum()
    vec[LOG_SLOT_NUM_ENVS] = float(e.num_envs)
    vec[LOG_SLOT_FRESH] = 0.0
    if e.desc.task.cur_cmd_lin:
        vec[LOG_SLOT_CMD_LIN] = e.command_levels[1]
    if e.desc.task.cur_cmd_ang:
        vec[LOG_SLOT_CMD_ANG] = e.command_levels[5]
    vec[LOG_SLOT_RANKS] = 1.0
    return vec


def reduce_episode_log(env, group=None, steps: int | None = None) -> EpisodeLogFuture:
    """SUM all-reduce of the packed episode-metric vector over the ranks of `group` (a torch.distributed process group; None: the default
    group when one is initialised, else this rank alone), off the caller's stream.  Collective: every rank of the group calls it.
    `steps`: the window the episode statistics cover (pack_episode_log); None = the env steps since this env's previous call (a rollout,
    when called once per iteration), 1 on the first call."""
    import torch.distributed as dist

    e = env.unwrapped if hasattr(env, "unwrapped") else env
    if steps is None:
        last = getattr(e, "_log_reduced_at", None)
        steps = 1 if last is None else max(1, int(e.common_step_counter) - last)
    e._log_reduced_at = int(e.common_step_counter)
    vec = pack_episode_log(e, steps)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return EpisodeLogFuture(e, vec, None, None)
    if dist.get_backend(group) == "gloo":  # the share-GPU self-test / CPU tier: host tensors (a host sync - never the production path)
        host = vec.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        return EpisodeLogFuture(e, host.to(vec.device), None, None)
    side = _side_streams.get(vec.device)
    if side is None:
        side = _side_streams[vec.device] = torch.cuda.Stream(device=vec.device)
    side.wait_stream(torch.cuda.current_stream(vec.device))  # the vector was packed on the caller's stream
    with torch.cuda.stream(side):
        vec.record_stream(side)
        work = dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group, async_op=True)  # RCCL over xGMI
    return EpisodeLogFuture(e, vec, work, side)
