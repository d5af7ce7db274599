"""Multi-GPU plumbing: independent replicas, one process per GPU.

A single KKT system per SQP iteration does not shard, so N GPUs run N
independent problems (BASELINE.json configs[4], the reference's thread_test.c
pattern).  There is no data-path collective: torch.distributed (backend "nccl"
= RCCL on ROCm, "gloo" in the CPU tests) is used only for the start/stop
barrier and the max-over-ranks reduction of the elapsed time.
"""
from __future__ import annotations

import os


class Replicas:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                # RCCL needs one GPU per rank; with fewer GPUs than ranks (a 1-GPU box exercising the
                # N-rank launcher) the ranks share devices and synchronise over gloo
                enough = torch.cuda.is_available() and torch.cuda.device_count() >= self.world
                backend = "nccl" if enough else "gloo"
                if torch.cuda.is_available() and not enough:
                    self.local_rank %= torch.cuda.device_count()
            self.backend = backend
            kwargs = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kwargs["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kwargs)
            self.dist = dist

    # every rank works on its own problem: the seed is the rank (configs[4]: seeds 0..7)
    def problem_seed(self) -> int:
        return self.rank

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch

        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, value: float) -> list:
        """Every rank's value, in rank order (per-GPU min / max reporting)."""
        if self.dist is None:
            return [float(value)]
        import torch

        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")
        mine = torch.tensor([value], dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return [float(p.item()) for p in parts]

    def aggregate_rate(self, steps_per_rank: int, t_max: float) -> float:
        """Whole-job throughput: units processed by all ranks / slowest rank's time."""
        return self.world * steps_per_rank / t_max

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
