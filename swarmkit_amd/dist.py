"""One process per GPU: the process group and the measurement protocol of bench.py (a barrier on both sides of the timed
region, the max-over-ranks elapsed time, the sum of the units every rank processed).

The data path between ranks — node-range shards of ONE cluster exchanging their proposals with an RCCL all-gather per
round — is swarmkit_amd/shard.py (RankShard) on top of the group this module opens. `--parallelism replicas` of bench.py
(every rank its own cluster, no data-path collective) uses the measurement protocol only."""
import os


class Ranks:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = None
        self.device = device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                kw["device_id"] = self.device
            else:
                self.device = torch.device("cpu")
            import datetime
            kw["timeout"] = datetime.timedelta(minutes=10)
            try:
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
                if backend == "nccl":      # connect now, so that a broken RCCL set-up shows here and not inside the timed region
                    dist.barrier()
            except Exception as e:         # measurement protocol only (no data-path collective): keep the run alive over gloo
                if backend != "nccl":
                    raise
                import sys
                print(f"swarmkit_amd.dist: RCCL process group failed ({e!r}); using gloo for the timing protocol", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                self.device = torch.device("cpu")
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world, timeout=kw["timeout"])
            self.backend = backend
            self.dist = dist

    def replica_seed(self, base):
        """Rank 0 keeps the canonical seed of the workload (so N=1 and rank 0 of N>1 run the same cluster);
        every other rank gets its own cluster of the same shape."""
        return None if self.rank == 0 else (int(base) + 1000 * self.rank + 3)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import ctypes
        import sys
        import torch
        # (the timing's last collective, in front of rank 0's result line: every rank empties its stdout buffers here — native
        # libraries' banners included, NCCL_DEBUG=VERSION — so that the line is the last thing the job prints)
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return int(x)
        import torch
        t = torch.tensor([int(x)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def whole_job_rate(ranks, units_this_rank, elapsed_this_rank):
    """bench.py's `value`: units all ranks processed / max-over-ranks time."""
    total = ranks.sum_over_ranks(units_this_rank)
    t = ranks.max_over_ranks(elapsed_this_rank)
    return total / t if t > 0 else 0.0, total, t
