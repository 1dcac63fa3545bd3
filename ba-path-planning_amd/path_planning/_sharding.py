"""Multi-GPU plumbing of the SCP hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL).

What shards: the O(N^2 K) pairwise passes.  Rank g owns the contiguous range of lexicographic pair indices
[q_begin, q_end) at every time step -- balanced by construction because it is cut in PAIR space, not in
agent space -- and the agents [i_begin, i_end) for the kinematics.  One exchange step per SCP iteration:
``allgather_positions`` (the per-shard trajectories, so that every rank sees all neighbours) and, for the
joint QP, ``allgather_rows`` of each rank's compact working rows.  Everything here works on CPU tensors with
the gloo backend too (tests/test_sharding_cpu.py runs it with world_size 2).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class Shard:
    """rank / world_size view of an N-agent problem; world_size == 1 needs no process group."""

    def __init__(self, N, rank=0, world_size=1, group=None, run_collectives_alone=False):
        self.N, self.rank, self.world, self.group = int(N), int(rank), int(world_size), group
        # a world of ONE rank normally skips every collective; run_collectives_alone sends them through the process group anyway
        # (a one-rank RCCL communicator): the device-side ("nccl") code path can then run on a single GPU (tests)
        self.alone = self.world == 1 and not run_collectives_alone
        if (self.world > 1 or run_collectives_alone) and not dist.is_initialized():
            raise RuntimeError("world_size > 1 needs an initialised torch.distributed process group")
        self.pairs = self.N * (self.N - 1) // 2
        # host wall time spent inside the exchanges (staging copies of a gloo rehearsal included) and their count: what a
        # multi-rank step adds to the single-rank one (bench.py reports it per step)
        self.comm_seconds, self.comm_calls = 0.0, 0

    # ---- partitions ----------------------------------------------------------------------------
    def pair_range(self, rank=None):
        r = self.rank if rank is None else rank
        return self.pairs * r // self.world, self.pairs * (r + 1) // self.world

    def agent_range(self, rank=None):
        r = self.rank if rank is None else rank
        return self.N * r // self.world, self.N * (r + 1) // self.world

    # ---- collectives ---------------------------------------------------------------------------
    def _host_staged(self, tensor):
        """gloo moves host memory: CUDA tensors are staged through the host (rehearsals on one GPU); with RCCL
        ("nccl") the collectives run on the device buffers directly."""
        return tensor.is_cuda and dist.get_backend(self.group) == "gloo"

    def _timed(fn):  # noqa: N805  (decorator in the class body)
        import functools
        import time

        @functools.wraps(fn)
        def wrapper(self, *a, **kw):
            if self.alone or getattr(self, "_in_exchange", False):
                return fn(self, *a, **kw)
            self._in_exchange = True
            t0 = time.perf_counter()
            try:
                return fn(self, *a, **kw)
            finally:
                self.comm_seconds += time.perf_counter() - t0
                self.comm_calls += 1
                self._in_exchange = False
        return wrapper

    @_timed
    def allgather_positions(self, local):
        """local: (n_local, K, D) trajectories of this rank's agents -> (N, K, D) on every rank."""
        if self.alone:
            return local
        if self._host_staged(local):
            return self.allgather_positions(local.cpu()).to(local.device)
        counts = [self.agent_range(r)[1] - self.agent_range(r)[0] for r in range(self.world)]
        if len(set(counts)) == 1:
            out = torch.empty((self.N,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
            return out
        width = max(counts)
        pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)

    @_timed
    def allgather_rows(self, rows, w_eta, w_l):
        """Concatenate every rank's compact rows (ids (n,), eta (n, D), l (n,)) in rank order.

        Two collectives of fixed, padded shape (ids as int64 in their own message -- never reinterpreted as floating
        point -- and [eta | l] as float64) plus one gather of the counts; the padding is cut on the device, the only
        host read is the list of counts."""
        if self.alone:
            return rows, w_eta, w_l
        if self._host_staged(rows):
            dev = rows.device
            r, e, l = self.allgather_rows(rows.cpu(), w_eta.cpu(), w_l.cpu())
            return r.to(dev), e.to(dev), l.to(dev)
        k = int(rows.numel())
        D = w_eta.shape[1] if w_eta.dim() == 2 else 1
        n = torch.tensor([k], dtype=torch.int64, device=rows.device)
        counts_t = torch.empty(self.world, dtype=torch.int64, device=rows.device)
        dist.all_gather_into_tensor(counts_t, n, group=self.group)
        counts = counts_t.tolist()  # one host read: the shapes of the result depend on it
        width = max(max(counts), 1)
        ids = torch.zeros(width, dtype=torch.int64, device=rows.device)
        vals = torch.zeros(width, D + 1, dtype=torch.float64, device=rows.device)
        if k:
            ids[:k] = rows.to(torch.int64)
            vals[:k, :D] = w_eta.reshape(k, D)
            vals[:k, D] = w_l
        all_ids = torch.empty(self.world * width, dtype=torch.int64, device=rows.device)
        all_vals = torch.empty(self.world * width, D + 1, dtype=torch.float64, device=rows.device)
        dist.all_gather_into_tensor(all_ids, ids, group=self.group)
        dist.all_gather_into_tensor(all_vals, vals, group=self.group)
        keep = torch.cat([torch.arange(r * width, r * width + c, device=rows.device) for r, c in enumerate(counts)])
        all_ids, all_vals = all_ids[keep], all_vals[keep]
        return all_ids.contiguous(), all_vals[:, :D].contiguous(), all_vals[:, D].contiguous()

    @_timed
    def allgather_ids(self, rows, extra=None):
        """Every rank's row ids (int64, any length) merged in ascending order -- the order a single rank would have found
        them in, so the replicated QP builds the same working set bit for bit on every world size -- plus, optionally, the
        max over ranks of one float (`extra`: the violations pass's max violation).  ONE collective: each rank sends
        [count, bits of extra, ids padded to the capacity every rank agreed on]; a second, longer one only when some rank's
        list outgrew that capacity.  Returns (ids, max_extra)."""
        if self.alone:
            return rows, extra
        dev = rows.device
        staged = self._host_staged(rows)
        work = rows.cpu() if staged else rows
        k = int(work.numel())
        cap = getattr(self, "_ids_cap", 4096)
        ex = float("-inf") if extra is None else float(extra)
        while True:
            msg = torch.zeros(cap + 2, dtype=torch.int64, device=work.device)
            msg[0] = k
            msg[1] = torch.tensor([ex], dtype=torch.float64).view(torch.int64)[0]
            if k:
                msg[2:2 + min(k, cap)] = work[: min(k, cap)]
            out = torch.empty(self.world * (cap + 2), dtype=torch.int64, device=work.device)
            dist.all_gather_into_tensor(out, msg, group=self.group)
            out = out.view(self.world, cap + 2)
            head = out[:, :2].cpu()  # one host read: counts and the extras
            counts = head[:, 0].tolist()
            if max(counts) <= cap:
                break
            cap = self._ids_cap = 2 * max(counts)  # every rank sees the same counts: the same decision, the same retry
        self._ids_cap = cap
        parts = [out[r, 2:2 + c] for r, c in enumerate(counts) if c]
        ids = torch.sort(torch.cat(parts)).values if parts else work[:0]
        mx = float(head[:, 1].contiguous().view(torch.float64).max().item())
        return (ids.to(dev) if staged else ids).contiguous(), (mx if extra is not None else None)

    @_timed
    def broadcast(self, tensor, src=0):
        if not self.alone:
            if self._host_staged(tensor):
                host = tensor.cpu()
                dist.broadcast(host, src=src, group=self.group)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def broadcast_ints(self, values, src=0):
        """Rank `src`'s integers on every rank: loop-control decisions (iteration counts, statuses) must be taken
        from ONE rank, otherwise fp-atomics noise in the replicated QP could make ranks leave a loop at different
        trip counts and the next collective would hang."""
        if self.alone:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self._dev())
        dist.broadcast(t, src=src, group=self.group)
        return [int(v) for v in t.tolist()]

    def all_min(self, value: float) -> float:
        if self.alone:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t.item())

    def all_max(self, value: float) -> float:
        if self.alone:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def all_min_int(self, value: int) -> int:
        """min over ranks of a non-negative integer < 2^63 (first-violation row ids; UINT64_MAX -> 2^63-1)."""
        if self.alone:
            return value
        v = min(int(value), (1 << 63) - 1)
        t = torch.tensor([v], dtype=torch.int64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def _dev(self):
        if dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")
