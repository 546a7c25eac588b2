"""Path sharding across the GPUs of one node: one process per GPU.  Two transports for the two
tiny exchanges below: RcclPricer (RCCL called from inside libomc.so, no torch) and ShardedPricer
(torch.distributed: backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed code at all (its only parallelism is a ProcessPoolExecutor
over S0 values, options_model_3.py:1053).  Paths are independent, so they shard by antithetic
pair; the Philox counter carries the GLOBAL pair index, hence the simulated path set -- and
the price -- does not depend on the number of shards.  Only two things cross GPUs:

  * regression moments: the two-pass flow's moments do not depend on exercise decisions, so
    the whole [n_steps+1][8] table goes in ONE all-reduce (~16 KB at 252 steps); the per-step
    flows need 8 doubles per time step (sequential by data dependence);
  * the final {sum, sumsq, counts}: one all-reduce of 6 doubles.
"""
from __future__ import annotations

import math


def shard(n_paths_global: int, world: int, rank: int, antithetic: bool = True):
    """-> (n_paths_local, pair_offset).  Shards are equal; pairs stay together."""
    unit = 2 if antithetic else 1
    if n_paths_global % (unit * world):
        raise ValueError(f"n_paths={n_paths_global} must be a multiple of {unit * world} "
                         f"({'pairs' if antithetic else 'paths'} x ranks)")
    local = n_paths_global // world
    return local, rank * (local // unit)


SUM_KEYS = ("sum", "sumsq", "n_paths", "n_exercised", "n_zero", "sum_nitm")


def merge(local: dict, all_reduce_sum) -> dict:
    """Combine per-shard results.  `all_reduce_sum(list[float]) -> list[float]`."""
    tot = all_reduce_sum([float(local[k]) for k in SUM_KEYS])
    out = dict(zip(SUM_KEYS, tot))
    n = out["n_paths"]
    out["price"] = out["sum"] / n
    var = max(out["sumsq"] / n - out["price"] ** 2, 0.0)
    out["std"] = math.sqrt(var)
    out["stderr"] = math.sqrt(var / n)
    out["zero_prob"] = out["n_zero"] / n
    for k in ("n_paths", "n_exercised", "n_zero", "sum_nitm"):
        out[k] = int(round(out[k]))
    return out


class _DevPtr:
    """Expose a raw device pointer to torch (no ownership) via __cuda_array_interface__."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


class _ShardedBase:
    """Shared by the two transports: shard the global problem, price the local shard through a
    context whose library calls already return GLOBAL sums (hook or native communicator)."""

    ctx = None
    rank = 0
    world = 1
    _ffi = None

    def _enter(self):  # context manager around library calls (torch transport: current stream)
        import contextlib
        return contextlib.nullcontext()

    def price_american(self, n_paths_global: int, **kw) -> dict:
        """kw: arguments of _ffi.make_params except n_paths / pair_offset."""
        anti = kw.get("antithetic", True)
        n_local, off = shard(n_paths_global, self.world, self.rank, anti)
        p = self._ffi.make_params(n_paths=n_local, pair_offset=off, **kw)
        with self._enter():
            res = self.ctx.price_american(p)
        # with world > 1 the sums in `res` are global already
        out = merge(res, lambda v: v)
        out["local"] = res
        return out

    def price_american_ols7(self, n_paths_global: int, **kw) -> dict:
        """Regressor "ols7" sharded: this rank's paths, the job's fit and sums (omc_price_american_ols7 on a distributed
        context merges the ranks' co-moments and adds their result sums)."""
        anti = kw.get("antithetic", True)
        n_local, off = shard(n_paths_global, self.world, self.rank, anti)
        p = self._ffi.make_params(n_paths=n_local, pair_offset=off, **kw)
        with self._enter():
            return self.ctx.price_american_ols7(p)

    def price_american_seq(self, n_paths_global: int, streams, **kw) -> list:
        """len(streams) pricings (one Philox stream id each) enqueued back to back, one wait at the end;
        the all-reduces are stream-ordered, so no rank waits on the host in between."""
        anti = kw.get("antithetic", True)
        n_local, off = shard(n_paths_global, self.world, self.rank, anti)
        ps = [self._ffi.make_params(n_paths=n_local, pair_offset=off, stream=int(s), **kw) for s in streams]
        with self._enter():
            res = self.ctx.price_american_seq(ps)
        outs = []
        for r in res:
            out = merge(r, lambda v: v)
            out["local"] = r
            outs.append(out)
        return outs

    def close(self):
        self.ctx.close()


# ---- who ran where: every rank's (device ordinal, PCI bus id, time per step), gathered through a SUM all-reduce ----------
def pci_to_number(pci_bus_id: str) -> float:
    """"0000:c1:00.0" -> an integer below 2**40 (exact in a double): domain:bus:device.function."""
    try:
        dom, bus, rest = pci_bus_id.strip().split(":")
        dev, fn = rest.split(".")
        return float((int(dom, 16) << 24) | (int(bus, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16))
    except Exception:
        return -1.0


def number_to_pci(x: float) -> str:
    v = int(x)
    if v < 0:
        return "unknown"
    return f"{v >> 24:04x}:{(v >> 16) & 0xff:02x}:{(v >> 8) & 0xff:02x}.{v & 0xff:x}"


RANK_ROW = 5  # doubles per rank: [present, device ordinal, pci number, ms per step, rank]


def gather_rank_table(allreduce_sum, rank: int, world: int, device: int, pci_bus_id: str, ms_per_step: float) -> list:
    """Every rank contributes its row at its own offset of a zero vector; one SUM all-reduce (the transport the pricing
    itself uses: native communicator, torch.distributed or a test double) hands the whole table to everybody."""
    vec = [0.0] * (RANK_ROW * world)
    vec[RANK_ROW * rank:RANK_ROW * (rank + 1)] = [1.0, float(device), pci_to_number(pci_bus_id), float(ms_per_step), float(rank)]
    out = list(allreduce_sum(vec))
    return [out[RANK_ROW * r:RANK_ROW * (r + 1)] for r in range(world)]


def check_rank_table(rows, world: int, allow_shared_device: bool = False) -> list:
    """-> [{rank, device, pci_bus_id, ms_per_step}] or ValueError: a rank that contributed no row or twice (the
    communicator does not have `world` distinct ranks), or two ranks on one card (two processes sharing a GPU would
    report a 'scaling' number that is really time slicing).  Every rank holds the same table, so every rank raises."""
    if len(rows) != world:
        raise ValueError(f"rank table has {len(rows)} rows, expected {world}")
    table = []
    for r, (present, device, pci, ms, rk) in enumerate(rows):
        if present != 1.0 or int(rk) != r:
            raise ValueError(f"rank {r} contributed {present:g} rows to the rank table (expected exactly 1): the "
                             f"communicator does not connect {world} distinct ranks")
        table.append(dict(rank=r, device=int(device), pci_bus_id=number_to_pci(pci), ms_per_step=ms))
    if not allow_shared_device:
        seen = {}
        for t in table:
            key = t["pci_bus_id"]
            if key in seen and key != "unknown":
                raise ValueError(f"ranks {seen[key]} and {t['rank']} both run on the card at PCI {key}: one rank per GPU "
                                 f"(check LOCAL_RANK / HIP_VISIBLE_DEVICES of the launcher)")
            seen[key] = t["rank"]
    return table


class RcclUnavailable(RuntimeError):
    """The native communicator cannot be used by this JOB.  Raised on every rank or on none: the ranks vote
    (rendezvous.agree) before and after omc_comm_init, so a caller may fall back to another transport knowing
    that all its peers do the same."""


class Watchdog:
    """Ends the process (exit code 3, message on stderr) unless cancel()led within `seconds`: the one bound
    that also covers a rank stuck INSIDE ncclCommInitRank or a collective, which have no timeout of their own
    and cannot be interrupted from Python.  Used around communicator bring-up and, by bench.py, around a whole
    multi-rank run."""

    def __init__(self, seconds: float, what: str):
        import threading
        self._ev = threading.Event()
        self._t = threading.Thread(target=self._run, args=(float(seconds), what), daemon=True)
        self._t.start()

    def _run(self, seconds, what):
        if not self._ev.wait(seconds):
            import os
            import sys
            print(f"options_model_amd: {what} did not finish within {seconds:.0f} s; exiting", file=sys.stderr, flush=True)
            os._exit(3)

    def cancel(self):
        self._ev.set()


class RcclPricer(_ShardedBase):
    """One per rank, NO torch: the library owns an RCCL communicator (omc_comm_init) and enqueues its
    all-reduces itself.  The unique id travels from rank 0 through options_model_amd.rendezvous.

    Bring-up is a collective decision with a deadline: (1) rank 0 draws the id (or publishes that it cannot);
    (2) every rank votes "I have a context, the library and the id"; unless all do, ALL raise RcclUnavailable;
    (3) omc_comm_init under a watchdog (`init_timeout_s`; a peer that died inside RCCL's bootstrap would
    otherwise block the others for ever); (4) every rank votes on the outcome, and again all continue or all
    raise.  A rank that never votes makes the others raise TimeoutError after `timeout_s`."""

    transport = "rccl-native"
    _generation = 0  # communicators brought up by this process so far (every rank counts the same way): the votes
                     # of one bring-up can never be taken for those of another

    def __init__(self, local_rank: int, rank: int, world: int, tag: str | None = None, timeout_s: float = 120.0,
                 init_timeout_s: float = 300.0):
        from . import _ffi, rendezvous

        self._ffi = _ffi
        RcclPricer._generation += 1
        self._gen = gen = RcclPricer._generation
        self.rank, self.world = int(rank), int(world)
        self.ctx = _ffi.Context(local_rank)
        marker = b"OMC_RCCL_UNAVAILABLE"
        why = []

        def make_uid():  # rank 0; a failure is published too, so that the other ranks do not wait for it
            try:
                return _ffi.comm_unique_id()
            except Exception as e:
                why.append(f"rank 0 has no unique id: {e}")
                return marker.ljust(128, b"\0")

        uid_tag = (rendezvous.default_tag() if tag is None else tag) + f"_g{gen}"
        uid, path = rendezvous.exchange(self.rank, make_uid, 128, uid_tag, timeout_s)
        ok = not uid.startswith(marker)
        if not ok and not why:
            why.append("rank 0 has no unique id (see its log)")
        try:
            all_ok = rendezvous.agree(self.rank, self.world, ok, f"uid{gen}", tag, timeout_s)
        finally:
            if path:
                rendezvous.retire(path)  # every rank has voted, i.e. has read the id
        if not all_ok:
            self.ctx.close()
            raise RcclUnavailable("; ".join(why) or "another rank cannot use the native communicator")
        dog = Watchdog(init_timeout_s, f"rank {self.rank}: omc_comm_init ({self.world} ranks)")
        try:
            self.ctx.comm_init(self.rank, self.world, uid)  # collective
            r, w = self.ctx.comm_info()
            ok = (r, w) == (self.rank, self.world)
            if not ok:
                why.append(f"communicator reports rank {r} of {w}, expected {self.rank} of {self.world}")
        except Exception as e:
            ok = False
            why.append(f"omc_comm_init failed on rank {self.rank}: {e}")
        finally:
            dog.cancel()
        all_ok = rendezvous.agree(self.rank, self.world, ok, f"init{gen}", tag, max(timeout_s, init_timeout_s))
        if not all_ok:
            try:
                if ok:
                    self.ctx.comm_destroy()
            finally:
                self.ctx.close()
            raise RcclUnavailable("; ".join(why) or "omc_comm_init failed on another rank")

    def comm_ranks(self) -> int:
        return self.ctx.comm_info()[1]

    def barrier(self):
        self.ctx.comm_allreduce([0.0])

    def allreduce_max(self, x: float) -> float:
        return float(self.ctx.comm_allreduce([x], "max")[0])

    def allreduce_sum(self, values) -> list:
        return [float(v) for v in self.ctx.comm_allreduce(list(values), "sum")]

    def enable_p2p(self, tag: str | None = None, timeout_s: float = 60.0) -> bool:
        """The per-step flows' moment exchange by direct writes into every peer's mailbox (omc_p2p_*; SURVEY.md
        5.8(b)) instead of an all-reduce per time step.  Collective: every rank exports its mailbox handle, all
        handles travel through the rendezvous directory, every rank maps every peer; the ranks vote after each
        stage and either ALL end up connected (True) or all stay with the collective (False).  The communicator
        keeps doing everything else."""
        from . import rendezvous
        base = (rendezvous.default_tag() if tag is None else tag) + f"_p2p{self._gen}"
        handle, ok = b"\0" * 64, True
        try:
            handle = self.ctx.p2p_export()
        except Exception:
            ok = False
        path = rendezvous.publish(handle, f"{base}_h{self.rank}")
        try:
            if not rendezvous.agree(self.rank, self.world, ok, f"p2p_export{self._gen}", tag, timeout_s):
                if ok:
                    self.ctx.p2p_disconnect()
                return False
            handles = b"".join(rendezvous.fetch(64, f"{base}_h{r}", timeout_s) for r in range(self.world))
            try:
                self.ctx.p2p_connect(self.rank, self.world, handles)
            except Exception:
                ok = False
            if not rendezvous.agree(self.rank, self.world, ok, f"p2p_connect{self._gen}", tag, timeout_s):
                self.ctx.p2p_disconnect()
                return False
            return True
        finally:
            rendezvous._retire_at_exit([path])

    def close(self):
        try:
            try:
                if self.ctx.p2p_status()[0]:
                    self.barrier()  # no rank unmaps its mailbox while a peer could still be writing into it
                self.ctx.p2p_disconnect()
            finally:
                self.ctx.comm_destroy()
        finally:
            self.ctx.close()


class ShardedPricer(_ShardedBase):
    """One per rank over torch.distributed (nccl == RCCL, or gloo for CPU-side rehearsals): the library
    calls back into Python for its all-reduces (omc_set_allreduce_hook)."""

    transport = "torch.distributed"

    def __init__(self, local_rank: int, group=None, force_hook: bool = False):
        import torch
        import torch.distributed as dist

        from . import _ffi

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        # One explicit (non-null) stream shared by the HIP kernels and, through torch's
        # "current stream", by the RCCL collectives: everything is stream-ordered, no host
        # syncs between a moment kernel, its all-reduce and the kernels that consume it.
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = _ffi.Context(local_rank, stream=self.stream.cuda_stream)
        self._ffi = _ffi
        self._alias = {}
        if self.world > 1 or force_hook:  # force_hook: exercise the RCCL path with one rank
            # the library calls the hook for the moment table(s) AND for its 8 result sums, and
            # normalises by n_local * world_size: the returned omc_result is already global
            self.ctx.set_allreduce_hook(self._allreduce_device)
            self.ctx.set_option("world_size", self.world)

    def _enter(self):
        return self.torch.cuda.stream(self.stream)  # the hook's all_reduce sees this stream as current

    def _allreduce_device(self, dptr: int, count: int):
        t = self._alias.get((dptr, count))
        if t is None:  # workspace pointers are stable across pricings of one size
            t = self.torch.as_tensor(_DevPtr(dptr, count), device=self.device)
            self._alias[(dptr, count)] = t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def comm_ranks(self) -> int:
        return self.dist.get_world_size(self.group)

    def barrier(self):
        self.dist.barrier(self.group)

    def allreduce_max(self, x: float) -> float:
        dev = self.device if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = self.torch.tensor([x], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allreduce_sum(self, values) -> list:
        dev = self.device if self.dist.get_backend(self.group) == "nccl" else "cpu"
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().tolist()
