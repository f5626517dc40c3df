"""Data-parallel gradient exchange: one process per GPU, batch sharded, parameters replicated.

The reference has no distributed code; `north_star` shards the batch over the 8 MI355X of a node.
Every sample's forward/residual/loss/backward is independent (GroupNorm/LayerNorm/attention are per-sample), so
the only exchange is the gradient average.  The engine writes all used gradients into ONE flat fp32 buffer in its
canonical parameter order (35.7 MB for the Darcy model, 521 MB for the mechanics model), so there is no bucketing
of small tensors and no per-parameter hook.  `GradientExchange` overlaps the exchange with backward:

* the engine runs its deferred gradient reduction in three phases - after the decoder half (`ups.*`, `final_conv.*`),
  after the encoder half (`downs.*`, `mid_*`), at the end (time MLP, FiLM linears, `init_conv`) - and records a HIP
  event after each (`pidm_unet_set_grad_events`); in the flat buffer those are three contiguous ranges;
* `allreduce()` (called right after `loss.backward()`, which only ENQUEUES the backward kernels) issues one RCCL
  all-reduce per range on a side stream that waits for that range's event, so the decoder's gradients travel over
  xGMI while the encoder half of backward is still computing; the caller's stream then waits for the side stream.
  xGMI is point-to-point (ring collectives are per-link bound): three large messages, not many small ones.

Fallbacks that keep the result identical: more than one backward per step (two activation tapes of
x0_estimation='sample', gradient accumulation) or a non-GPU backend (the gloo tests) -> one all-reduce per range on
the caller's stream after backward.  RNG: each rank seeds its own (t, eps) stream; equivalence with the single-process
global-batch step is tested with injected (t, eps) in tests/test_data_parallel.py.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from ._engine import get_engine
from ._lib import vp


def flat_gradient_buffers(model):
    """The flat gradient buffer(s) `p.grad` alias: the primary engine's (slot 0), one per image size.  Engines of other
    slots (second activation tape) add their result INTO it during backward (_engine._UnetFunction.backward)."""
    return [e.flat_grad for k, e in model.__dict__.get("_engines", {}).items() if e.flat_grad is not None and k[2] == 0]


class NativeComm:
    """The C-ABI communicator (include/pidm.h: pidm_comm_*): RCCL bound inside the library, no torch collective on the data path.
    Rank 0's 128-byte id reaches the other ranks through the process group the caller already has (one broadcast at set-up);
    `allreduce_avg` then takes a raw pointer, a count and the current stream.  Construct it through `negotiate_native_comm`
    (every rank of the group, collectively): that is what `GradientExchange` does by default with more than one rank on GPUs."""

    def __init__(self, lib, device, group=None, ident: bytes | None = None):
        self.lib, self.device = lib, torch.device(device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if ident is None:
            ident = _broadcast_id(lib, self.device, group, _new_id(lib) if rank == 0 else None)
        self.handle = vp()
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                lib.check(lib.pidm_comm_init(rank, world, ident, C.byref(self.handle)), "pidm_comm_init")
        else:
            lib.check(lib.pidm_comm_init(rank, world, ident, C.byref(self.handle)), "pidm_comm_init")
        self.rank, self.world = rank, world

    def allreduce(self, buf, average: bool):
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        st = torch.cuda.current_stream(buf.device).cuda_stream if buf.is_cuda else 0
        self.lib.check(self.lib.pidm_allreduce_f32(self.handle, vp(buf.data_ptr()), buf.numel(), 1 if average else 0, vp(st)), "pidm_allreduce_f32")

    def allreduce_avg(self, buf):
        self.allreduce(buf, True)

    def self_check(self):
        """One float per rank through the communicator: the sum of rank + 1 must be N (N + 1) / 2 and the mean (N + 1) / 2 - a
        communicator that initialised but moves nothing (or adds where it should average) is caught before the first step."""
        n = self.world
        x = torch.full((2,), float(self.rank + 1), dtype=torch.float32, device=self.device)
        self.allreduce(x[:1], False)
        self.allreduce(x[1:], True)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        got = x.cpu().tolist()
        want = [n * (n + 1) / 2.0, (n + 1) / 2.0]
        if any(abs(g - w) > 1e-5 * w for g, w in zip(got, want)):     # (the mean of small integers: exact for 2^k ranks, a rounding away otherwise)
            raise RuntimeError(f"pidm_allreduce_f32 self-check: got sum {got[0]}, mean {got[1]}; expected {want[0]}, {want[1]} (rank {self.rank} of {n})")

    def close(self):
        if self.handle:
            self.lib.pidm_comm_destroy(self.handle)
            self.handle = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _new_id(lib) -> bytes:
    buf = (C.c_ubyte * 128)()
    lib.check(lib.pidm_comm_unique_id(buf), "pidm_comm_unique_id")
    return bytes(buf)


def _carrier_device(device, group):
    return torch.device(device) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _broadcast_id(lib, device, group, ident: bytes | None) -> bytes:
    t = torch.tensor(list(ident), dtype=torch.uint8) if ident is not None else torch.zeros(128, dtype=torch.uint8)
    t = t.to(_carrier_device(device, group))
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(t.cpu().tolist())


def _all_ranks_ok(ok: bool, device, group) -> bool:
    """Logical AND over the ranks of the group, through the torch process group (the channel that is known to work)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_carrier_device(device, group))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def negotiate_native_comm(lib, device, group=None):
    """COLLECTIVE over the group: try to bring up the C-ABI communicator on every rank, check it, and agree on the outcome.

    Returns (NativeComm, None) when every rank initialised it and passed the one-float self-check, else (None, reason) on EVERY
    rank - never a mixture (a mixture would deadlock the first step: some ranks in ncclAllReduce of one communicator, the rest in
    torch.distributed's).  The caller falls back to torch.distributed.all_reduce and reports `reason`.  What cannot be caught
    here: a rank that dies INSIDE ncclCommInitRank leaves the others waiting in it (RCCL's own time-out applies)."""
    rank = dist.get_rank(group)
    why = None
    ident = None
    # 1. is the binding there on every rank?  pidm_comm_available only binds the symbols; the id comes from rank 0 alone
    #    (ncclGetUniqueId starts a bootstrap root - a thread and a listening socket - that the other ranks would never use)
    try:
        lib.check(lib.pidm_comm_available(), "pidm_comm_available")
        if rank == 0:
            ident = _new_id(lib)
    except Exception as e:  # noqa: BLE001 - any failure means "fall back", the reason is reported
        why = f"rank {rank}: {e}"
    if not _all_ranks_ok(why is None, device, group):
        return None, why or "pidm_comm_available / pidm_comm_unique_id failed on another rank"
    ident = _broadcast_id(lib, device, group, ident)
    # 2. every rank enters pidm_comm_init (ncclCommInitRank is itself collective)
    comm = None
    try:
        comm = NativeComm(lib, device, group, ident=ident)
    except Exception as e:  # noqa: BLE001
        why = f"rank {rank}: {e}"
    if not _all_ranks_ok(comm is not None, device, group):
        if comm is not None:
            comm.close()
        return None, why or "pidm_comm_init failed on another rank"
    # 3. one float through it
    try:
        comm.self_check()
    except Exception as e:  # noqa: BLE001
        why = f"rank {rank}: {e}"
    if not _all_ranks_ok(why is None, device, group):
        comm.close()
        return None, why or "the pidm_allreduce_f32 self-check failed on another rank"
    return comm, None


def _allreduce_avg(buf, world, group, native=None):
    if native is not None:
        native.allreduce_avg(buf)                                   # RCCL through the library's own C entry point
    elif dist.get_backend(group) == "nccl":
        dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group)     # RCCL over xGMI
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf.mul_(1.0 / world)


def allreduce_gradients(model, world_size: int | None = None, group=None):
    """Average the gradients over all ranks (in place), one collective on the caller's stream.  Call between
    loss.backward() and clip_grad_norm_.  (`GradientExchange` is the overlapped form.)"""
    if not dist.is_initialized():
        return
    world = world_size or dist.get_world_size(group)
    if world == 1:
        return
    bufs = flat_gradient_buffers(model)
    if not bufs:
        raise RuntimeError("allreduce_gradients: no engine gradient buffer - run loss.backward() first")
    for buf in bufs:
        _allreduce_avg(buf, world, group)


class GradientExchange:
    """Overlapped, bucketed gradient average for a `Unet3D` driven by the gfx950 engine.

        ex = GradientExchange(model, world, diffusion=diffusion_utils)     # once, after model.to(device)
        loss.backward(); ex.allreduce(); clip / optimizer.step()

    buckets: 1..3 phases of the engine's deferred reduction (PIDM_DP_BUCKETS overrides; 1 = a single collective after
    backward, the round-1 behaviour).  diffusion: the DenoisingDiffusion whose mechanics inequality term should be made
    data-parallel exact (it needs the world size for one scalar all-reduce).

    Which collective carries the gradients (`self.collective`): on GPUs the library's own communicator (`pidm_allreduce_f32`,
    RCCL behind the C ABI - torch is a container here, not the data path) after `negotiate_native_comm` brought it up on every rank
    and its self-check passed; `torch.distributed.all_reduce` otherwise (CPU / gloo, `native=False`, `PIDM_DP_NATIVE=0`, or any
    failure of the negotiation - then `self.collective_note` says why).  `native=True` asks for the C-ABI communicator on any
    device; `native=<NativeComm>` uses a communicator the caller negotiated (the CPU tests bring one up over a library whose
    pidm_comm_* entries they control)."""

    def __init__(self, model, world_size: int | None = None, image_size: int = 64, buckets: int = 3, group=None, lib=None,
                 diffusion=None, force: bool = False, native: bool | None = None):
        self.model, self.group = model, group
        self.native = None
        env_native = os.environ.get("PIDM_DP_NATIVE")
        # None = default: the C-ABI communicator on GPUs, torch.distributed elsewhere; a NativeComm object = use this one
        given = native if isinstance(native, NativeComm) else None
        self._want_native = (None if env_native is None else env_native == "1") if native is None else bool(native)
        self.world = world_size or (dist.get_world_size(group) if dist.is_initialized() else 1)
        # force: run the collectives even with one rank (exercises RCCL, the side stream and the phase events on a single-GPU box)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.eng = get_engine(model, image_size, lib)
        self.buckets = max(1, min(3, int(os.environ.get("PIDM_DP_BUCKETS", buckets))))
        if diffusion is not None:
            diffusion.data_parallel_world = self.world
            diffusion.data_parallel_group = group
        eng = self.eng
        # element ranges of the flat buffer per phase; the last phase takes everything the earlier ones do not cover
        offs = [0]
        for ne in eng.numels:
            offs.append(offs[-1] + ne)
        self.ranges = []
        covered = []
        a, b = C.c_int(), C.c_int()
        for k in range(self.buckets - 1):
            eng.lib.check(eng.lib.pidm_unet_grad_phase_range(eng.handle, self.buckets, k, C.byref(a), C.byref(b)), "pidm_unet_grad_phase_range")
            self.ranges.append([(offs[a.value], offs[b.value])])
            covered.append((offs[a.value], offs[b.value]))
        rest, pos = [], 0
        for lo, hi in sorted(covered):
            if lo > pos:
                rest.append((pos, lo))
            pos = max(pos, hi)
        if pos < offs[-1]:
            rest.append((pos, offs[-1]))
        self.ranges.append(rest)
        dev = eng.params[0].device
        self.on_gpu = dev.type == "cuda"
        self.collective_note = None
        want = self._want_native is True or (self._want_native is None and self.on_gpu)
        if given is None and self.active and dist.is_initialized():
            # the wish is per rank (PIDM_DP_NATIVE is read from each rank's own environment): every rank enters this vote, and the
            # collective negotiation below only runs when ALL of them want it - a rank that skipped it would leave the others waiting
            agreed = _all_ranks_ok(want, dev, group)
            if want and not agreed:
                self.collective_note = "PIDM_DP_NATIVE / native= differ between ranks: another rank asked for torch.distributed"
                print(f"GradientExchange: C-ABI communicator not used ({self.collective_note})", flush=True)
            want = agreed
        if given is not None:
            self.native = given if self.active else None
        elif self.active and want:
            self.native, self.collective_note = negotiate_native_comm(eng.lib, dev, group)
            if self.native is None:
                print(f"GradientExchange: C-ABI communicator not used ({self.collective_note}); falling back to torch.distributed.all_reduce",
                      flush=True)
        if self.native is not None:
            self.collective = "pidm_allreduce_f32 (C-ABI communicator over RCCL)"
        elif dist.is_initialized():
            self.collective = f"torch.distributed.all_reduce ({dist.get_backend(group)})"
        else:
            self.collective = "none (one rank)"
        self.events, self.stream = None, None
        self._closed = True                # until the engine has been told about this object
        handles = None
        if self.on_gpu and self.active and os.environ.get("PIDM_DP_NO_OVERLAP") != "1":
            try:
                self.stream = torch.cuda.Stream(device=dev)
                self.events = [torch.cuda.Event(enable_timing=False) for _ in range(self.buckets)]
                for ev in self.events:
                    ev.record(torch.cuda.current_stream(dev))      # creates the underlying hipEvent_t
                handles = (vp * 3)(*[vp(ev.cuda_event) for ev in self.events], *([vp(0)] * (3 - self.buckets)))
            except (AttributeError, RuntimeError) as e:            # no raw event handle on this torch build: exchange after backward
                print(f"GradientExchange: overlapped exchange disabled ({e})")
                self.events = self.stream = handles = None
        # an inactive exchange (one rank, not forced) keeps the engine's single deferred reduction: phases exist for the overlap only
        self.n_phases = self.buckets if self.active else 1
        eng.lib.check(eng.lib.pidm_unet_set_grad_events(eng.handle, self.n_phases, handles), "pidm_unet_set_grad_events")
        eng.backward_calls = 0
        eng._exchange_owner = id(self)     # a later exchange of the same engine takes over; close() of an older one is then a no-op
        self.last_overlapped = None        # how the latest allreduce() ran (tests / bench read it)
        self.measure = False               # True: time every allreduce() with events on the stream the collectives run on
        self._timing = []                  # (start event, end event) per measured call
        self._closed = False

    def close(self):
        """Detach from the engine: it holds the RAW hipEvent_t handles of this object's torch events and would record on
        destroyed events once they are garbage-collected.  Called by __del__; idempotent."""
        if self._closed:
            return
        self._closed = True
        try:
            eng = self.eng
            if self.stream is not None:
                self.stream.synchronize()
            if getattr(eng, "_exchange_owner", None) == id(self):
                eng.lib.check(eng.lib.pidm_unet_set_grad_events(eng.handle, 1, None), "pidm_unet_set_grad_events")
                eng._exchange_owner = None
        except Exception:
            pass
        self.events = self.stream = None
        self.active = False
        if self.native is not None:
            self.native.close()
            self.native = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce(self):
        """Average the gradients over all ranks (in place).  Call right after loss.backward()."""
        eng = self.eng
        calls, eng.backward_calls = eng.backward_calls, 0
        if self._closed:
            raise RuntimeError("GradientExchange.allreduce: the exchange was closed")
        if not self.active:
            return
        flat = eng.flat_grad
        if flat is None:
            raise RuntimeError("GradientExchange.allreduce: no engine gradient buffer - run loss.backward() first")
        overlapped = self.events is not None and calls == 1
        self.last_overlapped = overlapped
        timing = None
        if self.measure and self.on_gpu:
            timing = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._timing.append(timing)
        if not overlapped:
            # several backward passes wrote / accumulated into the buffer after the phase events: exchange once everything is in
            if timing:
                timing[0].record()
            for rs in self.ranges:
                for lo, hi in rs:
                    _allreduce_avg(flat[lo:hi], self.world, self.group, self.native)
            if timing:
                timing[1].record()
            return
        cur = torch.cuda.current_stream(flat.device)
        for k, rs in enumerate(self.ranges):
            self.stream.wait_event(self.events[k])
            with torch.cuda.stream(self.stream):
                if timing and k == 0:
                    timing[0].record()          # first phase's gradients are final: the exchange starts here
                for lo, hi in rs:
                    _allreduce_avg(flat[lo:hi], self.world, self.group, self.native)
                if timing and k == len(self.ranges) - 1:
                    timing[1].record()
        cur.wait_stream(self.stream)

    def exchange_ms(self):
        """Mean wall time (ms) from the start of the first collective to the end of the last one over the allreduce() calls made
        while `measure` was on (synchronises; with the overlap on, this span runs concurrently with the rest of backward)."""
        if not self._timing:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._timing]
        self._timing = []
        return sum(ms) / len(ms)


def shard_batch(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Equal contiguous shards of the global batch (C3: 512 -> 8 x 64)."""
    if batch.shape[0] % world:
        raise ValueError("global batch must be divisible by the number of ranks")
    n = batch.shape[0] // world
    return batch[rank * n:(rank + 1) * n]
