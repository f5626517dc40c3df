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
    `allreduce_avg` then takes a raw pointer, a count and the current stream.  Opt-in (`PIDM_DP_NATIVE=1`, or pass one to
    GradientExchange): torch.distributed's own RCCL path stays the default."""

    def __init__(self, lib, device, group=None):
        self.lib, self.device = lib, torch.device(device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            lib.check(lib.pidm_comm_unique_id(buf), "pidm_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        carrier = ident.to(self.device) if dist.get_backend(group) == "nccl" else ident
        dist.broadcast(carrier, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(carrier.cpu().tolist())
        self.handle = vp()
        with torch.cuda.device(self.device):
            lib.check(lib.pidm_comm_init(rank, world, raw, C.byref(self.handle)), "pidm_comm_init")
        self.world = world

    def allreduce_avg(self, buf):
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
        st = torch.cuda.current_stream(buf.device).cuda_stream
        self.lib.check(self.lib.pidm_allreduce_f32(self.handle, vp(buf.data_ptr()), buf.numel(), 1, vp(st)), "pidm_allreduce_f32")

    def close(self):
        if self.handle:
            self.lib.pidm_comm_destroy(self.handle)
            self.handle = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
    data-parallel exact (it needs the world size for one scalar all-reduce)."""

    def __init__(self, model, world_size: int | None = None, image_size: int = 64, buckets: int = 3, group=None, lib=None,
                 diffusion=None, force: bool = False, native: bool | None = None):
        self.model, self.group = model, group
        self.native = None
        self._want_native = (os.environ.get("PIDM_DP_NATIVE") == "1") if native is None else bool(native)
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
        if self._want_native and self.active and self.on_gpu:
            self.native = NativeComm(eng.lib, dev, group)
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
