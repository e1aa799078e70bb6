"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)`` (utils/model_utils.py:39-58,
pipelines/launch.py:20-50).  Here every parameter gradient already lives in ONE flat fp32 buffer in ``named_parameters()``
order -- transformer, embeddings, projections, class-branch encoder, heads, THEN the CSN body (stem, layer1..4) and the pool
decoder -- so the reducer is a handful of large all-reduces on contiguous windows of that buffer, issued as soon as a window is
final and overlapped with the remaining backward kernels:

    backward reaches ...            window that is final             -> all-reduce on the reducer's side stream
    body backward entry             everything laid out behind the body (pool decoder)
    end of layer4 / 3 / 2 / 1       that stage's parameters (offsets >= the stage's first block)
    end of backward                 the rest: stem, and the transformer / head window laid out before the body

Transport (``RcclComm``): RCCL bound directly through the C ABI (``csrc/collective.cpp``: ``ncclAllReduce`` from the librccl the
process already has), on a communicator and a HIP stream this module owns.  Ordering is by events only: the side stream waits for
the backward stream at each issue point, the optimizer waits for the side stream once; the host never blocks.  The averaging
(x 1/world) runs on the side stream right behind each window's all-reduce, so it is overlapped too.  Windows of frozen parameters
(``requires_grad = False``) hold no gradient and are never sent.  Optional bf16 compression (``TUBER_DDP_BF16=1`` or
``compress=True``): a window is cast to bf16 into a staging buffer, summed in bf16, and expanded + averaged back into the fp32
buffer -- half the xGMI bytes (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: the ring is per-link bound), at bf16 sum
precision.  Without a GPU / with the gloo backend (CPU tests, two ranks sharing one GPU) the same windows go through
``torch.distributed.all_reduce``.

The hipGraph step (training.GraphedTrainStep) does not use the backward hooks: it cuts its graph once, where layer3's backward
ends, and calls ``reduce()`` for the final windows before replaying the rest (layer2 / layer1 / stem backward) -- or, with
``TUBER_RCCL_IN_GRAPH=1``, keeps the hooks and captures the collectives INTO the graph as a forked branch.

No bucket copies, no unused-parameter bitmap (C2), no per-step buffer broadcast (C3): BatchNorm statistics stay local like the
reference's non-synchronised BatchNorm3d; rank 0's running statistics are what a checkpoint holds.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def trainable_ranges(store):
    """merged [begin, end) windows of the flat buffers that belong to parameters with ``requires_grad``."""
    out = []
    for n, p in zip(store.names, store.params):
        if not p.requires_grad:
            continue
        o = store.offsets[n]
        e = o + (p.numel() + 63) // 64 * 64
        if out and out[-1][1] == o:
            out[-1][1] = e
        else:
            out.append([o, e])
    return [(a, b) for a, b in out]


class RcclComm:
    """One RCCL communicator + one HIP stream owned by this process (= this GPU).  The 128-byte unique id travels from rank 0
    through whatever torch.distributed process group exists (any backend); a single-rank communicator needs none."""

    def __init__(self, device, rank=None, world=None):
        from . import lib
        self.lib = lib.load()
        self.device = torch.device(device)
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = rank, world
        v = self.lib.tuber_comm_version()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if v < 0:
            raise RuntimeError("RCCL unavailable: %s" % self.lib.tuber_comm_last_error().decode())
        self.version = v
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self.lib.tuber_comm_unique_id(ctypes.addressof(uid)), "tuber_comm_unique_id")
        if world > 1:
            box = [uid.raw]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        # deadline on the bootstrap (TUBER_RCCL_INIT_TIMEOUT_S, default 120 s; 0 = none): a missing rank is an error message, not a hang
        timeout_ms = int(float(os.environ.get("TUBER_RCCL_INIT_TIMEOUT_S", "120")) * 1000) if world > 1 else 0
        rc = self.lib.tuber_comm_init_timeout(ctypes.addressof(uid), world, rank, idx, timeout_ms, ctypes.addressof(comm))
        if rc == self.lib.tuber_comm_etimedout():        # TUBER_ETIMEDOUT (exported by the library): a rank never reached the bootstrap.  Fatal for the job: a helper thread is still
            raise RcclBootstrapTimeout(self.lib.tuber_comm_last_error().decode())      # parked inside RCCL on this device
        self._check(rc, "tuber_comm_init")
        self.comm = comm.value
        n, r = ctypes.c_int(-1), ctypes.c_int(-1)
        self._check(self.lib.tuber_comm_count(self.comm, ctypes.addressof(n), ctypes.addressof(r)), "tuber_comm_count")
        self.ranks_seen, self.rank_seen = n.value, r.value          # what RCCL itself reports (bench.py's `comm` object)
        if (self.ranks_seen, self.rank_seen) != (world, rank):
            raise RuntimeError("RCCL communicator has %d ranks (this one is %d); expected %d / %d" % (n.value, r.value, world, rank))
        self.stream = torch.cuda.Stream(device=self.device)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.tuber_comm_last_error().decode()))

    def all_reduce(self, ptr, count, bf16=False, stream=None):
        """in-place sum of ``count`` elements at device address ``ptr`` on ``stream`` (default: the owned side stream)."""
        s = (stream or self.stream).cuda_stream
        self._check(self.lib.tuber_comm_allreduce_sum(self.comm, ptr, count, 1 if bf16 else 0, s), "tuber_comm_allreduce_sum")

    def close(self):
        if getattr(self, "comm", None):
            torch.cuda.synchronize(self.device)
            self.lib.tuber_comm_destroy(self.comm)
            self.comm = None


class StreamEdge:
    """One-directional ordering edge between two HIP streams of this device: what was enqueued on ``src`` so far happens before what is
    enqueued on ``dst`` afterwards (hipEventRecord + hipStreamWaitEvent on a reusable event).

    torch's ``dst.wait_stream(src)`` records a DEFAULT event, and a default HIP event performs a SYSTEM-scope release when it completes
    (cache write-back + invalidate: /opt/rocm/include/hip/hip_runtime_api.h, hipEventDisableSystemFence).  Round 6 measured what that costs
    the captured step (scripts/r06_ddp_probe.sh): the pair of edges around the gradient exchange -- with NO collective enqueued between
    them -- slows the step by 0.7 - 0.8 ms, the same as with the one-rank ncclAllReduce (which is a no-op on the device: 5 us of stream
    time).  ``mode``: 'torch' = wait_stream; 'device' = own event with hipEventReleaseToDevice (an agent-scope release is all another
    stream of the SAME device needs); 'nofence' = hipEventDisableSystemFence."""
    _hip = None
    FLAGS = {"device": 0x2 | 0x40000000, "nofence": 0x2 | 0x20000000, "plain": 0x2}

    def __init__(self, mode):
        self.mode = mode
        self.ev = None
        if mode != "torch":
            if StreamEdge._hip is None:
                StreamEdge._hip = ctypes.CDLL("libamdhip64.so")
            ev = ctypes.c_void_p()
            rc = StreamEdge._hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(self.FLAGS[mode]))
            if rc != 0:
                raise RuntimeError("hipEventCreateWithFlags(%s) failed: %d" % (mode, rc))
            self.ev = ev

    def __call__(self, src, dst):
        if self.ev is None:
            dst.wait_stream(src)
            return
        h = StreamEdge._hip
        rc = h.hipEventRecord(self.ev, ctypes.c_void_p(src.cuda_stream))
        rc = rc or h.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), self.ev, ctypes.c_uint(0))
        if rc != 0:
            raise RuntimeError("stream edge failed: %d" % rc)


class RcclBootstrapTimeout(RuntimeError):
    """ncclCommInitRank did not complete before TUBER_RCCL_INIT_TIMEOUT_S: a rank is missing.  Not recoverable in-process."""


class FlatGradReducer:
    def __init__(self, store, world_size=None, min_bucket=8 << 20, comm=None, compress=None):
        self.store = store
        self.comm = comm
        self.world = world_size if world_size is not None else (comm.world if comm is not None else (dist.get_world_size() if dist.is_initialized() else 1))
        self.min_bucket = min_bucket          # elements
        self.compress = bool(os.environ.get("TUBER_DDP_BF16")) if compress is None else bool(compress)
        self.stage = None                     # bf16 staging buffer of the compressed path (allocated on first use)
        self.dry = False                      # hooks fire but nothing is sent (capture warm-up)
        self._joined = True
        self.handles = []
        self.issued = 0                       # elements handed to the transport since begin() (tests / logging)
        self.windows = 0                      # all-reduce calls since begin()
        self.measure = False                  # bench.py: time how long the optimizer's stream stalls on the transport (HIP events)
        self.exposed = []                     # [(event before the wait, event after it)] of the measured steps
        self.win_events = []                  # measured steps: per step [(t0 on the backward stream at begin(), [(issue, start, end, bytes)])]
        self.done_from = store.total          # everything at offsets >= done_from has been handed to the transport
        self.late = []                        # [(begin, end)] windows that must wait for the end of backward
        for n in store.names:
            if n.endswith("query_embed.weight") or n.endswith("query_pool.weight"):
                o = store.offsets[n]
                self.late.append((o, o + (store.module.get_parameter(n).numel() + 63) // 64 * 64))
        self.late.sort()
        self.ranges = trainable_ranges(store)
        # how the transport's stream learns that a window is final on the backward stream (see flag_points): 'flag' = a counter in device
        # memory, bumped by a one-thread kernel that is the last node of a graph part and polled by a one-wave kernel in front of the
        # collective (csrc/stream_flag.hip); 'event' = hipEventRecord + hipStreamWaitEvent as until round 5 ('device' / 'nofence' /
        # 'plain': the same with other event flags -- measured identical); 'hybrid' (default) = counter at the first cut, events after.
        self.edge_mode = os.environ.get("TUBER_DDP_EDGE", "hybrid")
        ev_mode = "torch" if self.edge_mode in ("flag", "event", "hybrid") else self.edge_mode
        self.edge_out = [StreamEdge(ev_mode) for _ in range(8)]  # backward stream -> transport stream, one reusable event per issue point of a step
        self.edge_back = StreamEdge(ev_mode)                      # transport stream -> optimizer's stream (few kernels behind it: cheap)
        self.edge_k = 0
        self.flags = torch.zeros(16, dtype=torch.int32, device=store.device) if comm is not None else None      # [0..7] counters, [15] error word
        self.expect = [0] * 8

    # -- step protocol -----------------------------------------------------------------------------------------------------
    def begin(self):
        self.handles = []
        self.issued = 0
        self.windows = 0
        self.done_from = self.store.total
        self.edge_k = 0
        self.ranges = trainable_ranges(self.store)       # windows of frozen parameters hold no gradient: never sent
        self._joined = True
        if self.measure and self.comm is not None and not torch.cuda.is_current_stream_capturing():
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            self.win_events.append((e, []))

    def flag_points(self):
        """issue points of the captured step (0 = the first cut) that are ordered by a device-memory counter instead of an event.
        Measured on the one-rank line, per graph part, against the same cut graphs with no edges (profiles/r06_ddp_edges.txt):
          event edge   : the part in front of it + 0.35 ms when that part is the 520 small launches of forward + transformer backward
                         (a pending hipStreamWaitEvent costs every dispatch of the other queue), + 0.09 / + 0.06 ms for the layer3 /
                         layer2-1 backward parts (fewer, longer kernels);
          counter edge : + 0.02 ms in front of the first cut, but + 0.41 / + 0.27 ms on the layer3 / layer2-1 backward parts: the polling
                         wave holds a few VGPRs of one SIMD, and the one-workgroup-per-CU backward kernels (dwconv_tile_bwd_both, conv1_bwd,
                         conv4_bwd: 8 waves x up to 256 VGPRs = a CU's whole register file) then find 255 free CUs and run a second round.
        'hybrid' (default) takes the cheap one at each point: counter for the first cut, events behind it."""
        if self.comm is None or self.edge_mode not in ("flag", "hybrid"):
            return frozenset()
        return frozenset(range(8)) if self.edge_mode == "flag" else frozenset([0])

    def signal(self, i):
        """on the CURRENT stream (the backward stream; capturable): counter i += 1 once everything enqueued before has completed"""
        from . import lib
        lib.call("tuber_flag_signal", self.flags.data_ptr() + 4 * i)

    def wait_for(self, i):
        """the transport's stream goes on once signal(i) of this step has executed (no runtime-level dependency between the queues)"""
        from . import lib
        self.expect[i] = (self.expect[i] + 1) & 0xFFFFFFFF
        lib.call("tuber_flag_wait", self.flags.data_ptr() + 4 * i, self.expect[i], self.flags.data_ptr() + 4 * 15, self.comm.stream.cuda_stream)
        self._joined = False

    def check(self):
        """host-side (syncs): a flag wait that gave up after 2 s means a signal was lost -- the step ordering is broken"""
        if self.flags is not None and int(self.flags[15].item()):
            raise RuntimeError("gradient exchange: a stream-flag wait timed out (counters %s, expected %s)" % (self.flags[:8].tolist(), self.expect))

    def reduce(self, lo, hi, edge=True):
        """all-reduce (and average) the trainable part of gflat[lo:hi); returns immediately.  ``edge``: order the transport's stream behind the
        current stream with an event first (False: the caller has already ordered it -- wait_for(), or an earlier window of the same issue point)."""
        if hi <= lo or (self.world <= 1 and self.comm is None):
            return
        wins = [(max(a, lo), min(b, hi)) for a, b in self.ranges]
        wins = [(a, b) for a, b in wins if b > a]
        if not wins or self.dry:
            return
        st = self.store
        if self.comm is None:
            for a, b in wins:
                self.handles.append(dist.all_reduce(st.gflat[a:b], op=dist.ReduceOp.SUM, async_op=True))
                self.issued += b - a
                self.windows += 1
            return
        from . import lib
        cs = self.comm.stream
        timed = self.measure and bool(self.win_events) and not torch.cuda.is_current_stream_capturing()
        if timed:
            issue = torch.cuda.Event(enable_timing=True)
            issue.record(torch.cuda.current_stream())           # when the backward stream reaches this issue point
        if edge:
            self.edge_out[self.edge_k % len(self.edge_out)](torch.cuda.current_stream(), cs)      # the windows are final on the backward stream up to here
            self.edge_k += 1
        self._joined = False
        inv = 1.0 / self.world
        base = st.gflat.data_ptr()
        skip_call = bool(os.environ.get("TUBER_DDP_SKIP_CALL"))      # diagnostic: everything but the ncclAllReduce call itself (stream edges only)
        import time as _time
        h0 = _time.perf_counter()
        with torch.cuda.stream(cs):
            if timed:
                w0 = torch.cuda.Event(enable_timing=True)
                w0.record(cs)
            for a, b in wins:
                n = b - a
                if self.compress:
                    if self.stage is None:
                        self.stage = torch.empty(st.total, dtype=torch.bfloat16, device=st.device)
                    sp = self.stage.data_ptr() + 2 * a
                    lib.call("tuber_cast_f32_bf16", base + 4 * a, sp, n)
                    self.comm.all_reduce(sp, n, bf16=True)
                    lib.call("tuber_cast_bf16_f32_scale", sp, base + 4 * a, n, inv)
                else:
                    if not skip_call:
                        self.comm.all_reduce(base + 4 * a, n)
                    if self.world > 1:
                        lib.call("tuber_scale_f32", base + 4 * a, n, None, inv)
                self.issued += n
                self.windows += 1
            if timed:
                w1 = torch.cuda.Event(enable_timing=True)
                w1.record(cs)
                self.win_events[-1][1].append((issue, w0, w1, sum(b - a for a, b in wins) * (2 if self.compress else 4), _time.perf_counter() - h0))

    def _reduce_excluding_late(self, lo, hi):
        cur = lo
        for a, b in self.late:
            if b <= lo or a >= hi:
                continue
            self.reduce(cur, max(cur, a))
            cur = max(cur, b)
        self.reduce(cur, hi)

    def notify(self, offset, force=False):
        """Backward has finished every parameter at flat offsets >= ``offset``."""
        if offset >= self.done_from:
            return
        if not force and self.done_from - offset < self.min_bucket:
            return
        self._reduce_excluding_late(offset, self.done_from)
        self.done_from = offset

    def finish(self, rest=True):
        """End of backward: reduce what is left (stem + late windows) when ``rest``, then make the optimizer's stream wait for
        the transport (events only) -- gradients are averaged when this returns (in stream order)."""
        if rest:
            self.notify(0, force=True)
            for a, b in self.late:
                self.reduce(a, b)
        if self.comm is not None:
            if not self._joined:
                cur = torch.cuda.current_stream()
                timed = self.measure and not torch.cuda.is_current_stream_capturing()
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                self.edge_back(self.comm.stream, cur)
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(cur)
                    self.exposed.append((e0, e1))
                self._joined = True
            return
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.world > 1 and not self.dry:
            from . import lib
            st = self.store
            if st.gflat.is_cuda:
                for a, b in self.ranges:
                    lib.call("tuber_scale_f32", st.gflat.data_ptr() + 4 * a, b - a, None, 1.0 / self.world)
            else:
                for a, b in self.ranges:
                    st.gflat[a:b].mul_(1.0 / self.world)


    def describe(self):
        """what bench.py prints as its ``comm`` object: who carries the gradients, how much, in how many pieces, and how long the
        optimizer's stream actually waited for it (mean over the measured steps; None when nothing was measured)."""
        exposed = None
        if self.exposed:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self.exposed]
            exposed = sum(ms) / len(ms)
        bpe = 2 if self.compress else 4
        per_issue = None
        if self.win_events:
            # per issue point (mean over the measured steps): when the backward stream reached it (ms after begin()), how long the
            # transport's stream was busy with it, and how late after the issue it finished
            torch.cuda.synchronize()
            k = min(len(w) for _, w in self.win_events)
            per_issue = []
            for i in range(k):
                rows = [(t0.elapsed_time(w[i][0]), w[i][1].elapsed_time(w[i][2]), w[i][0].elapsed_time(w[i][2]), w[i][3], 1e3 * w[i][4]) for t0, w in self.win_events]
                m = [sum(r[j] for r in rows) / len(rows) for j in (0, 1, 2, 4)]
                per_issue.append({"MB": round(rows[0][3] / 1e6, 1), "issued_at_ms": round(m[0], 3), "stream_busy_ms": round(m[1], 3), "done_after_issue_ms": round(m[2], 3),
                                  "host_call_ms": round(m[3], 3)})
            self.win_events = []
        return {"transport": "own RCCL communicator (csrc/collective.cpp)" if self.comm is not None else "torch.distributed process group (%s)" % (dist.get_backend() if dist.is_initialized() else "none"),
                "world": self.world, "ranks_seen_by_rccl": getattr(self.comm, "ranks_seen", None) if self.comm is not None else None,
                "rccl_version": getattr(self.comm, "version", None) if self.comm is not None else None,
                "bf16_compressed": bool(self.compress), "windows_per_step": self.windows, "bytes_per_step": self.issued * bpe,
                "exposed_ms": exposed, "issue_points": per_issue}


def broadcast_parameters(store, src=0):
    """Initial parameter + buffer broadcast (DDP constructor, collective C4)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.broadcast(store.flat, src)
    for b in store.module.buffers():
        dist.broadcast(b, src)


def attach_reducer(store, force=False):
    """reducer for ``store`` when a process group with > 1 rank exists (or ``force``: a one-rank communicator, used to exercise the
    whole N > 1 code path -- RCCL init, stream ordering, graph cut -- on a single GPU).  RCCL transport when the parameters are on
    a GPU and the process group is nccl (or there is none / TUBER_OWN_RCCL=1); torch.distributed otherwise."""
    ddp = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if ddp else 1
    if world <= 1 and not force:
        return None
    comm = None
    own = store.device.type == "cuda" and (not ddp or dist.get_backend() == "nccl" or os.environ.get("TUBER_OWN_RCCL"))
    if own and not os.environ.get("TUBER_NO_OWN_RCCL"):
        def agree(ok):
            """MIN over the ranks of a local yes / no (every rank calls this the same number of times, whatever failed locally)"""
            if world <= 1:
                return ok
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=store.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag) == 1

        # ncclCommInitRank is itself a collective: agree that EVERY rank can load librccl and owns a usable device BEFORE anyone enters
        # it -- a rank that failed locally would otherwise leave the others blocked inside the bootstrap (which also has a deadline)
        err = None
        try:
            from . import lib
            if lib.load().tuber_comm_version() < 0:
                raise RuntimeError(lib.load().tuber_comm_last_error().decode())
            torch.cuda.set_device(store.device)
        except Exception as e:                      # noqa: BLE001 -- reported below, on every rank
            err = e
        if world <= 1 and err is not None:
            raise err
        ready = agree(err is None)
        if ready:
            try:
                comm = RcclComm(store.device)
            except RcclBootstrapTimeout:            # every rank still in the bootstrap hits the same deadline: the job ends, readable
                raise
            except Exception as e:                  # noqa: BLE001
                if world <= 1:
                    raise
                err = e
            # all ranks must end up on the SAME transport: if the directly bound communicator failed anywhere (e.g. the bootstrap
            # deadline), everybody takes the process group's (both are RCCL over xGMI; the own one adds its own stream + the graph cut)
            ready = agree(err is None)
        if not ready:
            import sys
            print("[tuber ddp] rank %d: own RCCL communicator unavailable (%s); using the torch.distributed process group"
                  % (dist.get_rank() if ddp else 0, err if err is not None else "failed on another rank"), file=sys.stderr, flush=True)
            if comm is not None:
                comm.close()
            comm = None
    store.reducer = FlatGradReducer(store, world_size=world, comm=comm)
    return store.reducer
