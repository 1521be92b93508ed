"""Stage-to-stage data plane.

The reference moves hidden states as JSON text over HTTP through a public tunnel, relayed by
the orchestrator, the whole sequence on every hop (orchestration.py:114-137; Worker1.py:216-239).
Here an `Edge` is one directed hop of the pipeline (stage i -> i+1, or the token feedback from
the last stage to stage 0) and carries only the new tokens' rows:

* `TorchDistEdge` — `torch.distributed` point-to-point. Backend "nccl" is RCCL on ROCm: the
  buffer goes GPU-to-GPU over the xGMI link joining the two devices, enqueued on RCCL's own
  stream and chained to the compute stream by events, so the host never blocks and hop *t*
  overlaps the next micro-batch's compute. Backend "gloo" is the CPU path used by the tests.
  Every edge gets its OWN 2-rank process group (own communicator, own stream): forward hops and
  the feedback hop can never queue behind each other, which is what would deadlock a full
  pipeline if they shared one in-order stream.
* `LoopbackEdge` — in-process queues, for single-process pipelines in tests.
* HTTP (`serving/worker.py`) keeps the reference's `/process` wire format for plumbing parity.

An edge also carries the control messages (step descriptors) of its hop: over a shared-memory
ring when both ends are on one node (`ShmControl`), or over gloo otherwise.
"""
from __future__ import annotations

import queue
import threading
from typing import List, Optional

import torch


class Handle:
    """Completion of an async send/recv."""

    def wait(self):          # make the current stream / the host observe completion
        pass

    def done(self) -> bool:
        return True


class _WorkHandle(Handle):
    def __init__(self, work, keep=None):
        self.work, self.keep = work, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def done(self) -> bool:
        return self.work is None or self.work.is_completed()


class _PolledWorkHandle(Handle):
    """gloo point-to-point work never reports is_completed() until someone calls wait(), so
    polling needs a helper thread that waits and raises a flag."""

    def __init__(self, work, keep=None):
        self.work, self.keep = work, keep
        self._flag = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self._err: Optional[BaseException] = None

    def _run(self):
        try:
            self.work.wait()
        except BaseException as e:      # surfaced by wait()
            self._err = e
        self._flag.set()

    def _ensure(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def wait(self):
        self._ensure()
        self._flag.wait()
        if self._err is not None:
            raise self._err

    def done(self) -> bool:
        self._ensure()
        return self._flag.is_set()


class _StagedRecvHandle(_PolledWorkHandle):
    """Receive into pinned host memory over gloo, then copy to the GPU buffer on the stream that is
    current when the consumer waits (the copy is ordered before whatever that stream runs next)."""

    def __init__(self, work, host: torch.Tensor, dev_buf: torch.Tensor):
        super().__init__(work, keep=host)
        self.dev_buf = dev_buf
        self._copied = False

    def wait(self):
        super().wait()
        if not self._copied:
            self.dev_buf.copy_(self.keep, non_blocking=True)
            self._copied = True


class Edge:
    def send(self, t: torch.Tensor) -> Handle:
        raise NotImplementedError

    def recv(self, buf: torch.Tensor) -> Handle:
        raise NotImplementedError

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------
class LoopbackEdge(Edge):
    def __init__(self):
        self.q: "queue.Queue" = queue.Queue()      # (tensor, ready event or None)
        self._order_lock = threading.Lock()
        self._next_ticket = 0
        self._served = 0

    def send(self, t):
        c = t.detach().clone()
        ev = None
        if c.is_cuda:                      # the receiver may run on another stream: hand it an event
            ev = torch.cuda.Event()
            ev.record()
        self.q.put((c, ev))
        return Handle()

    def recv(self, buf, timeout: float = 60.0):
        return _LoopbackRecv(self, buf, timeout)


class _LoopbackRecv(Handle):
    """Receives are matched in posting order, like a real in-order transport: a handle may only
    take from the queue once every earlier receive on the edge has taken its message."""

    def __init__(self, edge: "LoopbackEdge", buf: torch.Tensor, timeout: float):
        self.edge, self.buf, self.timeout = edge, buf, timeout
        self._done = False
        with edge._order_lock:
            self.ticket = edge._next_ticket
            edge._next_ticket += 1

    def _take(self, block: bool) -> bool:
        e = self.edge
        with e._order_lock:
            if self._done:
                return True
            if e._served != self.ticket:
                return False
            try:
                t, ev = e.q.get(block=False)
            except queue.Empty:
                return False
            if ev is not None:
                torch.cuda.current_stream(t.device).wait_event(ev)
            if t.numel() != self.buf.numel():
                raise RuntimeError(f"loopback edge: expected {self.buf.numel()} elements, got {t.numel()}")
            self.buf.copy_(t.reshape(self.buf.shape).to(self.buf.dtype))
            e._served += 1
            self._done = True
            return True

    def done(self) -> bool:
        return self._take(block=False)

    def wait(self):
        import time as _t
        t0 = _t.monotonic()
        while not self._take(block=False):
            if _t.monotonic() - t0 > self.timeout:
                raise TimeoutError("loopback edge: no message")
            _t.sleep(0.0002)


class TorchDistEdge(Edge):
    """One directed hop src_rank -> dst_rank over its own 2-rank process group."""

    def __init__(self, group, src_rank: int, dst_rank: int, my_rank: int, host_staged: bool = False):
        self.group, self.src, self.dst, self.me = group, src_rank, dst_rank, my_rank
        self.host_staged = host_staged      # gloo group carrying GPU tensors: bounce through pinned memory

    def send(self, t):
        import torch.distributed as dist
        assert self.me == self.src
        t = t.contiguous()
        if t.is_cuda and self.host_staged:
            host = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
            host.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()      # the bytes must be on the host before gloo reads them
            return _PolledWorkHandle(dist.isend(host, self.dst, group=self.group), keep=host)
        cls = _WorkHandle if t.is_cuda else _PolledWorkHandle
        return cls(dist.isend(t, self.dst, group=self.group), keep=t)

    def recv(self, buf):
        import torch.distributed as dist
        assert self.me == self.dst
        assert buf.is_contiguous()
        if buf.is_cuda and self.host_staged:
            host = torch.empty(buf.shape, dtype=buf.dtype, device="cpu", pin_memory=True)
            return _StagedRecvHandle(dist.irecv(host, self.src, group=self.group), host, buf)
        cls = _WorkHandle if buf.is_cuda else _PolledWorkHandle
        return cls(dist.irecv(buf, self.src, group=self.group))


# ---------------------------------------------------------------------------------------------
class Control:
    """FIFO of small byte messages along one hop."""

    def put(self, msg: bytes):
        raise NotImplementedError

    def get(self, timeout_s: Optional[float] = None) -> Optional[bytes]:
        raise NotImplementedError

    def close(self):
        pass


class QueueControl(Control):
    def __init__(self):
        self.q: "queue.Queue[bytes]" = queue.Queue()

    def put(self, msg):
        self.q.put(msg)

    def get(self, timeout_s=None):
        try:
            return self.q.get(timeout=timeout_s)
        except queue.Empty:
            return None


class ShmControl(Control):
    def __init__(self, name: str, create: bool, slots: int = 512, slot_bytes: int = 1 << 17):
        from .shm import ShmRing
        self.ring = ShmRing(name, create, slots, slot_bytes)

    def put(self, msg):
        self.ring.put(msg, timeout_s=120.0)

    def pending(self) -> int:
        return self.ring.pending()

    def get(self, timeout_s=None):
        return self.ring.get(timeout_s)

    def close(self):
        self.ring.close()


class GlooControl(Control):
    """Control messages as length-prefixed uint8 tensors over a gloo group (multi-node fallback)."""

    def __init__(self, group, src_rank: int, dst_rank: int, my_rank: int):
        self.group, self.src, self.dst, self.me = group, src_rank, dst_rank, my_rank

    def put(self, msg):
        import torch.distributed as dist
        n = torch.tensor([len(msg)], dtype=torch.int64)
        dist.send(n, self.dst, group=self.group)
        dist.send(torch.frombuffer(bytearray(msg), dtype=torch.uint8), self.dst, group=self.group)

    def get(self, timeout_s=None):
        import torch.distributed as dist
        n = torch.zeros(1, dtype=torch.int64)
        dist.recv(n, self.src, group=self.group)
        buf = torch.zeros(int(n.item()), dtype=torch.uint8)
        dist.recv(buf, self.src, group=self.group)
        return bytes(buf.numpy().tobytes())


# ---------------------------------------------------------------------------------------------
class PipelineLinks:
    """Everything one stage needs to talk to its neighbours."""

    def __init__(self, stage: int, num_stages: int):
        self.stage, self.num_stages = stage, num_stages
        self.prev_data: Optional[Edge] = None      # hidden states from stage-1
        self.next_data: Optional[Edge] = None      # hidden states to stage+1
        self.fb_data: Optional[Edge] = None        # tokens: last stage -> stage 0 (both ends hold it)
        self.prev_ctrl: Optional[Control] = None   # descriptors from stage-1
        self.next_ctrl: Optional[Control] = None   # descriptors to stage+1
        self.transport: Optional[str] = None       # set when the transport was chosen at run time (fallback chain)

    @property
    def is_first(self) -> bool:
        return self.stage == 0

    @property
    def is_last(self) -> bool:
        return self.stage == self.num_stages - 1

    def abort(self):
        """Abandon the data hops without waiting for transfers in flight (a peer is gone)."""
        for x in (self.prev_data, self.next_data, self.fb_data):
            if x is not None and hasattr(x, "abort"):
                try:
                    x.abort()
                except Exception:
                    pass

    def close(self):
        for x in (self.prev_data, self.next_data, self.fb_data, self.prev_ctrl, self.next_ctrl):
            if x is not None:
                try:
                    x.close()
                except Exception:
                    pass


def make_loopback_links(num_stages: int, device=None, max_bytes: int = 0, fb_bytes: int = 0, job: Optional[str] = None,
                        ctrl_slot_bytes: int = 1 << 17, timeout_s: float = 60.0) -> List[PipelineLinks]:
    """Links of an in-process pipeline. With a GPU `device` (and a `job` name for the rings) the hops
    are native device hops (`LocalDeviceEdge`: staging ring + events, `max_bytes` per hidden-state
    message, `fb_bytes` per token message) and the descriptors travel over shared-memory rings —
    the pieces the native stage loop is built on; otherwise in-process queues. `timeout_s`: how long
    a device hop's receive waits for its sender to have enqueued the message (a dead stage thread
    surfaces after this long)."""
    links = [PipelineLinks(s, num_stages) for s in range(num_stages)]
    native = device is not None and torch.device(device).type == "cuda" and job is not None and num_stages > 1
    if native:
        from .local_edge import LocalDeviceEdge
    for s in range(num_stages - 1):
        if native:
            e = LocalDeviceEdge(device, max_bytes, timeout_s=timeout_s, name=f"{s}->{s + 1}")
            cp = ShmControl(f"{job}_c{s}", create=True, slot_bytes=ctrl_slot_bytes)
            cc = ShmControl(f"{job}_c{s}", create=False, slot_bytes=ctrl_slot_bytes)
            links[s].next_ctrl, links[s + 1].prev_ctrl = cp, cc
        else:
            e, c = LoopbackEdge(), QueueControl()
            links[s].next_ctrl, links[s + 1].prev_ctrl = c, c
        links[s].next_data, links[s + 1].prev_data = e, e
    if num_stages > 1:
        fb = LocalDeviceEdge(device, max(fb_bytes, 256), timeout_s=timeout_s, name=f"{num_stages - 1}->0 (tokens)") if native else LoopbackEdge()
        links[0].fb_data = links[-1].fb_data = fb
        if native:
            for l in links:
                l.transport = "local-device"
    return links


def _warm_up_edges(edges, rank: int, backend: str):
    """One blocking 4-byte message along every edge, in the same global order on every rank.

    RCCL communicators are created lazily by the first send/recv on a group, and that creation
    blocks the host until the peer makes its matching first call. Inside the serving loop the two
    ends of the feedback edge reach their first call at unrelated times (stage 0 may pre-post the
    token receive before anything has gone down the chain), which can turn the lazy creation into
    a cycle of hosts waiting for each other. Doing it here, chain order then feedback, cannot:
    every rank's first blocking call is matched by a rank that has already finished its own."""
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    for g, src, dst in edges:
        if rank == src:
            dist.send(torch.full((1,), src, dtype=torch.int32, device=dev), dst, group=g)
        elif rank == dst:
            t = torch.zeros(1, dtype=torch.int32, device=dev)
            dist.recv(t, src, group=g)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            if int(t.item()) != src:
                raise RuntimeError(f"pipeline edge {src}->{dst}: warm-up message corrupted ({int(t.item())})")
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def _attach_shm_ctrl(links: PipelineLinks, rank: int, world: int, job: str, slot_bytes: int = 1 << 17):
    # producer creates, consumer attaches (with retry) — no ordering requirement between them
    if rank < world - 1:
        links.next_ctrl = ShmControl(f"{job}_c{rank}", create=True, slot_bytes=slot_bytes)
    if rank > 0:
        links.prev_ctrl = ShmControl(f"{job}_c{rank - 1}", create=False, slot_bytes=slot_bytes)


def make_rccl_links(rank: int, world: int, job: str, device, ctrl_slot_bytes: int = 1 << 17,
                    warmup_timeout_s: float = 120.0) -> PipelineLinks:
    """Native RCCL hops for rank `rank` of a `world`-stage pipeline (one process per GPU).

    `torch.distributed` must be initialised with a CPU backend (gloo): it only carries the
    communicator ids and barriers, never activations. Rank 0 draws one id per hop and broadcasts
    them; every hop is a two-rank communicator (sender = rank 0 inside it). Communicator creation
    blocks until the peer joins, so all ranks open their hops in one global order — the chain from
    stage 0 upwards, then the feedback hop — in which every rank's next blocking call is matched
    by a rank that has already finished its own earlier ones. One 4-byte message then crosses every
    hop (RCCL sets a peer connection up on first use; that must not happen inside a captured
    graph or at an arbitrary point of the serving loop) and is checked on arrival."""
    import torch.distributed as dist
    from . import rccl
    links = PipelineLinks(rank, world)
    n_hops = world if world > 1 else 0              # world - 1 forward hops + the feedback hop
    ids = [rccl.unique_id() for _ in range(n_hops)] if rank == 0 else [None] * n_hops
    if n_hops:
        dist.broadcast_object_list(ids, src=0)
    dev = torch.device(device)
    opened = []                                      # (edge, src stage, dst stage)
    for s in range(world - 1):
        if rank == s:
            links.next_data = rccl.RcclEdge(ids[s], 2, 0, 1, dev, name=f"{s}->{s + 1}")
            opened.append((links.next_data, s, s + 1))
        elif rank == s + 1:
            links.prev_data = rccl.RcclEdge(ids[s], 2, 1, 0, dev, name=f"{s}->{s + 1}")
            opened.append((links.prev_data, s, s + 1))
    if world > 1 and rank in (0, world - 1):
        last = world - 1
        links.fb_data = rccl.RcclEdge(ids[last], 2, 0 if rank == last else 1, 1 if rank == last else 0, dev,
                                      name=f"{last}->0 (tokens)")
        opened.append((links.fb_data, last, 0))
    for edge, src, dst in opened:
        if rank == src:
            edge.send(torch.full((1,), src + 1, dtype=torch.int32, device=dev)).sync(warmup_timeout_s)
        else:
            t = torch.zeros(1, dtype=torch.int32, device=dev)
            edge.recv(t).sync(warmup_timeout_s)
            if int(t.item()) != src + 1:
                raise RuntimeError(f"pipeline hop {edge.name}: warm-up message corrupted ({int(t.item())})")
    _attach_shm_ctrl(links, rank, world, job, ctrl_slot_bytes)
    return links


def make_dist_links(rank: int, world: int, backend: str, job: str, use_shm_ctrl: bool = True,
                    host_staged: bool = False, ctrl_slot_bytes: int = 1 << 17) -> PipelineLinks:
    """Process groups and control rings for rank `rank` of a `world`-stage pipeline.

    `torch.distributed` must already be initialised (any backend; sub-groups are created with
    `backend`). new_group() is collective over the default group, so every rank creates every
    edge's group in the same order, then keeps only its own. `host_staged`: the stages compute on
    GPUs but the groups are gloo (TCP between nodes without an RCCL fabric, or several stages
    rehearsing on one GPU) — activations bounce through pinned host memory."""
    import torch.distributed as dist
    links = PipelineLinks(rank, world)
    edges = []
    for s in range(world - 1):
        g = dist.new_group(ranks=[s, s + 1], backend=backend)
        edges.append((g, s, s + 1))
        if rank == s:
            links.next_data = TorchDistEdge(g, s, s + 1, rank, host_staged)
        elif rank == s + 1:
            links.prev_data = TorchDistEdge(g, s, s + 1, rank, host_staged)
    if world > 1:
        g = dist.new_group(ranks=[0, world - 1], backend=backend)     # feedback hop, own communicator
        edges.append((g, world - 1, 0))
        if rank in (0, world - 1):
            links.fb_data = TorchDistEdge(g, world - 1, 0, rank, host_staged)
    _warm_up_edges(edges, rank, backend)
    if use_shm_ctrl:
        _attach_shm_ctrl(links, rank, world, job, ctrl_slot_bytes)
    else:
        for s in range(world - 1):
            g = dist.new_group(ranks=[s, s + 1], backend="gloo")
            if rank == s:
                links.next_ctrl = GlooControl(g, s, s + 1, rank)
            elif rank == s + 1:
                links.prev_ctrl = GlooControl(g, s, s + 1, rank)
    return links
