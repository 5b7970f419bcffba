"""Collectives of the data-parallel gradient exchange (one process per GPU).

Two backends with one interface (in-place collectives over contiguous slices of the flat gradient / parameter buffers):

  * RcclColl  -- the library's own RCCL communicator behind the C ABI (include/vct_hip.h, vct_comm_*): collectives run on
    a stream the communicator owns, ordered behind the compute stream by event edges, recordable into launch lists.  The
    128-byte unique id travels through the torch.distributed process group the launcher already set up (store / any
    backend); nothing else of torch.distributed is on the data path.
  * C10dColl  -- the same operations through torch.distributed (gloo on CPU or for several ranks sharing one GPU in the
    tests; any backend).  Synchronous with respect to the current stream.

replaces: DistributedDataParallel's reducer (reference train.py:217-219, utils.py:137-146)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class C10dColl:
    """Host-side collectives.  Inside a recorded launch list (ops.LaunchList) each call becomes a host command of the list
    (ops.host_call): at that point of every replay the stream it was issued on is drained, the collective runs on the host
    thread, and the device is idle again before the next recorded launch -- slow, but it lets the recorded executor carry the
    exchange in the one-GPU multi-rank tests exactly as it carries RCCL's stream-side collectives on a real node."""
    owns_stream = False
    recordable = True

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stream = None

    def _run(self, fn, t):
        from . import ops
        if t.is_cuda and ops.is_recording():
            def at_replay():
                fn()
                torch.cuda.synchronize(t.device)
            ops.host_call(at_replay)
        else:
            if t.is_cuda:      # host-side collective on device buffers: everything enqueued so far on this stream has to be done
                torch.cuda.current_stream(t.device).synchronize()
            fn()

    def allreduce_avg(self, t, after=None):
        if self.world > 1:
            def go():
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.mul_(1.0 / self.world)
            self._run(go, t)

    def reduce_scatter_avg(self, t, n, after=None):
        """Own shard t[rank*n:(rank+1)*n) holds the mean afterwards (the other shards are unspecified)."""
        self.allreduce_avg(t)

    def all_gather(self, t, n, after=None):
        if self.world > 1:
            def go():
                mine = t[self.rank * n:(self.rank + 1) * n].clone()
                dist.all_gather([t[q * n:(q + 1) * n] for q in range(self.world)], mine, group=self.group)
            self._run(go, t)

    def broadcast(self, t, root=0, after=None):
        if self.world > 1:
            self._run(lambda: dist.broadcast(t, src=root, group=self.group), t)

    def wait(self, stream=None):
        pass

    def close(self):
        pass


class RcclColl:
    owns_stream = True
    recordable = True

    def __init__(self, group=None, device=None):
        lib = L.load()
        if not lib.vct_comm_available():
            raise RuntimeError("no RCCL library could be bound (vct_comm_available() == 0)")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            L.check(lib.vct_comm_unique_id(ident), "vct_comm_unique_id")
        box = [bytes(ident)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = L.vp()
        L.check(lib.vct_comm_init(ident, self.rank, self.world, C.byref(h)), "vct_comm_init")
        self._h = h
        s = L.vp()
        L.check(lib.vct_comm_stream(h, C.byref(s)), "vct_comm_stream")
        self.stream = torch.cuda.ExternalStream(s.value, device=device)
        self.lib_world = int(lib.vct_comm_world(h))          # ranks the RCCL communicator itself reports

    @staticmethod
    def _after(after):
        if after is None:
            return L.stream_ptr(), 1
        if after is False:
            return 0, 0
        return after.cuda_stream, 1

    def allreduce_avg(self, t, after=None):
        st, o = self._after(after)
        L.check(L.load().vct_comm_allreduce_avg(self._h, t.data_ptr(), t.numel(), L.dtype_code(t.dtype), st, o), "vct_comm_allreduce_avg")

    def reduce_scatter_avg(self, t, n, after=None):
        assert t.numel() == n * self.world
        st, o = self._after(after)
        L.check(L.load().vct_comm_reduce_scatter_avg(self._h, t.data_ptr(), n, L.dtype_code(t.dtype), st, o), "vct_comm_reduce_scatter_avg")

    def all_gather(self, t, n, after=None):
        assert t.numel() == n * self.world
        st, o = self._after(after)
        L.check(L.load().vct_comm_all_gather(self._h, t.data_ptr(), n, L.dtype_code(t.dtype), st, o), "vct_comm_all_gather")

    def broadcast(self, t, root=0, after=None):
        st, o = self._after(after)
        L.check(L.load().vct_comm_broadcast(self._h, t.data_ptr(), t.numel(), L.dtype_code(t.dtype), int(root), st, o), "vct_comm_broadcast")

    def wait(self, stream=None):
        L.check(L.load().vct_comm_wait(self._h, stream.cuda_stream if stream is not None else L.stream_ptr()), "vct_comm_wait")

    def close(self):
        if self._h is not None and self._h.value:
            L.load().vct_comm_destroy(self._h)
            self._h = None

    def self_test(self) -> bool:
        """All-reduce / reduce-scatter / all-gather of a small pattern, checked against the closed form: every rank must
        agree before the communicator carries gradients."""
        W, r = self.world, self.rank
        n = 256
        dev = self.stream.device
        x = torch.arange(n * W, dtype=torch.float32, device=dev) + 1000.0 * r
        self.allreduce_avg(x)
        y = torch.arange(n * W, dtype=torch.float32, device=dev) * (r + 1)
        self.reduce_scatter_avg(y, n)
        z = torch.full((n * W,), -1.0, device=dev)
        z[r * n:(r + 1) * n] = float(r)
        self.all_gather(z, n, after=None)
        self.wait()
        torch.cuda.synchronize()
        base = torch.arange(n * W, dtype=torch.float32, device=dev)
        ok = torch.allclose(x, base + 1000.0 * (W - 1) / 2.0)
        ok &= torch.allclose(y[r * n:(r + 1) * n], base[r * n:(r + 1) * n] * (W + 1) / 2.0)
        ok &= torch.equal(z, torch.arange(W, device=dev, dtype=torch.float32).repeat_interleave(n))
        return bool(ok)
