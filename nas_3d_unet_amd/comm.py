"""The data-parallel exchange step's communicator: RCCL through the C ABI (n3d_comm_*), ONE per (process group, device) and shared
by every trainer of the process.

Why not torch.distributed's own collectives for the step (round 5, profiles/r05_dp_capture_loop.log): ProcessGroupNCCL keeps a
Work object per collective whose end event its watchdog thread polls with hipEventQuery every 100 ms until the work is retired.
HIP refuses that query with hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing stream")
when the stream the event was LAST RECORDED ON is capturing -- also when the record itself happened long before the capture began.
The bucketed schedule issues its all-reduces on the weight-gradient stream and then captures that stream: a watchdog poll that
falls between the warm-up's collective and its retirement throws inside the watchdog thread, which torch turns into
std::terminate -- the silent SIGABRT of round 4 (35 of 200 fresh-process captures without the 0.25 s sleep that hid it).
A collective issued here has no Work object, no event and no thread that looks at it later: nothing is left to collide with a capture.

torch.distributed is still what the caller initialises (rank, world size, rendezvous): the 128-byte ncclUniqueId travels through
its key-value store, no torch collective is issued.  CPU tensors (gloo, the CPU test-suite) and CUDA tensors on a non-NCCL group
keep torch.distributed.all_reduce (`for_group` returns None)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib

_cache = {}      # id(process group object) -> (the group object: keeps the id from being reused, device index, Comm)
_serial = {}     # group name -> communicators made so far (every rank makes them in the same order: the store key)


def _is_nccl(pg):
    try:
        return "nccl" in str(dist.get_backend(pg))
    except Exception:
        return False


class Comm:
    """an RCCL communicator over the ranks of a torch.distributed process group, bound to one device"""

    def __init__(self, pg, device):
        self.device = torch.device(device)
        self.rank = dist.get_rank(pg)
        self.world = dist.get_world_size(pg)
        lib = _lib.load()
        buf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(lib.n3d_comm_unique_id(buf), "n3d_comm_unique_id")
        idb = (C.c_char * 128).from_buffer_copy(self._exchange_id(pg, bytes(buf.raw)))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):     # the communicator binds to the current HIP device
            _lib.check(lib.n3d_comm_init(idb, self.world, self.rank, C.byref(handle)), "n3d_comm_init")
        self._h = handle
        self._scratch = torch.zeros(4, dtype=torch.float32, device=self.device)

    def _exchange_id(self, pg, mine):
        """rank 0's unique id to every rank of the group, through the rendezvous store (no collective, nothing the NCCL watchdog
        would have to retire); without a reachable store: broadcast_object_list"""
        if self.world == 1:
            return mine
        name = getattr(pg, "group_name", None) if pg is not None else "default"
        n = _serial[name] = _serial.get(name, 0) + 1
        try:
            store = dist.distributed_c10d._get_default_store()
            key = "n3d_comm/%s/%d" % (name, n)
            if self.rank == 0:
                store.set(key, mine)
                return mine
            return bytes(store.get(key))
        except Exception:
            box = [mine]
            src = dist.get_global_rank(pg, 0) if pg is not None else 0
            dist.broadcast_object_list(box, src=src, group=pg)
            return box[0]

    def _stream(self, stream):
        return C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)

    def allreduce_sum_ptr(self, ptr, n, stream=None):
        """in-place SUM all-reduce of n fp32 words at device address ptr, stream-ordered (default: the current stream)"""
        _lib.check(_lib.load().n3d_comm_allreduce_sum(self._h, C.c_void_p(ptr), n, self._stream(stream)), "n3d_comm_allreduce_sum")

    def allreduce_sum(self, t, stream=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self.allreduce_sum_ptr(t.data_ptr(), t.numel(), stream)

    def broadcast(self, t, root=0, stream=None):
        """in-place broadcast of a contiguous fp32 tensor from group rank `root`"""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(_lib.load().n3d_comm_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel(), int(root), self._stream(stream)), "n3d_comm_broadcast")

    def all_true(self, flag):
        """True only if `flag` is true on every rank (one 4-float all-reduce and a host read)"""
        self._scratch.fill_(0.0 if flag else 1.0)
        self.allreduce_sum(self._scratch)
        return float(self._scratch[0].item()) == 0.0

    def gather_floats(self, value):
        """[every rank's `value`] on every rank (one all-reduce of `world` floats, each rank filling its own slot, and a host read)"""
        n = (self.world + 3) // 4 * 4
        v = torch.zeros(n, dtype=torch.float32, device=self.device)
        v[self.rank] = float(value)
        self.allreduce_sum(v)
        return [float(a) for a in v[:self.world].tolist()]

    def barrier(self):
        """every rank has reached this point (and this rank's current stream has drained)"""
        self.all_true(True)

    def close(self):
        h, self._h = self._h, None
        if h is not None:
            try:
                _lib.load().n3d_comm_destroy(h)
            except Exception:
                pass


def for_group(pg, device):
    """the process's communicator for (process group, device); None when the exchange has to stay with torch.distributed (not
    initialised, CPU tensors, or a group whose backend is not NCCL/RCCL -- e.g. two gloo processes sharing one GPU)"""
    device = torch.device(device)
    if device.type != "cuda" or not dist.is_initialized() or not _is_nccl(pg):
        return None
    g = pg if pg is not None else dist.distributed_c10d._get_default_group()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    hit = _cache.get((id(g), idx))
    if hit is not None:
        return hit[1]
    c = Comm(pg, torch.device("cuda", idx))
    _cache[(id(g), idx)] = (g, c)
    return c


def close_all():
    """destroy every communicator made here (before dist.destroy_process_group at the end of a program; optional)"""
    for _, c in list(_cache.values()):
        c.close()
    _cache.clear()
