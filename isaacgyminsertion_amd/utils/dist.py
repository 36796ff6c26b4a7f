"""One process per GPU (train.py:58-64, frozen_ppo.py:116-126): rank / world from the launcher's environment,
device = the local rank's GPU, process group over RCCL ("nccl" is RCCL on ROCm).

``IGI_DIST_BACKEND=gloo`` keeps the whole multi-rank control flow testable on a single-GPU box: ranks share the
devices round-robin and the collectives run through gloo (which carries device tensors)."""
import os

# dmabuf IPC is the only mode the host driver supports; the ROCm runtime reads this when HSA initialises, i.e. at the
# first GPU call of the process -- so it is set at import, before anything here (or after it) touches a device.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_rank_device():
    """-> (local_rank, world_size, device string); initialises the default process group once."""
    rank = int(os.getenv("LOCAL_RANK", "0"))
    world = int(os.getenv("WORLD_SIZE", "1"))
    backend = os.environ.get("IGI_DIST_BACKEND", "nccl")
    index = rank if backend == "nccl" else rank % max(torch.cuda.device_count(), 1)
    device = "cuda:" + str(index)
    torch.cuda.set_device(index)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", rank=int(os.getenv("RANK", rank)), world_size=world,
                                    device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=int(os.getenv("RANK", rank)), world_size=world)
    return rank, world, device


class NativeComm:
    """The library's own RCCL communicator for this rank (csrc/comm.h): an ``ncclComm_t`` + a communication stream +
    the events that fence it against the compute stream, created on the CURRENT device.  Rank 0 draws the 128-byte id;
    it reaches the other ranks through the process group that is already up (``torch.distributed``: any backend --
    the launcher's rendezvous is the only piece of torch.distributed the data-parallel update still needs).  With
    ``world == 1`` nothing else is required: a one-rank communicator on one GPU exercises every RCCL call of the update.

    ``ident``: the 128 id bytes when the caller has already distributed them (``native_comm_or_none`` does, so that a
    rank that fails never leaves the others inside a collective it skipped)."""

    def __init__(self, rank=None, world=None, ident=None):
        import ctypes as C
        from .. import _lib
        L = _lib.lib()
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if ident is None:
            ident = self.draw_id() if rank == 0 else None
            if world > 1:
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                ident = box[0]
            if ident is None:
                raise RuntimeError("igi_comm_unique_id failed on rank 0")
        buf = (C.c_char * 128).from_buffer_copy(ident)
        h = C.c_void_p()
        rc = L.igi_comm_create(buf, int(rank), int(world), C.byref(h))
        if rc != 0:
            raise RuntimeError("igi_comm_create: " + (L.igi_comm_last_error(None) or b"").decode() + " / " +
                               L.igi_last_error().decode())
        self.handle, self.rank, self.world = h.value, int(rank), int(world)
        self._L = L

    @staticmethod
    def draw_id():
        """128 bytes of a fresh ncclUniqueId, or None when RCCL refuses (the caller broadcasts either)."""
        import ctypes as C
        from .. import _lib
        ident = (C.c_char * 128)()
        try:
            if _lib.lib().igi_comm_unique_id(ident) != 0:
                return None
        except Exception:   # noqa: BLE001  (library missing on this rank: the vote below turns the native path off)
            return None
        return bytes(ident.raw)

    def all_reduce_(self, t):
        """in place, SUM, on the current stream (fp32 contiguous tensor)"""
        import ctypes as C
        from .. import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        rc = self._L.igi_comm_all_reduce_sum_f32(C.c_void_p(self.handle), C.c_void_p(t.data_ptr()), t.numel(),
                                                 C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
        if rc == -6:
            raise RuntimeError("RCCL: " + self._L.igi_comm_last_error(C.c_void_p(self.handle)).decode())
        _lib.check(rc, "igi_comm_all_reduce_sum_f32")
        return t

    def all_reduce_async_(self, t):
        """in place, SUM, on the communicator's own stream behind everything enqueued on the current stream so far; work
        enqueued on the current stream afterwards overlaps it.  ``join()`` before reading ``t``."""
        import ctypes as C
        from .. import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        rc = self._L.igi_comm_all_reduce_async_f32(C.c_void_p(self.handle), C.c_void_p(t.data_ptr()), t.numel(),
                                                   C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
        if rc == -6:
            raise RuntimeError("RCCL: " + self._L.igi_comm_last_error(C.c_void_p(self.handle)).decode())
        _lib.check(rc, "igi_comm_all_reduce_async_f32")
        return t

    def join(self, device=None):
        """the current stream waits for the last ``all_reduce_async_``"""
        import ctypes as C
        from .. import _lib
        _lib.check(self._L.igi_comm_join(C.c_void_p(self.handle),
                                         C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "igi_comm_join")

    def broadcast_(self, t, root=0):
        """in place broadcast of a contiguous device tensor from ``root`` (the parameter broadcast of
        frozen_ppo.py:376-381 as one flat vector)"""
        import ctypes as C
        from .. import _lib
        assert t.is_cuda and t.is_contiguous()
        rc = self._L.igi_comm_broadcast(C.c_void_p(self.handle), C.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                        int(root), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
        if rc == -6:
            raise RuntimeError("RCCL: " + self._L.igi_comm_last_error(C.c_void_p(self.handle)).decode())
        _lib.check(rc, "igi_comm_broadcast")
        return t

    def rccl_ranks(self):
        """``ncclCommCount`` of this communicator: the number of ranks RCCL itself connected (``world`` is what the caller
        passed in)."""
        import ctypes as C
        n = self._L.igi_comm_count(C.c_void_p(self.handle))
        if n < 0:
            raise RuntimeError("RCCL: " + self._L.igi_comm_last_error(C.c_void_p(self.handle)).decode())
        return int(n)

    def rccl_version(self):
        """ncclGetVersion as 'major.minor.patch'"""
        v = int(self._L.igi_rccl_version())
        return f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v > 0 else None

    def close(self):
        import ctypes as C
        if getattr(self, "handle", None):
            self._L.igi_comm_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _all_agree(ok, device):
    """MIN over the ranks of a 0 / 1 flag through the launcher's process group."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def native_comm_or_none(device, world):
    """The library's RCCL communicator for this rank when the process group runs over RCCL ("nccl") and
    IGI_DP_NATIVE != 0; None otherwise -- callers then use torch.distributed collectives.  Every rank returns the same
    answer, and no rank can be left waiting in a collective another rank skipped:

      1. rank 0 draws the id and ALWAYS broadcasts (the id, or None when drawing failed) -- every rank takes part;
      2. vote: every rank has an id and can load the library, else None everywhere (nothing native was entered);
      3. ``igi_comm_create`` on every rank (``ncclCommInitRank`` is itself a rendezvous with RCCL's own time-out: a
         rank that fails in it fails the others too), then a vote on the outcome; ranks that succeeded while another
         failed destroy their communicator;
      4. probe: a SUM of ones must give the world size (all ranks hold a communicator here), then the last vote.

    Used by the trainers and by bench.py (one helper: round 3 carried a copy in bench.py)."""
    if os.environ.get("IGI_DP_NATIVE", "1") == "0" or not dist.is_initialized() or dist.get_backend() != "nccl":
        return None
    rank = dist.get_rank()
    box = [NativeComm.draw_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)                       # (1)
    ident = box[0]
    lib_ok = True
    try:
        from .. import _lib
        _lib.lib()
    except Exception:   # noqa: BLE001
        lib_ok = False
    if not _all_agree(ident is not None and lib_ok, device):     # (2)
        return None
    comm = None
    try:
        comm = NativeComm(rank=rank, world=world, ident=ident)   # (3)
    except Exception:   # noqa: BLE001
        comm = None
    if not _all_agree(comm is not None, device):
        if comm is not None:
            comm.close()
        return None
    ok = False
    try:
        probe = torch.ones(4, dtype=torch.float32, device=device)
        comm.all_reduce_(probe)                                  # (4)
        ok = bool((probe == float(world)).all().item())
    except Exception:   # noqa: BLE001
        ok = False
    if not _all_agree(ok, device):
        comm.close()
        return None
    return comm
