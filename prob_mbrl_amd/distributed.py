"""Multi-GPU decomposition of the rollout: particle rows are independent except
inside a moment-matching group, so rank r owns a contiguous block of whole groups
(row = particle*S + sample, group = particle).  Weights, reward constants and the
discount vector are replicated; every rank produces a partial flat policy
gradient already scaled by 1/B_global; ONE sum all-reduce (RCCL over xGMI,
`backend="nccl"`) per optimiser iteration; then the identical clip + Adam runs
on every rank (deterministic replicas, no parameter broadcast)."""
import ctypes as C
import os

import torch

from . import _lib


def shard_bounds(B, mm_groups, world, rank):
    """[lo, hi) rows of `rank`: whole groups, as even as possible."""
    G = mm_groups if mm_groups else B
    if B % G:
        raise ValueError('B must be divisible by mm_groups')
    M = B // G
    base, extra = divmod(G, world)
    g_lo = rank * base + min(rank, extra)
    g_hi = g_lo + base + (1 if rank < extra else 0)
    return g_lo * M, g_hi * M


def local_groups(B, mm_groups, world, rank):
    lo, hi = shard_bounds(B, mm_groups, world, rank)
    if not mm_groups:
        return None
    return (hi - lo) // (B // mm_groups)


def allreduce_sum_(t, group=None):
    """In-place sum all-reduce of the flat gradient (no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def max_over_ranks(x, device, group=None):
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class GradComm:
    """RCCL communicator behind the C ABI (pmbrl_comm_*): the gradient all-reduce is issued on the
    compute stream right behind the adjoint sweep's dW reduction -- one library call, no second
    stream, no event hand-off.  Rank 0 creates the RCCL id; it travels to the other ranks through
    the torch.distributed group the caller already has (used for bootstrap only)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.lib = _lib.load()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        idbuf = C.create_string_buffer(128)
        box = [None]
        if self.rank == 0:
            # rank 0 ALWAYS takes part in the broadcast below: the id, or the reason it has none -- raising before it
            # would leave the other ranks waiting in the broadcast while rank 0 moves on to a different collective
            try:
                _lib.check(self.lib.pmbrl_comm_unique_id(idbuf), 'pmbrl_comm_unique_id')
                box = [bytes(idbuf.raw)]
            except Exception as e:      # noqa: BLE001
                box = ['error: %s' % e]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                       group=group)
        self.comm = C.c_void_p()
        if not isinstance(box[0], bytes):
            raise RuntimeError('rank 0 could not create the RCCL id (%s)' % box[0])
        _lib.check(self.lib.pmbrl_comm_init(C.c_char_p(box[0]), self.rank, self.world, dev.index or 0,
                                            C.byref(self.comm)), 'pmbrl_comm_init')

    def allreduce_(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.pmbrl_allreduce_sum(self.comm, C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                C.c_void_p(t.data_ptr()), t.numel()), 'pmbrl_allreduce_sum')
        return t

    def count(self):
        """Ranks as the RCCL communicator itself reports them (ncclCommCount)."""
        n = C.c_int32(0)
        _lib.check(self.lib.pmbrl_comm_count(self.comm, C.byref(n)), 'pmbrl_comm_count')
        return int(n.value)

    def close(self):
        if self.comm:
            self.lib.pmbrl_comm_destroy(self.comm)
            self.comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_COMMS = {}


def get_comm(group=None, device=None):
    """The RCCL communicator behind the C ABI for this process group (one per group and device, created on first
    use; collective), or None when the group does not run on RCCL (gloo: CPU tests and the one-device debugging
    mode of bench.py) or RCCL cannot be initialised here (reported once)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    key = (id(group), str(device))
    if key not in _COMMS:
        comm = None
        if dist.get_backend(group) == 'nccl' and not os.environ.get('PMBRL_TORCH_ALLREDUCE'):
            try:
                comm = GradComm(group, device)
            except Exception as e:      # noqa: BLE001 -- any failure here must not take the run down
                if dist.get_rank(group) == 0:
                    print('prob_mbrl_amd: RCCL communicator through the C ABI unavailable (%s); '
                          'using torch.distributed.all_reduce' % e)
            # every rank must take the same path: one that failed alone would leave the others in a collective
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32,
                              device=device if device is not None else 'cuda:%d' % torch.cuda.current_device())
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
        _COMMS[key] = comm
    return _COMMS[key]


def grad_allreduce(group=None, device=None):
    """The in-place sum all-reduce of the flat gradient for this process group: through the C ABI on
    the compute stream when the group runs on RCCL (backend nccl), through torch.distributed
    otherwise (gloo: CPU tests and the one-device debugging mode of bench.py) or when RCCL cannot be
    initialised here (reported once)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lambda t: t
    comm = get_comm(group, device)
    if comm is not None:
        return comm.allreduce_
    return lambda t: allreduce_sum_(t, group)
