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


class P2PComm:
    """One-shot peer-to-peer all-reduce behind the C ABI (pmbrl_p2p_*, csrc/pmbrl_p2p.hip) for the latency-bound
    messages of a sharded run: every rank writes into IPC-mapped slots of every peer and sums in rank order, one kernel
    per rank on the compute stream.  The 64-byte IPC handles travel through the torch.distributed group the caller
    already has (bootstrap only).  Ranks of one node: on different GPUs, or several on one GPU."""

    def __init__(self, group=None, device=None, max_bytes=1 << 20):
        import torch.distributed as dist
        self.lib = _lib.load()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.device = dev
        self.max_bytes = int(max_bytes)
        self.p2p = C.c_void_p()
        _lib.check(self.lib.pmbrl_p2p_create(self.rank, self.world, dev.index or 0, int(max_bytes), C.byref(self.p2p)),
                   'pmbrl_p2p_create')
        hb = C.create_string_buffer(64)
        _lib.check(self.lib.pmbrl_p2p_handle(self.p2p, hb), 'pmbrl_p2p_handle')
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(hb.raw), group=group)
        for q, h in enumerate(handles):
            if q != self.rank:
                _lib.check(self.lib.pmbrl_p2p_open(self.p2p, q, C.c_char_p(h)), 'pmbrl_p2p_open')
        dist.barrier(group=group)      # every rank has mapped every peer before the first store

    def allreduce_(self, t):
        """In-place sum over the ranks of a contiguous fp32 / fp64 device tensor (same bits on every rank)."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        fn = self.lib.pmbrl_p2p_allreduce_f32 if t.dtype == torch.float32 else self.lib.pmbrl_p2p_allreduce_f64
        _lib.check(fn(self.p2p, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(t.data_ptr()), t.numel()),
                   'pmbrl_p2p_allreduce')
        return t

    def __call__(self, view):          # a transport for Engine.attach_collective (groups spread over ranks)
        return self.allreduce_(view)

    def failed(self):
        """Host sync: did a wait time out (a peer never arrived) since the last call?"""
        e = C.c_int32(0)
        _lib.check(self.lib.pmbrl_p2p_error(self.p2p, C.byref(e)), 'pmbrl_p2p_error')
        return bool(e.value)

    def close(self):
        if self.p2p:
            self.lib.pmbrl_p2p_destroy(self.p2p)
            self.p2p = C.c_void_p()


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


_P2PS = {}


def get_p2p(group=None, device=None, max_bytes=4 << 20):
    """The one-shot peer-to-peer transport for this process group (created on first use; collective).  Chosen with
    PMBRL_P2P=1 for the latency-bound messages of a sharded run (the flat gradient up to max_bytes, the per-step
    statistics of groups spread over ranks); all ranks must sit on one node."""
    key = (id(group), str(device))
    if key not in _P2PS:
        _P2PS[key] = P2PComm(group, device, max_bytes=max_bytes)
    return _P2PS[key]


def _env_int(name, default=0):
    v = os.environ.get(name)
    if v is None or v.strip() == '':
        return default
    try:
        return int(v)
    except ValueError:
        return 1 if v.strip().lower() in ('true', 'yes', 'on') else 0


def p2p_wanted():
    """PMBRL_P2P=1 selects the peer-to-peer transport; 0 / unset / empty leave RCCL (parsed like the library's own
    integer switches, PMBRL_LDS_TILES / PMBRL_MM_XCH)."""
    return _env_int('PMBRL_P2P') != 0


def p2p_check(group=None, device=None):
    """Host sync + collective, once per iteration of a sharded run on the peer-to-peer transport: did ANY rank's wait
    for a peer time out since the last check (pmbrl_p2p_error)?  A timed-out exchange returns with its buffer
    un-reduced -- the local gradient, or local moment-matching statistics, as if they were the sum -- so every rank
    raises together instead of letting the replicas drift apart.  No-op on the RCCL transport (which simply waits)."""
    import torch.distributed as dist
    if not p2p_wanted() or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    key = (id(group), str(device))
    p2p = _P2PS.get(key)
    # (a rank without the cached object still ENTERS the collective below: p2p_wanted() is the same on every rank, the
    #  cache lookup need not be -- a rank that returned here would leave the others waiting in the all-reduce)
    bad = 1 if (p2p is not None and p2p.failed()) else 0
    if dist.get_backend(group) == 'nccl':
        fl = torch.tensor([bad], dtype=torch.int32, device=device)
        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
    else:
        fl = torch.tensor([bad], dtype=torch.int32)
        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=group)
    if int(fl[0]):
        raise RuntimeError('peer-to-peer exchange: a rank waited for a peer that did not arrive in time (the exchange '
                           'was not completed; PMBRL_P2P_TIMEOUT_SPINS sets the wait)')


def grad_allreduce(group=None, device=None):
    """The in-place sum all-reduce of the flat gradient for this process group: through the C ABI on
    the compute stream when the group runs on RCCL (backend nccl), through torch.distributed
    otherwise (gloo: CPU tests and the one-device debugging mode of bench.py) or when RCCL cannot be
    initialised here (reported once)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lambda t: t
    if p2p_wanted():
        p2p = get_p2p(group, device)
        fallback = get_comm(group, device)

        def allreduce(t):      # (messages beyond the slots: the ring, which is bandwidth-bound territory anyway)
            if t.numel() * t.element_size() <= p2p.max_bytes:
                return p2p.allreduce_(t)
            return fallback.allreduce_(t) if fallback is not None else allreduce_sum_(t, group)
        return allreduce
    comm = get_comm(group, device)
    if comm is not None:
        return comm.allreduce_
    return lambda t: allreduce_sum_(t, group)
