"""Data parallelism for the hot path: one process per GPU, utterances sharded by rank, gradients summed range by
range (RCCL over xGMI).

The reference trains on one device only (`training_GPUs=[0]`, ecog2txt/trainers.py:131); this exchange step is what
SURVEY.md 8e adds.  Buckets are contiguous ranges of the flat gradient buffer in the order backward produces them
(vocab projection/decoder -> encoder top ... bottom -> conv), so each all-reduce is issued as soon as its stage of
backward has been enqueued and runs on the communicator's stream underneath the remaining BPTT launches.  xGMI is
point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound: buckets are kept large (one per
backward stage, 7-20 MB at the default sizes) rather than many small ones.

Two transports with one interface (`allreduce_range`, `wait`, `pending_ranges`, `world`, `grad_scale`):
  * `RcclSync`  -- the product path on GPUs: librccl called directly through the C ABI (e2t_comm_*), on a stream the
    communicator owns, ordered by events; torch.distributed is not involved in the step.  The 128-byte unique id is
    handed from rank 0 to the others through a torch TCPStore (or an existing process group) -- bootstrap only.
  * `GradSync`  -- torch.distributed ("gloo" in the CPU tests, where there is no device to run RCCL on).

Sharding (`global_batches` / `rank_slice`): every GLOBAL batch of world x B utterances is cut into `world` slices of
B, so all ranks run the same number of steps whatever n is; a rank whose slice is short or empty pads with
zero-length utterances (the kernels ignore them) and still takes part in every collective.  Losses are normalised by
the GLOBAL token counts of the batch (every rank holds the host copy of the targets and can count), so the SUM of the
ranks' gradients is exactly the gradient of the global mean loss -- `grad_scale` is then 1.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items utterances for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_batches(n, B, world, rng=None):
    """Index arrays of the global batches (world * B utterances each, the last one possibly short) of one pass over
    n utterances; the permutation comes from `rng`, which every rank seeds alike."""
    order = np.arange(n) if rng is None else rng.permutation(n)
    G = B * world
    return [order[i:i + G] for i in range(0, n, G)]


def rank_slice(idx, B, rank, world=None, lengths=None):
    """Rank `rank`'s share of one global batch.  Without `lengths`: utterances [rank*B, (rank+1)*B) of it (possibly fewer,
    possibly none).  With `lengths` (per-utterance sample counts of the WHOLE partition, indexed like `idx`) and `world`:
    the global batch is dealt out by length -- sorted, longest first, rank r takes every world-th one -- so every rank holds
    the same mix of long and short utterances and a short last batch is spread over all ranks instead of leaving the last
    ones empty ("bucket utterances by length before sharding").  The captured step itself costs the same
    whatever the lengths are (fixed shapes: T of the partition, every step of every layer runs); what the dealing balances is
    the number of VALID samples and tokens per rank, i.e. each rank's share of the globally normalised loss."""
    if lengths is None or world is None:
        return idx[rank * B:(rank + 1) * B]
    idx = np.asarray(idx)
    order = idx[np.argsort(-np.asarray(lengths)[idx], kind='stable')]
    return order[rank::world][:B]


class GradSync:
    """torch.distributed transport (gloo on CPU; also works on "nccl")."""

    def __init__(self, flat_grad, group=None, sum_of_global_means=False):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.pending = []
        self.pending_ranges = []           # (work, a, b) in issue order: lets the optimiser follow range by range
        self.sum_of_global_means = sum_of_global_means

    def allreduce_range(self, a, b):
        """Asynchronously sum flat_grad[a:b] over ranks (no-op for one rank)."""
        if self.world == 1 or b <= a:
            return
        w = dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(w)
        self.pending_ranges.append((w, a, b))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        self.pending_ranges = []

    def allreduce_flag(self, word):
        """MAXIMUM of a one-element int32 device tensor over the ranks, in place (the 'this step is invalid' word of the persistent
        recurrences: a rank that must skip its update makes every rank skip it, so the replicas cannot drift apart).  The word
        stays raised until the host looks at it -- once per epoch -- and is reduced again at every step in between: the maximum is
        idempotent and keeps the error's code (a SUM would multiply a raised word by the number of ranks per step and wrap an
        int32 to zero after 32 / log2(world) steps, upon which the optimiser would silently resume)."""
        if self.world > 1:
            self._flag = dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
            self.pending.append(self._flag)

    def wait_flag(self):
        w, self._flag = getattr(self, '_flag', None), None
        if w is not None:
            w.wait()

    @property
    def grad_scale(self):
        """1 when every rank normalised its loss by the GLOBAL token counts (the sum of the ranks' gradients is the
        global gradient); 1/world when each rank averaged over its own shard (rank-average of per-rank means: exact
        only for equal token counts)."""
        return 1.0 if self.sum_of_global_means else 1.0 / self.world

    # small host-side exchanges (assessment: token ids, counts)
    def allreduce_numpy(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.world > 1:
            if dist.get_backend(self.group) == 'nccl':
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def broadcast_(self, tensors, src=0):
        broadcast_flat(tensors, src, self.group)

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


class _Ticket:
    def __init__(self, sync, t):
        self.sync, self.t = sync, t

    def wait(self):
        """The CURRENT stream waits for this collective (nothing blocks on the host)."""
        self.sync.lib.e2t_comm_wait(self.sync.comm, self.t, torch.cuda.current_stream().cuda_stream)


class RcclSync:
    """RCCL called directly through the C ABI (include/ecog2txt_hip.h, e2t_comm_*): the exchange step of the
    data-parallel path without torch.distributed."""

    def __init__(self, flat_grad, rank, world, unique_id, device, sum_of_global_means=True):
        from .hip_lib import lib
        self.lib = lib
        self.g = flat_grad
        self.rank, self.world = rank, world
        self.sum_of_global_means = sum_of_global_means
        comm = C.c_void_p()
        # the CUs left to RCCL's channel kernels: the persistent recurrences of the 256-electrode configuration occupy 200
        # (forward) / 224 (BPTT) of the 256 CUs for a whole layer sweep, one workgroup each; the exchange gets the other 32
        # (RCCL reads the variable when the communicator is made; a value the user set wins).  HSA_ENABLE_IPC_MODE_LEGACY=0 --
        # the host driver only supports dmabuf IPC -- is defaulted at package import: the HSA runtime reads it when it
        # initialises, long before a communicator is made.
        os.environ.setdefault('NCCL_MAX_NCHANNELS', '32')
        lib.e2t_comm_init(C.byref(comm), rank, world, unique_id, int(device))
        self.comm = comm
        self.pending_ranges = []
        self._flag_pending = False
        self.capturable = True             # collectives may be recorded into a hipGraph (csrc/comm.hip)

    @staticmethod
    def unique_id():
        from .hip_lib import lib, COMM_ID_BYTES
        buf = C.create_string_buffer(COMM_ID_BYTES)
        lib.e2t_comm_unique_id(buf)
        return buf.raw

    def allreduce_range(self, a, b):
        if b <= a:
            return
        t = C.c_int(-1)
        self.lib.e2t_comm_allreduce_f32(self.comm, self.g.data_ptr() + 4 * a, b - a, torch.cuda.current_stream().cuda_stream,
                                        C.byref(t))
        self.pending_ranges.append((_Ticket(self, t.value), a, b))

    def wait(self):
        if self.pending_ranges or self._flag_pending:
            self.lib.e2t_comm_wait(self.comm, -1, torch.cuda.current_stream().cuda_stream)
        self.pending_ranges = []
        self._flag_pending = False

    def warm_up(self, sizes):
        """One eager all-reduce per message size the captured step will issue (fp32 ranges of `sizes` elements, the int32 flag
        word), on scratch memory, and a device synchronisation: RCCL sets a collective's channels, protocols and peer connections
        up when it first runs one of that shape -- allocations and IPC exchanges that must not happen while a stream is being
        captured.  Called once per communicator before the first capture."""
        if getattr(self, '_warm', None) is None:
            self._warm = set()
        todo = sorted(set(int(n) for n in sizes if n > 0) - self._warm)
        if not todo and 'flag' in self._warm:
            return
        st = torch.cuda.current_stream().cuda_stream
        scratch = torch.zeros(max(todo + [1]), dtype=torch.float32, device=self.g.device)
        for n in todo:
            self.lib.e2t_comm_allreduce_f32(self.comm, scratch.data_ptr(), n, st, None)
            self._warm.add(n)
        flag = torch.zeros(1, dtype=torch.int32, device=self.g.device)
        self.lib.e2t_comm_allreduce_max_i32(self.comm, flag.data_ptr(), 1, st, None)
        self._warm.add('flag')
        self.lib.e2t_comm_wait(self.comm, -1, st)
        torch.cuda.current_stream().synchronize()

    def attach(self):
        """The communicator's stream is ordered behind the current stream's work so far and issues nothing: a captured step
        calls this on its main stream before anything else, so that the communicator's stream enters the capture from the
        capture's origin stream (a stream pulled in by a side stream crashes hipGraphInstantiate of ROCm 7.0)."""
        self.lib.e2t_comm_order_after(self.comm, torch.cuda.current_stream().cuda_stream)

    def join(self):
        """The current stream waits for every collective issued so far (legal inside a stream capture: the communicator's
        stream, which joined the capture through the collectives' event edges, is joined back)."""
        self.lib.e2t_comm_wait(self.comm, -1, torch.cuda.current_stream().cuda_stream)

    def allreduce_flag(self, word):
        """MAXIMUM of a one-element int32 device tensor over the ranks, in place, ordered behind the current stream's work so far
        (the 'this step is invalid' word of the persistent recurrences: a rank that must skip its update makes every rank skip
        it; idempotent from step to step and code-preserving, see GradSync.allreduce_flag)."""
        t = C.c_int(-1)
        self.lib.e2t_comm_allreduce_max_i32(self.comm, word.data_ptr(), 1, torch.cuda.current_stream().cuda_stream, C.byref(t))
        self._flag_pending = _Ticket(self, t.value)

    def wait_flag(self):
        """The current stream waits for the last allreduce_flag (and, collectives being ordered, everything issued before it)."""
        if self._flag_pending:
            self._flag_pending.wait()

    @property
    def grad_scale(self):
        return 1.0 if self.sum_of_global_means else 1.0 / self.world

    def allreduce_numpy(self, arr):
        """Sum a small host array over ranks (int32 / float32 on the wire; counts and token ids fit)."""
        a = np.ascontiguousarray(arr)
        st = torch.cuda.current_stream().cuda_stream
        if a.dtype.kind in 'iub':
            t = torch.from_numpy(a.astype(np.int32)).cuda()
            self.lib.e2t_comm_allreduce_i32(self.comm, t.data_ptr(), t.numel(), st, None)
        else:
            t = torch.from_numpy(a.astype(np.float32)).cuda()
            self.lib.e2t_comm_allreduce_f32(self.comm, t.data_ptr(), t.numel(), st, None)
        self.lib.e2t_comm_wait(self.comm, -1, st)
        torch.cuda.current_stream().synchronize()
        return t.cpu().numpy().astype(a.dtype)

    def broadcast_(self, tensors, src=0):
        st = torch.cuda.current_stream().cuda_stream
        for t in tensors:
            self.lib.e2t_comm_broadcast(self.comm, t.data_ptr(), t.numel() * t.element_size(), src, st, None)
        self.lib.e2t_comm_wait(self.comm, -1, st)

    def barrier(self):
        self.allreduce_numpy(np.zeros(1, np.int32))

    def close(self):
        if self.comm is not None:
            self.lib.e2t_comm_destroy(self.comm)
            self.comm = None


_BOOTSTRAPS = 0
_OWN_STORE = {}


def share_from_rank0(payload, rank, world, group=None, port_offset=1, key='e2t_rccl_uid'):
    """Hand rank 0's bytes to every rank -- bootstrap only.  Through an existing torch.distributed group if there is one;
    else through the launcher's own store when torch.distributed.run hosts one on MASTER_PORT (the workers connect as
    clients: no second port to bind, nothing left in TIME_WAIT for the next launch); else through a TCPStore that rank 0
    hosts at MASTER_ADDR:(MASTER_PORT + port_offset)."""
    if world == 1:
        return payload
    if dist.is_available() and dist.is_initialized():
        box = [payload]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    import datetime
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    timeout = datetime.timedelta(seconds=300)
    # every bootstrap of a job gets its own key: the ranks call this function the same number of times (once per communicator,
    # and SequenceNetwork makes a new one whenever the subject set changes), so a per-process counter agrees across ranks -- with
    # one key for all of them a worker that reaches `get` before rank 0's `set` would read the PREVIOUS unique id
    global _BOOTSTRAPS
    _BOOTSTRAPS += 1
    if os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True':
        store = dist.TCPStore(addr, port, world, is_master=False, timeout=timeout, wait_for_workers=False)
        key = '%s/%s/%s/%d' % (key, os.environ.get('TORCHELASTIC_RUN_ID', ''), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), _BOOTSTRAPS)
    else:
        store = _OWN_STORE.get((addr, port + port_offset))
        if store is None:
            # kept for the life of the process: rank 0 hosts the server, and a slow worker's `get` must still find it
            store = dist.TCPStore(addr, port + port_offset, world, is_master=(rank == 0), timeout=timeout)
            _OWN_STORE[(addr, port + port_offset)] = store
        key = '%s/%d' % (key, _BOOTSTRAPS)
    if rank == 0:
        store.set(key, payload)
    return store.get(key)


def bootstrap_unique_id(rank, world, group=None, port_offset=1):
    """Rank 0 makes the RCCL unique id; everybody gets it (share_from_rank0)."""
    uid = RcclSync.unique_id() if rank == 0 else None
    return share_from_rank0(uid, rank, world, group, port_offset)


def make_sync(flat_grad, group=None, sum_of_global_means=True):
    """The gradient exchange for the current process layout, or None for a single process: RcclSync when the gradients
    live on a GPU (rank / world from torch.distributed if initialised, else from RANK / WORLD_SIZE), GradSync else."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return None
    if flat_grad.is_cuda and os.environ.get('E2T_COMM', 'rccl') == 'rccl':
        uid = bootstrap_unique_id(rank, world, group)
        return RcclSync(flat_grad, rank, world, uid, flat_grad.device.index or 0, sum_of_global_means)
    return GradSync(flat_grad, group, sum_of_global_means)


def broadcast_flat(tensors, src=0, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
