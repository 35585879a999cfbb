"""Clip sharding across GPUs: one process per GPU, contiguous index ranges,
no data-path collective; one all_gather of the [N/world, heads] result rows at the end (RCCL when
the process group's backend is "nccl", gloo in the CPU tests).  Replaces nn.DataParallel
(reference NISQA_model.py:56-57)."""
import numpy as np
import torch


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def world():
    if _dist_on():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced [lo, hi) of rank's share of n items (first n % world ranks get one extra)."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_range(n):
    r, w = world()
    return shard_bounds(n, r, w)


def balanced_bounds(weights, world_size):
    """Contiguous [lo, hi) per rank with (nearly) equal WORK: boundary k sits where the prefix sum of ``weights`` (segments
    per clip from the header probe) crosses k / world of the total, rounded to the nearer side.  Equal clip COUNTS give the
    last rank of a length-sorted list of 3-30 s clips several times the first rank's work (SURVEY.md 8e asks for contiguous
    index ranges; it does not ask for equal counts).  Every rank computes the same list from the same weights."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), 0.0)
    n, ws = len(w), int(world_size)
    if n == 0 or w.sum() <= 0:
        return [shard_bounds(n, r, ws) for r in range(ws)]
    pre = np.concatenate(([0.0], np.cumsum(w)))
    cuts = [0]
    for k in range(1, ws):
        t = pre[-1] * k / ws
        j = int(np.searchsorted(pre, t, side='left'))              # first prefix >= target
        if j > 0 and t - pre[j - 1] < pre[min(j, n)] - t:
            j -= 1
        cuts.append(min(max(j, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(ws)]


def all_reduce_sum_i64(a):
    """Sum of a host int64 array over the ranks (the header-probe exchange of the work-balanced shards: every rank fills
    its share of a zero vector).  Device staging under RCCL, host tensors under gloo."""
    import torch.distributed as dist
    r, w = world()
    if w == 1:
        return a
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t)
    return t.cpu().numpy()


_RAISABLE = {'ValueError': ValueError, 'RuntimeError': RuntimeError, 'NotImplementedError': NotImplementedError,
             'FileNotFoundError': FileNotFoundError, 'OSError': OSError, 'KeyError': KeyError, 'TypeError': TypeError,
             'MemoryError': MemoryError}


def raise_together(err):
    """Fail together: every rank calls this at the same point with the exception it caught on its shard (or None).
    One all_reduce(MIN) names the first rank that failed, one broadcast carries its exception (type name + message), and
    EVERY rank raises it -- the rank that caught it raises the original object.  The reference is one process and fails at
    once with ValueError('Could not load file ...') / ('Sample too short ...') / ('n_wins ... > max_length ...')
    (NISQA_lib.py:2305-2306, 2259-2263, 2276-2277); without this exchange the rank that raised would leave _predict while
    the others block in the closing all_gather until the RCCL watchdog fires.  No-op (plain raise) without a process group."""
    r, w = world()
    if w == 1:
        if err is not None:
            raise err
        return
    import torch.distributed as dist
    nccl = dist.get_backend() == 'nccl'
    tdev = torch.device('cuda', torch.cuda.current_device()) if nccl else torch.device('cpu')
    first = torch.tensor([r if err is not None else w], dtype=torch.int64, device=tdev)
    dist.all_reduce(first, op=dist.ReduceOp.MIN)
    first = int(first.item())
    if first >= w:
        return
    payload = b''
    if r == first:
        payload = (type(err).__name__ + '\0' + (str(err.args[0]) if len(err.args) == 1 else str(err))).encode('utf-8', 'replace')[:1 << 16]
    size = torch.tensor([len(payload)], dtype=torch.int64, device=tdev)
    dist.broadcast(size, first)
    buf = torch.zeros(int(size.item()), dtype=torch.uint8, device=tdev)
    if r == first:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, first)
    if r == first:
        raise err
    name, _, msg = bytes(buf.cpu().numpy().tobytes()).decode('utf-8', 'replace').partition('\0')
    raise _RAISABLE.get(name, RuntimeError)(msg if name in _RAISABLE else '{}: {}'.format(name, msg))


def gather_rows(local, n, lo, hi, dev, bounds=None):
    """All ranks get the full [n, C] array assembled from every rank's [hi-lo, C] rows.  bounds: the [lo, hi) of every
    rank when the shards are not shard_bounds(n, k, world) (work-balanced shards)."""
    r, w = world()
    if w == 1:
        return local
    if bounds is None:
        bounds = [shard_bounds(n, k, w) for k in range(w)]
    C = local.shape[1]
    cap = max(1, max(b - a for a, b in bounds))         # rows per rank, padded to the largest shard
    backend = torch.distributed.get_backend()
    tdev = torch.device(dev) if backend == 'nccl' else torch.device('cpu')
    buf = torch.zeros((cap, C), dtype=torch.float32, device=tdev)
    buf[:hi - lo] = torch.from_numpy(local).to(tdev)
    parts = [torch.empty_like(buf) for _ in range(w)]
    torch.distributed.all_gather(parts, buf)
    out = np.zeros((n, C), dtype=np.float32)
    for k in range(w):
        a, b = bounds[k]
        out[a:b] = parts[k][:b - a].cpu().numpy()
    return out


def all_reduce_sum_(t):
    """In-place sum of a tensor over the ranks (no-op without a process group).  RCCL takes the device tensor as it
    is; under gloo (CPU tests, shared-GPU tests) the tensor is staged through host memory."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    nccl = dist.get_backend() == 'nccl'
    if nccl == t.is_cuda:                                   # RCCL takes device tensors, gloo host tensors
        dist.all_reduce(t)
    else:                                                   # a host tensor under RCCL (label counts), a device tensor under gloo
        c = t.cuda() if nccl else t.cpu()
        dist.all_reduce(c)
        t.copy_(c)
    return t


def broadcast_(t, src=0):
    """In-place broadcast from rank ``src`` (same staging rule as all_reduce_sum_)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    nccl = dist.get_backend() == 'nccl'
    if nccl == t.is_cuda:
        dist.broadcast(t, src)
    else:
        c = t.cuda() if nccl else t.cpu()
        dist.broadcast(c, src)
        t.copy_(c)
    return t
