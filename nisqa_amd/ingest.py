"""Host ingest of the predict loop: WAV files -> page-locked staging -> one H2D copy per batch.

The kernels take ~1 ms for a batch of 64 ten-second clips, i.e. 60 GB/s of PCM16; a loader that builds a Python
``bytes`` per file, slices it, concatenates the batch and pins the result (five copies per sample) feeds them at
1.7 k clips/s, and Python threads doing the reads themselves stop at ~8 k (interpreter lock).  Here a sample is
copied ONCE on the host and no Python runs per file: libnisqa_ingest.so (csrc/ingest.cpp, include/nisqa_ingest.h)
parses the RIFF headers of the whole batch on a native thread pool, the batch layout is fixed from the headers
alone, and the same pool then ``pread``-s every data chunk straight into its slice of a persistent page-locked
buffer (page cache -> pinned memory).  Three such buffers rotate: one being filled by the producer thread, one in flight over PCIe, one spare;
a slot is recycled only after the HIP event recorded behind its H2D copy has completed.

Files that are not mono PCM16 (stereo, 8/24/32-bit, float) are decoded by ``wavio.read_wav`` with the reference's
``lb.load`` semantics and staged as float32 through the same buffers.  FLAC files (lb.load reads them through the same
soundfile call) are decoded by the native reader threads: mono 16-bit streams straight into their int16 slot, others via ``wavio``.
"""
import ctypes
import os
import queue
import threading
import time

import numpy as np
import torch

from . import lib as _lib
from . import wavio

_ALIGN = 64
_noted = set()


def _note_once(msg):
    if msg not in _noted:
        _noted.add(msg)
        print(msg)


def cpu_budget():
    """CPUs this process may actually burn: the affinity mask, cut down to the cgroup's CPU quota (cpu.max of cgroup v2,
    cpu.cfs_quota_us of v1).  A container with 256 visible cores and a 16-CPU quota that starts 32 reader threads spends
    its quota in half of every 100 ms period and is then frozen for the other half -- consumer thread included."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def gpu_local_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function), cut to this process's affinity
    mask; empty set when unknown.  Reader threads that copy page-cache pages into the page-locked staging buffer run
    1.5x faster on the GPU's own node (2.2 against 3.4 ms per 245 MB batch), and with eight ranks on a two-socket host
    it keeps every rank's staging traffic off the inter-socket links."""
    try:
        p = torch.cuda.get_device_properties(device)
        path = '/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open(path) as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(','):
            if '-' in part:
                a, b = part.split('-')
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        return cpus & set(os.sched_getaffinity(0))
    except Exception:
        return set()


def probe_headers(ds, indices, num_workers=0):
    """RIFF header records (numpy structured array of lib.WavInfo: status, n_frames, sample_rate, ...) of the items
    ``indices`` of a SpeechQualityDataset, native threads, headers only.  Nothing is raised for an unreadable file: its
    ``status`` says so (the work-balanced shards of the predict loop only need lengths; the reference's ValueError is
    raised where the reference raises it, when the file is loaded)."""
    idx = list(indices)
    n = len(idx)
    col, d = ds.df[ds.filename_column].tolist(), ds.data_dir
    enc = [os.fsencode(os.path.join(d, col[i])) for i in idx]
    paths = (ctypes.c_char_p * n)(*enc)
    infos = (_lib.WavInfo * n)()
    asked = int(num_workers or 0)
    try:
        local_world = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
    except ValueError:
        local_world = 1
    budget = max(1, cpu_budget() // local_world)
    workers = max(1, min(asked if asked > 0 else budget, budget))
    _lib.load_ingest().nisqa_ingest_probe(paths, n, infos, workers)
    return np.ctypeslib.as_array(infos).copy()


def _verbatim_i16(info):
    """Which files of a header array reach the staging buffer as int16: mono PCM16 data chunks (copied verbatim) and mono 16-bit
    FLAC streams (decoded into the slot by the reader threads, csrc/flac.hpp)."""
    return ((info['tag'] == _lib.WAV_TAG_PCM) | (info['tag'] == _lib.WAV_TAG_FLAC)) & (info['bits'] == 16) & (info['channels'] == 1)


class StagingRing(object):
    """``n_slots`` page-locked byte buffers, grown on demand.  A slot handed to the consumer comes back with
    ``release_after(slot, event)``; ``acquire`` blocks until then and until ``event`` (the HIP event recorded
    behind the slot's H2D copies) has completed."""

    def __init__(self, n_slots, pin):
        self.pin = pin
        self.buf = [None] * n_slots
        self.busy = [None] * n_slots
        self.back = [threading.Event() for _ in range(n_slots)]
        for e in self.back:
            e.set()
        self.k = 0

    def acquire(self, nbytes):
        k = self.k
        self.k = (k + 1) % len(self.buf)
        self.back[k].wait()
        if self.busy[k] is not None:
            self.busy[k].synchronize()
            self.busy[k] = None
        if self.buf[k] is None or self.buf[k].numel() < nbytes:
            self.buf[k] = None
            self.buf[k] = torch.empty(nbytes + nbytes // 4 + 4096, dtype=torch.uint8, pin_memory=self.pin)
        self.back[k].clear()
        return k

    def release_after(self, k, event):
        self.busy[k] = event
        self.back[k].set()

    def abandon(self):
        for e in self.back:
            e.set()

    def reset(self):
        """Ready for a new loop: every slot free (after the copies still reading from it), buffers kept."""
        for k in range(len(self.buf)):
            if self.busy[k] is not None:
                self.busy[k].synchronize()
                self.busy[k] = None
            self.back[k].set()
        self.k = 0
        return self


# Page-locking a few hundred MB costs tens of milliseconds per buffer: the rings of finished loops are kept for the next
# one (one predict() call per process is the common case, repeated calls and the train/validate alternation the other).
_spare_rings = {}
_spare_lock = threading.Lock()


def _take_ring(n_slots, pin):
    with _spare_lock:
        lst = _spare_rings.get((n_slots, pin))
        if lst:
            return lst.pop().reset()
    return StagingRing(n_slots, pin)


def _give_ring(ring):
    with _spare_lock:
        lst = _spare_rings.setdefault((len(ring.buf), ring.pin), [])
        if len(lst) < 2:
            lst.append(ring)


class Group(object):
    """Clips of one sample rate inside a staged batch."""
    __slots__ = ('ids', 'lengths', 'sr', 'offset', 'nbytes', 'is_i16', 'names')

    def __init__(self, ids, lengths, sr, offset, nbytes, is_i16, names=None):
        self.ids, self.lengths, self.sr, self.offset, self.nbytes, self.is_i16 = ids, lengths, sr, offset, nbytes, is_i16
        self.names = names


class Staged(object):
    __slots__ = ('slot', 'groups')

    def __init__(self, slot, groups):
        self.slot, self.groups = slot, groups


class LengthAware(object):
    """Batching policy of the predict loop: the producer probes the RIFF headers of a
    window of items (headers only, native threads), sorts the window by (sample rate, length) and cuts batches by WORK,
    not by a clip count -- a batch is closed when it holds at least ``bs`` clips AND at least ``min_clips`` clips AND at
    least ``min_tokens`` segments, or when the next clip would push its staged bytes over ``byte_cap``.

    ``bs`` (the reference's --bs / tr_bs_val) is therefore a LOWER bound: the reference's default ``--bs 1`` coalesces
    into full launches (results do not depend on the batch composition: eval-mode BatchNorm, per-clip masks; tested), and
    clips of similar length share a batch, so a launch of the BiLSTM (as long as its longest clip) or of the attention
    kernels carries no short clips waiting for a long one.  The consumer scatters result rows back by item index, so the
    output order is the input order.

    tokens_of(n_frames[int64 array], sample_rate[int array]) -> segments per clip (host arithmetic on header fields)."""

    def __init__(self, indices, bs, tokens_of, min_tokens=0, min_clips=1, byte_cap=256 << 20, window=16384):
        self.indices = list(indices)
        self.bs, self.tokens_of = max(1, int(bs)), tokens_of
        self.min_tokens, self.min_clips = int(min_tokens), max(1, int(min_clips))
        self.byte_cap, self.window = int(byte_cap), max(1, int(window))

    def __len__(self):
        return len(self.indices)

    @staticmethod
    def staged_bytes(frames, srs, widths, pos):
        """Bytes Ingest._stage lays out for the clips ``pos``: clips of one sample rate form a group, and a group is staged
        as int16 only if ALL its clips are mono PCM16 (width 2) -- one stereo / 24-bit / float / G.711 file widens its whole
        group to float32."""
        pos = np.asarray(pos, dtype=np.int64)
        total = 0
        for sr in np.unique(srs[pos]):
            sel = pos[srs[pos] == sr]
            total += int(frames[sel].sum()) * (2 if (widths[sel] == 2).all() else 4)
        return total

    def cut(self, frames, srs, widths):
        """Batches (lists of POSITIONS into the window) for clips with the given header fields.  Within a sample rate the
        clips that stage as int16 (mono PCM16, width 2) come before the ones that need the host decoder (width 4), so a
        batch mixes the two only at the seam; the byte cap is charged with what _stage really lays out (staged_bytes: a
        float clip in a rate group makes every clip of that group 4 bytes per sample).  ``bs`` / ``min_clips`` /
        ``min_tokens`` are lower bounds, ``byte_cap`` is the memory knob (page-locked host memory and HBM per batch in
        flight): --bs does not bound memory."""
        n = len(frames)
        if n == 0:
            return []
        tokens = np.maximum(1, np.asarray(self.tokens_of(frames, srs), dtype=np.int64))
        order = np.lexsort((np.arange(n), frames, widths, srs))        # by rate, then int16-before-float, then length, then input order
        out, cur, ct = [], [], 0
        grp, cost = {}, 0                                              # rate -> [int16 frames, float frames] of the open batch; its staged bytes
        need = max(self.bs, self.min_clips)
        cap, min_tokens = self.byte_cap, self.min_tokens
        fr_l, sr_l, w_l, tk_l = frames.tolist(), srs.tolist(), widths.tolist(), tokens.tolist()   # (plain ints: this loop runs per item)

        def rate_cost(a, b):
            return 4 * (a + b) if b else 2 * a

        for k in order.tolist():
            sr, f = sr_l[k], fr_l[k]
            slow = w_l[k] != 2
            a, b = grp.get(sr, (0, 0))
            a2, b2 = (a, b + f) if slow else (a + f, b)
            add = rate_cost(a2, b2) - rate_cost(a, b)
            if cur and (cost + add > cap or sr != sr_l[cur[-1]] and len(cur) >= need):
                out.append(cur)
                cur, ct, grp, cost = [], 0, {}, 0
                a2, b2 = (0, f) if slow else (f, 0)
                add = rate_cost(a2, b2)
            cur.append(k)
            grp[sr] = (a2, b2)
            cost += add
            ct += tk_l[k]
            if len(cur) >= need and ct >= min_tokens:
                out.append(cur)
                cur, ct, grp, cost = [], 0, {}, 0
        if cur:
            out.append(cur)
        # a small remainder does not get a launch chain of its own when the batch before it can take it (the byte cap is a
        # staging-buffer size, soft by half): a BiLSTM launch over 7 long clips lasts as long as one over 128 of them
        if len(out) >= 2 and len(out[-1]) * 2 < need and srs[out[-1][0]] == srs[out[-2][-1]] \
                and self.staged_bytes(frames, srs, widths, out[-2] + out[-1]) <= self.byte_cap + self.byte_cap // 2:
            tail = out.pop()
            out[-1] = out[-1] + tail
        return out


class Ingest(object):
    """Iterate over staged batches of ``ds`` (a SpeechQualityDataset): ``for staged in Ingest(...)``; the caller
    turns ``staged.groups`` into H2D copies out of ``ring.buf[staged.slot]`` and reports the event behind them with
    ``ring.release_after``.  Batches are prepared ``depth`` ahead on a producer thread."""

    def __init__(self, ds, batches, pin, num_workers, depth=2, device=None):
        """``batches``: a list of index lists (staged exactly as given), or a LengthAware policy (the producer forms the
        batches itself from the headers)."""
        self.ds, self.batches = ds, batches
        self.device = device
        self.ring = _take_ring(depth + 1, pin)
        # reader threads: what the caller asked for, but never more than the CPU budget leaves next to the producer and
        # consumer threads (over-subscribing a quota-limited container stalls the whole loop, see cpu_budget), and -- when
        # the producer pins itself and its native pool to the GPU's NUMA node -- never more than that node offers
        budget = cpu_budget()
        # one process per GPU (torchrun): the ranks of a node share its CPU quota
        try:
            local_world = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
        except ValueError:
            local_world = 1
        if local_world > 1:
            budget = max(2, budget // local_world)
        self.numa_cpus = set()
        if device is not None and os.environ.get('NISQA_INGEST_NUMA', '1') != '0':
            self.numa_cpus = gpu_local_cpus(device)
            if self.numa_cpus:
                budget = min(budget, len(self.numa_cpus))
        asked = int(num_workers or 0)
        if asked <= 0 and isinstance(batches, LengthAware):
            asked = budget                      # the reference's default (--num_workers 0): as many readers as the box allows
        # next to the readers run the producer and the consumer thread (mostly blocked in native calls / on events)
        self.workers = max(1, min(asked, budget - (3 if local_world == 1 else 1)))
        if asked > self.workers:
            _note_once('nisqa_amd.ingest: %d reader threads instead of the %d requested (CPU budget of this process: %d)'
                       % (self.workers, asked, budget))
        self.lib = _lib.load_ingest()
        # the filename column is read ONCE per loop (a pandas scalar lookup per item costs more host time than staging the
        # item's samples); not cached on the dataset: an in-place edit of the column between two calls must be seen
        self._col = ds.df[ds.filename_column].tolist() if hasattr(ds, 'df') and hasattr(ds, 'filename_column') else None
        self.q = queue.Queue(maxsize=depth)
        self.stop = threading.Event()
        self.stats = {'paths': 0.0, 'probe': 0.0, 'layout': 0.0, 'slot_wait': 0.0, 'read': 0.0, 'queue_wait': 0.0, 'batches': 0}
        self._helper = None                        # header-probe helper of _planned (joined in close)
        self.thread = threading.Thread(target=self._produce, name='nisqa-ingest', daemon=True)
        self.thread.start()

    # -- producer side ---------------------------------------------------------------------------------
    def _probe(self, idx, stats=None, chunk=None):
        """RIFF headers of items ``idx`` -> (names, encoded paths, header records as a numpy structured array).  stats: the
        dict the 'paths' / 'probe' seconds are added to (the helper thread of _planned passes its own: two threads never
        write self.stats).  chunk: headers per native call.  The native pool runs ONE job at a time (csrc/ingest.cpp: Pool::run), so a
        probe of 16 384 headers in one call keeps the readers that stage the batches in hand out of the pool for its whole 20 ms --
        the copy stream ran dry for 5-15 ms at every window boundary (rocprofv3 --memory-copy-trace); in chunks of 1 024 the reads
        wait 1.3 ms at most."""
        ds, L, n = self.ds, self.lib, len(idx)
        T, t0 = self.stats if stats is None else stats, time.perf_counter()
        if self._col is not None:                                  # the column as it was when this loop started
            col, d = self._col, ds.data_dir
            names = [os.path.join(d, col[i]) for i in idx]
        else:
            names = [ds.file_path(i) for i in idx]
        enc = [os.fsencode(p) for p in names]
        paths = (ctypes.c_char_p * n)(*enc)
        infos = (_lib.WavInfo * n)()
        t1 = time.perf_counter()
        T['paths'] += t1 - t0
        if chunk is None or n <= chunk:
            rc = L.nisqa_ingest_probe(paths, n, infos, self.workers)
        else:
            rc = 0
            for s in range(0, n, chunk):
                m = min(chunk, n - s)
                rc += L.nisqa_ingest_probe(ctypes.cast(ctypes.byref(paths, s * ctypes.sizeof(ctypes.c_char_p)), ctypes.POINTER(ctypes.c_char_p)),
                                           m, ctypes.cast(ctypes.byref(infos, s * ctypes.sizeof(_lib.WavInfo)), ctypes.POINTER(_lib.WavInfo)),
                                           self.workers)
        T['probe'] += time.perf_counter() - t1
        if rc:
            bad = next(k for k in range(n) if infos[k].status != _lib.WAV_OK)
            raise ValueError('Could not load file {}'.format(names[bad]))      # NISQA_lib.py:2305-2306
        return names, enc, np.ctypeslib.as_array(infos).copy()

    def _stage(self, idx, probed=None):
        """Stage one batch: layout from the headers, then every data chunk straight into a page-locked slot."""
        ds, L, n = self.ds, self.lib, len(idx)
        T = self.stats
        names, enc, info = probed if probed is not None else self._probe(idx)
        t2 = time.perf_counter()
        paths = (ctypes.c_char_p * n)(*enc)
        info = np.ascontiguousarray(info)
        infos = ctypes.cast(info.ctypes.data, ctypes.POINTER(_lib.WavInfo))
        frames, srs = info['n_frames'], info['sample_rate']
        fast = _verbatim_i16(info)
        # batch layout from the headers alone: clips of one rate are contiguous, int16 if ALL of them are mono PCM16
        layout, total = [], 0
        dst_off = np.full(n, -1, dtype=np.int64)
        for sr in dict.fromkeys(srs.tolist()):
            sel = np.flatnonzero(srs == sr)
            is_i16 = bool(fast[sel].all())
            width = 2 if is_i16 else 4
            off = total + np.concatenate(([0], np.cumsum(frames[sel][:-1]))) * width
            nbytes = int(frames[sel].sum()) * width
            if is_i16:
                dst_off[sel] = off
            layout.append((int(sr), sel, is_i16, off, nbytes))
            total = (total + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        t3 = time.perf_counter()
        T['layout'] += t3 - t2
        slot = self.ring.acquire(max(total, _ALIGN))
        buf = self.ring.buf[slot]
        t4 = time.perf_counter()
        T['slot_wait'] += t4 - t3
        rc = L.nisqa_ingest_read(paths, n, infos, ctypes.c_void_p(buf.data_ptr()),
                                 dst_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), self.workers)
        T['read'] += time.perf_counter() - t4
        T['batches'] += 1
        if rc:
            bad = next(k for k in range(n) if info['status'][k] != _lib.WAV_OK)
            self.ring.release_after(slot, None)
            raise ValueError('Could not load file {}'.format(names[bad]))
        groups = []
        raw = None
        for sr, sel, is_i16, off, nbytes in layout:
            if not is_i16:                                          # stereo / 8, 24, 32-bit / float: host decode
                raw = buf.numpy() if raw is None else raw
                for k, o in zip(sel.tolist(), off.tolist()):
                    y, _ = wavio.read_wav(names[k], ds.ms_channel)
                    if y.dtype == np.int16:
                        y = y.astype(np.float32) / np.float32(32768.0)
                    raw[o:o + 4 * len(y)].view(np.float32)[:] = y
            groups.append(Group([idx[k] for k in sel.tolist()], frames[sel].tolist(), sr, int(off[0]) if len(off) else 0,
                                nbytes, is_i16, [names[k] for k in sel.tolist()]))
        return Staged(slot, groups)

    def _planned(self):
        """Batches of a LengthAware policy: yields (idx, probed) window by window; the headers of window w + 1 are
        probed on a helper thread while the batches of window w are staged."""
        pol = self.batches
        # the FIRST window is small: its probe and cut run before anything is staged (25-35 ms for 16 384 headers -- 1.5 % of a
        # 100 000-row job during which the link carries nothing); every later window is prepared under the one before it
        first = min(pol.window, 2048)
        starts = [0] + list(range(first, len(pol.indices), pol.window)) if len(pol.indices) > first else [0]
        wins = [pol.indices[s:e] for s, e in zip(starts, starts[1:] + [len(pol.indices)])] if len(pol.indices) else []
        box = {}

        def probe_into(w):
            # runs on the helper thread: timings stay local and are merged by the producer; a loop that was closed
            # (self.stop) probes nothing more
            loc = {'paths': 0.0, 'probe': 0.0}
            try:
                if self.stop.is_set():
                    box[w] = ('stop', None, loc)
                    return
                names, enc, info = self._probe(wins[w], loc, chunk=None if w == 0 else 1024)    # (w >= 1: in the background, under window w - 1's staging)
                # the window's batches are cut HERE, on the helper thread: sorting and walking 16 384 items in Python is
                # 10-20 ms during which the producer staged nothing and the link ran dry once per window
                fast = _verbatim_i16(info)
                cuts = pol.cut(info['n_frames'].astype(np.int64), info['sample_rate'].astype(np.int64), np.where(fast, 2, 4))
                box[w] = ('ok', (names, enc, info, cuts), loc)
            except BaseException as e:
                box[w] = ('err', e, loc)

        helper = None
        if wins:
            probe_into(0)
        for w in range(len(wins)):
            if helper is not None:
                helper.join()
                self._helper = None
            kind, val, loc = box.pop(w)
            for k_, v_ in loc.items():
                self.stats[k_] += v_
            if kind == 'stop':
                return
            if kind == 'err':
                raise val
            if w + 1 < len(wins):
                helper = self._helper = threading.Thread(target=probe_into, args=(w + 1,), name='nisqa-probe', daemon=True)
                helper.start()
            else:
                helper = None
            names, enc, info, cuts = val
            for pos in cuts:
                yield [wins[w][k] for k in pos], ([names[k] for k in pos], [enc[k] for k in pos], info[pos])

    def _produce(self):
        try:
            if self.numa_cpus:
                try:                                       # this thread only; the native reader pool inherits it
                    os.sched_setaffinity(0, self.numa_cpus)
                except OSError:
                    pass
            it = self._planned() if isinstance(self.batches, LengthAware) else ((idx, None) for idx in self.batches)
            for idx, probed in it:
                if self.stop.is_set():
                    return
                st = self._stage(idx, probed)
                t0 = time.perf_counter()
                self.q.put(('ok', st))
                self.stats['queue_wait'] += time.perf_counter() - t0
            self.q.put(('end', None))
        except BaseException as e:             # surfaces in the consumer, like a DataLoader worker error
            self.q.put(('err', e))

    # -- consumer side ---------------------------------------------------------------------------------
    def __iter__(self):
        while True:
            kind, val = self.q.get()
            if kind == 'end':
                return
            if kind == 'err':
                raise val
            yield val

    def close(self):
        if self.ring is None:                      # already closed
            return
        self.stop.set()
        self.ring.abandon()
        try:
            while True:                        # unblock a producer waiting on a full queue
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=30)
        helper = self._helper
        if helper is not None:                     # a probe of the next window still running: it sees self.stop, or finishes
            helper.join(timeout=30)
        if not self.thread.is_alive():
            _give_ring(self.ring)                  # the consumer's release events are still attached: reset() waits for them
        self.ring = None
