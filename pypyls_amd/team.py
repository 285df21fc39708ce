"""
ONE ordinary call, all the GPUs of the node.

In the reference ``n_proc=8`` turns a single ``behavioral_pls(...)`` call into
eight joblib workers (pyls/utils.py:252-279, pyls/base.py:286-292, 490-507,
644-650; pyls/structures.py:162-168).  Here the workers are the GPUs: a
:class:`Team` is one process that owns one device context per GPU and one host
thread per context ("single process driving 8 devices
(ncclCommInitAll) is sufficient; no torch.distributed").  Rank r of the team
runs its shard of the permutations and bootstraps on ITS device -- the index
arrays are drawn once, by the one generator thread of the call, and shared --
and the ranks meet in ONE all-gather that a single thread issues for all of
them (``plsx_allgather_all``: one ncclGroupStart / n x ncclAllGather /
ncclGroupEnd over the communicators ``plsx_comm_init_all`` opened with
ncclCommInitAll; peer copies when a device is listed twice, which is how a
single GPU exercises this path in the tests).  Rank 0 finishes the analysis.

``torch.distributed`` is not involved; a script launched under ``torchrun``
with an initialised process group keeps the one-process-per-GPU path of
parallel.py and ignores ``n_proc`` / ``device_ids``.
"""
import ctypes
import threading

import numpy as np

TRANSPORT = {'auto': 0, 'rccl': 1, 'peer': 2}


def resolve_devices(n_proc=None, device_ids=None):
    """Device ordinals a front-end call should use, or None for the ordinary one-device call.

    ``device_ids`` (an explicit list; an ordinal may repeat: contexts then share that GPU) wins; otherwise
    ``n_proc`` -- the reference's worker count, already resolved by PLSInputs ('max' / -1 -> every CPU) -- is the
    number of GPUs wanted, capped by the GPUs visible, starting at the current device."""
    import torch
    if not torch.cuda.is_available():
        return None
    have = torch.cuda.device_count()
    if device_ids is not None:
        ids = [int(d) for d in np.atleast_1d(device_ids)]
        if not ids:
            raise ValueError('device_ids is empty')
        for d in ids:
            if d < 0 or d >= have:
                raise ValueError('device_ids: no GPU {} ({} visible)'.format(d, have))
        return ids if len(ids) > 1 else (None if ids[0] == torch.cuda.current_device() else ids)
    if n_proc is None:
        return None
    want = min(int(n_proc), have)
    if want <= 1:
        return None
    cur = torch.cuda.current_device()
    return [(cur + i) % have for i in range(want)]


class Team(object):
    """Contexts 0 .. n-1 on ``device_ids`` as the ranks of one communicator (include/plsx.h, plsx_comm_init_all)."""

    def __init__(self, device_ids, transport='auto'):
        from . import engine as _engine
        self.device_ids = [int(d) for d in device_ids]
        self.world = len(self.device_ids)
        seen = {}
        self.engines = []
        for d in self.device_ids:
            k = seen.get(d, 0)
            seen[d] = k + 1
            self.engines.append(_engine.default_engine(d, replica=k))
        self.lib = self.engines[0].lib
        self._ctxs = (ctypes.c_void_p * self.world)(*[e.ctx for e in self.engines])
        self.transport = None
        self.why = ''
        eng0 = self.engines[0]
        want = TRANSPORT[transport]
        if want != TRANSPORT['peer']:
            try:
                eng0.comm_load()
            except _engine.PlsxError as exc:            # no librccl: peer copies move the same bytes over xGMI
                if want == TRANSPORT['rccl']:
                    raise
                self.why, want = str(exc)[:200], TRANSPORT['peer']
        rc = self.lib.plsx_comm_init_all(self._ctxs, self.world, want)
        if rc != 0 and want == TRANSPORT['auto']:
            self.why = (self.lib.plsx_last_error(eng0.ctx) or b'').decode()[:200]
            rc = self.lib.plsx_comm_init_all(self._ctxs, self.world, TRANSPORT['peer'])
        eng0._check(rc)
        rccl = self.lib.plsx_comm_transport(eng0.ctx) == TRANSPORT['rccl']
        self.transport = 'rccl' if rccl else 'peer'
        if self.why:
            import warnings
            warnings.warn('team of devices {}: RCCL communicator unavailable ({}); the all-gather runs as peer '
                          'copies (hipMemcpyPeerAsync)'.format(self.device_ids, self.why))
        self.barrier = threading.Barrier(self.world)
        self._send = [None] * self.world
        self._recv = [None] * self.world
        self._streams = [None] * self.world
        self._rc = 0
        self.unrefined = 0
        self.run_lock = threading.Lock()                # one analysis at a time per team

    def collective_name(self):
        return ('plsx_allgather_all: ncclAllGather x {} in one group (ncclCommInitAll, single process)'
                if self.transport == 'rccl' else
                'plsx_allgather_all: hipMemcpyPeerAsync pulls, {} ranks (single process)').format(self.world)

    def close(self):
        for e in self.engines:
            if getattr(e, 'ctx', None):
                try:
                    e.comm_destroy()
                except Exception:                       # noqa: BLE001 -- at exit the device may be gone already
                    pass

    # ------------------------------------------------------------------
    def allgather(self, rank, flat):
        """Called by EVERY rank's thread with its packed fp64 buffer (equal lengths): returns the (world, n) tensor on
        the rank's device.  The threads meet at a barrier, rank 0 issues the one plsx_allgather_all for all of them on
        the ranks' current streams, and every rank drains its stream before any send buffer may be released."""
        import torch
        eng = self.engines[rank]
        recv = torch.empty((self.world, flat.numel()), dtype=flat.dtype, device=flat.device)
        self._send[rank], self._recv[rank] = flat, recv
        self._streams[rank] = torch.cuda.current_stream(eng.device).cuda_stream
        self.barrier.wait()
        if rank == 0:
            n = self.world
            nbytes = flat.numel() * flat.element_size()
            ok = all(t.numel() == flat.numel() and t.dtype == flat.dtype and t.is_contiguous() for t in self._send)
            if not ok:
                self._rc = 'ranks disagree about the packed buffer: {}'.format([tuple(t.shape) for t in self._send])
            else:
                vp = ctypes.c_void_p
                sends = (vp * n)(*[t.data_ptr() for t in self._send])
                recvs = (vp * n)(*[t.data_ptr() for t in self._recv])
                streams = (vp * n)(*self._streams)
                self._rc = self.lib.plsx_allgather_all(self._ctxs, n, sends, recvs, nbytes, streams)
        self.barrier.wait()
        if self._rc != 0:
            if isinstance(self._rc, str):
                raise RuntimeError(self._rc)
            self.engines[0]._check(self._rc)
        torch.cuda.synchronize(eng.device)
        self.barrier.wait()                             # every stream has passed the collective: buffers are free
        self._send[rank] = self._recv[rank] = None
        return recv

    def run(self, fn):
        """``fn(rank, world, engine)`` on one host thread per rank (rank 0 on the caller's thread); returns rank 0's
        result.  An exception on any rank breaks the barrier so that no peer waits for it, and the first one raised
        is re-raised here."""
        import torch
        errors = [None] * self.world
        out = [None] * self.world
        counts = [0] * self.world

        def work(rank):
            eng = self.engines[rank]
            try:
                torch.cuda.set_device(eng.device)       # (the current device is per host thread)
                ok = False
                with eng.lock:
                    try:
                        out[rank] = fn(rank, self.world, eng)
                        ok = True
                    finally:
                        if getattr(eng, 'ctx', None):
                            counts[rank] = eng.end_analysis(warn=ok) or 0
            except BaseException as exc:                # noqa: BLE001 -- re-raised by the caller's thread
                errors[rank] = exc
                self.barrier.abort()

        with self.run_lock:
            self.barrier.reset()
            self._rc = 0
            prev = torch.cuda.current_device()
            threads = [threading.Thread(target=work, args=(r,), name='plsx-rank{}'.format(r), daemon=True)
                       for r in range(1, self.world)]
            for t in threads:
                t.start()
            try:
                work(0)
            finally:
                for t in threads:
                    t.join()
                torch.cuda.set_device(prev)
            real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
            if real:
                raise real[0]
            for e in errors:
                if e is not None:
                    raise e
            self.unrefined = sum(counts)                    # graded resamples no rank could refine: the front-end warns
        return out[0]


_TEAMS = {}
_TEAMS_LOCK = threading.Lock()


def team_for(device_ids, transport='auto'):
    """The cached team of these devices (the communicator and the contexts stay open between calls, like the default
    engine of a single device); ``release_teams()`` closes them."""
    key = (tuple(int(d) for d in device_ids), transport)
    with _TEAMS_LOCK:
        t = _TEAMS.get(key)
        if t is None or any(not getattr(e, 'ctx', None) for e in t.engines):
            for other_key in [k for k in _TEAMS if set(k[0]) & set(key[0])]:
                _TEAMS.pop(other_key).close()           # a context belongs to one communicator at a time
            if not _TEAMS:
                import atexit
                atexit.register(release_teams)
            t = _TEAMS[key] = Team(key[0], transport)
    return t


def release_teams():
    with _TEAMS_LOCK:
        teams = list(_TEAMS.values())
        _TEAMS.clear()
    for t in teams:
        t.close()
