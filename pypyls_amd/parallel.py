"""
Multi-GPU sharding of the resampling loops: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  Resamples are independent given their index arrays -- the
reference runs them in independent joblib workers (pyls/base.py:490-507,
644-650) -- so rank r takes a contiguous slice of the permutations and a chunk-cyclic
share of the bootstraps (shard_chunks), every rank holds a full replica of X, and there is exactly ONE
collective: an all-gather of a packed per-rank buffer

    [ perm_singval slice | distrib slice | partial sum U | partial sum U^2 ]

after which every rank puts the slices back into global order and adds the
partial sums in rank order (fixed order -> deterministic).

Where the collective runs.  The rendezvous -- who the ranks are -- is the host launcher's
(``torch.distributed``: torchrun's env and store).  The DATA collective itself is the C ABI's
``plsx_allgather`` (include/plsx.h): an RCCL communicator of one rank per GPU opened by
``plsx_comm_init`` from a 128-byte unique id that rank 0 draws and the process group broadcasts
once per process -- when the launcher asked for it with :func:`open_native_comm`, an explicit
collective call right after ``init_process_group`` (bench.py makes it).  libplsx.so binds RCCL with
dlopen and is told to use the copy PyTorch-ROCm already loaded, so the process keeps ONE
communication runtime.  Without that call, or when the communicator cannot be opened on every
rank (no librccl, init failure or time-out), all ranks use ``all_gather_into_tensor`` of the
process group -- still RCCL, never the CPU -- and :func:`collective_name`, a pure query, says which
of the two runs.  ``gloo`` groups (CPU tests) never touch it.

ONE process driving several GPUs (``n_proc`` / ``device_ids`` of the front-ends) does not come
through here at all: team.py, ``plsx_comm_init_all`` / ``plsx_allgather_all``.
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:                                   # pragma: no cover
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def rank_world():
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(), d.get_world_size()


# the communicator behind plsx_allgather: one per process, on its own small context (no data is ever bound to it,
# so it costs no device memory and is independent of which Engine an analysis uses)
_NATIVE = {'state': None, 'engine': None, 'why': '', 'group': None}     # state: None = not opened, True = open, False = fall back
NATIVE_INIT_TIMEOUT_S = 120.0


def native_comm():
    """The Engine whose context holds the RCCL communicator :func:`open_native_comm` opened for the CURRENT process
    group, or None.  No side effects: it never opens anything, so it is safe on one rank only (logging)."""
    d = _dist()
    if d is None or not _NATIVE['state'] or _NATIVE['group'] is not d.group.WORLD:
        return None
    return _NATIVE['engine']


def open_native_comm(timeout_s=None):
    """Open the communicator behind ``plsx_allgather`` for the current ``nccl`` process group.  EXPLICIT and
    COLLECTIVE: call it on EVERY rank, once, right after ``init_process_group`` (bench.py does; the front-ends never
    do it behind the caller's back -- without it the data collective is the process group's own
    ``all_gather_into_tensor``, RCCL all the same).  Every rank binds librccl, rank 0 draws the unique id, the group
    broadcasts it, every rank calls ``plsx_comm_init`` (ncclCommInitRank) on a helper thread that is given
    ``timeout_s`` (default NATIVE_INIT_TIMEOUT_S); the outcome -- bound, initialised in time, rank order proven by
    a two-word gather -- is agreed by an all-reduce after each step, so that either every rank uses
    ``plsx_allgather`` or none does.  A rank whose init never returns reports failure at the agreement and all
    ranks fall back together (its helper thread is left behind as a daemon); what this cannot cover is a rank that
    DIES inside the open -- the peers then wait in the process group's own all-reduce until ITS timeout fires.
    Returns the Engine or None."""
    d = _dist()
    if d is None or d.get_backend() != 'nccl':
        return None
    group = d.group.WORLD
    if _NATIVE['state'] is not None and _NATIVE['group'] is group:
        return _NATIVE['engine'] if _NATIVE['state'] else None
    if _NATIVE['state'] is not None:                    # a communicator of an earlier process group
        release_native_comm()
    import torch
    from . import engine as _engine
    rank, world = d.get_rank(), d.get_world_size()
    dev = torch.device('cuda', torch.cuda.current_device())

    def agreed(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        d.all_reduce(flag, op=d.ReduceOp.MIN)
        return bool(flag.item())

    eng, why = None, ''
    try:
        eng = _engine.Engine(dev.index)
        eng.comm_load()
    except Exception as exc:                            # noqa: BLE001 -- any failure means "fall back", on all ranks
        why = 'bind: ' + str(exc)[:200]
    ok = agreed(not why)
    if ok:
        box = [None]
        if rank == 0:
            try:
                box[0] = eng.comm_unique_id()
            except Exception as exc:                    # noqa: BLE001
                why = 'unique id: ' + str(exc)[:200]
        d.broadcast_object_list(box, src=0)
        ok = box[0] is not None
        if ok:
            import threading
            done = {}

            def init():
                try:
                    torch.cuda.set_device(dev)
                    eng.comm_init(box[0], rank, world)
                    done['ok'] = True
                except Exception as exc:                # noqa: BLE001
                    done['why'] = 'init: ' + str(exc)[:200]
            th = threading.Thread(target=init, name='plsx-comm-init', daemon=True)
            th.start()
            th.join(NATIVE_INIT_TIMEOUT_S if timeout_s is None else float(timeout_s))
            if th.is_alive():
                why = 'init: ncclCommInitRank did not return within the timeout'
            elif not done.get('ok'):
                why = done.get('why', 'init failed')
            ok = agreed(not why)
    if ok:                                              # prove the rank order before anything depends on it
        try:
            mine = torch.full((2,), float(rank), dtype=torch.float64, device=dev)
            got = torch.empty((world, 2), dtype=torch.float64, device=dev)
            eng.allgather_into(mine, got)
            torch.cuda.synchronize(dev)
            good = bool((got[:, 0].cpu() == torch.arange(world, dtype=torch.float64)).all())
            if not good:
                why = 'probe all-gather returned the wrong rank order'
        except Exception as exc:                        # noqa: BLE001
            good, why = False, 'probe: ' + str(exc)[:200]
        ok = agreed(good)
    if not ok:
        if eng is not None:
            try:
                eng.close()
            except Exception:                           # noqa: BLE001
                pass
        eng = None
        if rank == 0:
            import warnings
            warnings.warn('plsx_allgather unavailable ({}): the collective falls back to the process group\'s '
                          'all_gather_into_tensor (RCCL all the same)'.format(why or 'a peer rank failed'))
    _NATIVE.update(state=bool(ok), engine=eng, why=why, group=group)
    if ok and not _NATIVE.get('atexit'):
        import atexit
        atexit.register(release_native_comm)            # before the interpreter tears torch / HIP down
        _NATIVE['atexit'] = True
    return eng


def release_native_comm():
    """Destroy the communicator (call before ``destroy_process_group``; every rank)."""
    eng = _NATIVE['engine']
    _NATIVE.update(state=None, engine=None, why='', group=None)
    if eng is not None:
        try:
            eng.close()
        except Exception:                               # noqa: BLE001 -- at exit the device may be gone already
            pass


def collective_name():
    """Which all-gather :func:`gather_device` issues under the current process group (a pure query)."""
    d = _dist()
    if d is None:
        return 'none (no process group)'
    if d.get_backend() == 'nccl' and native_comm() is not None:
        return 'plsx_allgather (RCCL ncclAllGather behind the C ABI)'
    return '{} all_gather_into_tensor (torch.distributed)'.format(d.get_backend())


def shared_seed(seed):
    """Seed every rank must use so that all ranks draw IDENTICAL index arrays.

    The reference draws permutation / bootstrap / split arrays from one
    RandomState (pyls/base.py:37,109,188); sharding them over ranks only works
    when every rank generates the same full arrays.  An integer seed is already
    shared.  ``None`` (numpy's global state) or a RandomState instance differs
    per process, so rank 0 draws one integer from it and broadcasts it."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return seed
    import numbers
    if isinstance(seed, numbers.Integral):
        return seed
    box = [None]
    if d.get_rank() == 0:
        rs = np.random.mtrand._rand if (seed is None or seed is np.random) else seed
        box[0] = int(rs.randint(0, 2 ** 31 - 1))
    d.broadcast_object_list(box, src=0)
    return box[0]


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n resamples owned by ``rank``; sizes
    differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


BOOT_PARTS = 1       # chunks per rank of the bootstrap shards (1 = one contiguous slice); a module constant, NOT read
                     # from the environment: every rank derives every rank's row order from it (collect_slices), so
                     # it must be identical everywhere.  Chunk-cyclic shards
                     # (2, 4) were measured on the emulated 8-rank critical path of c4 and LOSE: the last rank's
                     # bootstraps exist 26 ms into the step instead of 28 (all 10 000 bootstrap rows are drawn in
                     # 13 ms once the 10 000 permutations are, 17 ms), while every extra launch costs a wave of
                     # the small solver and a partly filled moment block: 254 ms against 250 (profiles/
                     # r03_rank_timeline.jsonl).  What the last rank waits for is the PERMUTATION draw, which the
                     # reference's stream order puts first.


def shard_chunks(n, rank, world, parts=None):
    """Chunk-cyclic shard of n resamples: the rows are cut into world * parts
    chunks (sizes differ by at most one) and rank r owns chunks r, r + world, ...
    Returns its [(lo, hi), ...] in ascending order.

    Why not one contiguous slice: every rank draws the FULL index arrays from the
    shared seed, in the reference's order (permutations, then bootstraps), and can
    only launch rows that exist.  With contiguous slices the last rank's rows are
    drawn last -- its device sits idle for the whole draw and then still has its
    whole shard to do; with cyclic chunks every rank has work after 1 / parts of
    the draw and only its last chunk (1 / parts of its shard) waits for the end.
    The bootstrap shards of the front-ends go through this with BOOT_PARTS chunks per
    rank -- 1 by default (see there: at c4 the draw is too fast for the chunks to pay);
    world 1 is the single range [0, n)."""
    n, world = int(n), int(world)
    if world <= 1:
        return [(0, n)]
    total = world * int(BOOT_PARTS if parts is None else parts)
    out = []
    for k in range(int(rank), total, world):
        lo, hi = shard_bounds(n, k, total)
        if hi > lo:
            out.append((lo, hi))
    return out


def shard_rows(n, rank, world, parts=None):
    """Global row numbers of :func:`shard_chunks`, in the rank's local order."""
    ch = shard_chunks(n, rank, world, parts)
    return np.concatenate([np.arange(lo, hi) for lo, hi in ch]) if ch else np.zeros(0, dtype=np.int64)


def gather_device(slices, sums, device=None):
    """The one collective on device tensors.

    slices: list of tensors whose LEADING axis is this rank's shard of resamples
            (equal trailing shapes on every rank, shard sizes may differ by one);
            each is padded to ``nmax`` rows, the largest shard, given as
            (tensor, nmax).
    sums:   list of tensors to be added over ranks (same shape everywhere).
    Returns (gathered, summed): gathered[i] has shape (world, nmax, ...) --
    the caller drops the padding rows with shard_bounds -- and summed[i] the
    rank-ordered sum.  With no process group (or world 1 on gloo) nothing moves.
    Everything is packed into ONE flat fp64 buffer so that exactly one
    all-gather is issued (plsx_allgather = RCCL behind the C ABI when the backend is nccl)."""
    import torch
    d = _dist()
    if d is None:
        return [_pad_rows(t, n)[None] for t, n in slices], list(sums)
    world = d.get_world_size()
    on_gpu = d.get_backend() == 'nccl'
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
    pieces, shapes = _pack(slices, sums)
    homes = [p.device for p in pieces]
    flat = torch.cat([p.to(device=device, dtype=torch.float64) for p in pieces]) if pieces else \
        torch.zeros(0, dtype=torch.float64, device=device)
    gathered = torch.empty((world, flat.numel()), dtype=torch.float64, device=device)
    comm = native_comm() if (on_gpu and flat.is_cuda) else None
    if comm is not None:
        with torch.cuda.device(flat.device):
            comm.allgather_into(flat, gathered)         # plsx_allgather on the current stream of the data
    else:
        d.all_gather_into_tensor(gathered.view(-1), flat)
    out_g, out_s = _unpack(gathered, shapes, len(slices), world)
    ns = len(slices)
    out_s = [acc.to(homes[ns + i]) if acc.device != homes[ns + i] else acc for i, acc in enumerate(out_s)]
    return out_g, out_s


def _pad_rows(t, n):
    import torch
    if t.shape[0] == n:
        return t
    if t.shape[0] > n:
        return t[:n]
    pad = torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad])


def _pack(slices, sums):
    """(tensor, nmax) slices padded to nmax rows + sums -> flat pieces and their shapes."""
    pieces, shapes = [], []
    for t, nmax in slices:
        t = _pad_rows(t, nmax)
        shapes.append(tuple(t.shape))
        pieces.append(t.reshape(-1))
    for t in sums:
        shapes.append(tuple(t.shape))
        pieces.append(t.reshape(-1))
    return pieces, shapes


def _unpack(gathered, shapes, n_slices, world):
    """(world, n) gathered buffer -> per-slice (world, nmax, ...) blocks and the rank-ordered sums."""
    out_g, out_s, off = [], [], 0
    for i, shp in enumerate(shapes):
        n = int(np.prod(shp)) if len(shp) else 1
        blk = gathered[:, off:off + n].reshape((world,) + shp)
        off += n
        if i < n_slices:
            out_g.append(blk)
        else:
            acc = blk[0].clone()
            for r in range(1, world):                   # fixed rank order -> deterministic
                acc += blk[r]
            out_s.append(acc)
    return out_g, out_s


def _surrogate_gather(slices, sums, world):
    """Stand-in for the all-gather of an EMULATED world (one GPU, no peers; bench.py --emulate-world): the packed
    buffer of this rank is replicated ``world`` times on the device -- the volume a real gather would deliver --
    and the partial sums are added in rank order, as after a real one.  The values are of course this rank's own."""
    import torch
    pieces, shapes = _pack(slices, sums)
    flat = torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.float64)
    gathered = flat.unsqueeze(0).repeat(world, 1)
    return _unpack(gathered, shapes, len(slices), world)


def _team_gather(slices, sums, rank, team):
    """The collective of a single-process team (team.py): every rank's thread packs on its own device and meets
    the others in ``Team.allgather`` (one plsx_allgather_all issued for all ranks)."""
    import torch
    dev = team.engines[rank].device
    pieces, shapes = _pack(slices, sums)
    flat = torch.cat([p.to(device=dev, dtype=torch.float64) for p in pieces]) if pieces else \
        torch.zeros(0, dtype=torch.float64, device=dev)
    gathered = team.allgather(rank, flat)
    return _unpack(gathered, shapes, len(slices), team.world)


def collect_device(slices, totals, sums, cyclic=(), emulate=None, team=None):
    """THE collective of a front-end call on tensors that are already where the backend wants them (device
    tensors under RCCL: the shard results never visit the host before the gather), with the result LEFT on the
    device of the inputs: the front-end finishes there (percentile intervals, bootstrap ratios, the layout of
    PLSResults) and ships only what PLSResults holds, once.

    slices: torch tensors whose leading axis is this rank's shard of ``totals[i]`` resamples -- contiguous
            (shard_bounds), or chunk-cyclic in the rank's local order (shard_rows) for the positions listed in
            ``cyclic``; sums: tensors to add over ranks.  Returns (list of tensors with the FULL leading axis
            in global order, list of rank-ordered sums), identical on every rank.  Without a process group
            nothing moves.  emulate = (rank, world): no peers, see _surrogate_gather.  team = (rank, Team): this thread is
    one rank of a single-process team (team.py); every rank's thread makes this call."""
    import torch
    d = _dist()
    if team is not None:
        rank, world = team[0], team[1].world
    elif emulate is None:
        if d is None:
            return list(slices), list(sums)
        rank, world = d.get_rank(), d.get_world_size()
    else:
        rank, world = emulate
    cyclic = set(cyclic)
    counts = []
    for i, n in enumerate(totals):
        if i in cyclic:
            counts.append([sum(hi - lo for lo, hi in shard_chunks(n, r, world)) for r in range(world)])
        else:
            counts.append([int(np.diff(shard_bounds(n, r, world))[0]) for r in range(world)])
    padded = [(t, max(c)) for t, c in zip(slices, counts)]
    if team is not None:
        got, summed = _team_gather(padded, sums, rank, team[1])
    elif emulate is not None:
        got, summed = _surrogate_gather(padded, sums, world)
    else:
        if d.get_backend() == 'nccl':           # RCCL: pack and gather on the GPU
            device = torch.device('cuda', torch.cuda.current_device())
            for t in list(slices) + list(sums):
                if t.is_cuda:
                    device = t.device
                    break
        else:                                   # gloo (CPU tests / single-GPU dry runs)
            device = torch.device('cpu')
        got, summed = gather_device(padded, sums, device)
    full = []
    for i, (blk, n) in enumerate(zip(got, totals)):
        home = slices[i].device
        cat = torch.cat([blk[r][:counts[i][r]] for r in range(world)], dim=0)       # (n, ...)
        if i in cyclic and world > 1:
            order = torch.from_numpy(np.concatenate([shard_rows(n, r, world) for r in range(world)])).to(cat.device)
            out = torch.empty_like(cat)
            out[order] = cat
            cat = out
        full.append(cat.to(home) if cat.device != home else cat)
    return full, summed


def collect_slices(slices, totals, sums, cyclic=()):
    """:func:`collect_device` with the gathered blocks as numpy arrays (regression front-end, tests)."""
    full, summed = collect_device(slices, totals, sums, cyclic=cyclic)
    return [t.detach().cpu().numpy() for t in full], summed


def collect(local_perm, n_perm, local_dist, n_boot, usum, usq):
    """Host-array form of :func:`collect_slices` (regression front-end, tests).
    local_perm (L, p_loc) ndarray or None; local_dist (T', L, r_loc) ndarray or
    None; usum / usq torch tensors (B, L) or None.  Returns (perm (L, n_perm) |
    None, dist (T', L, n_boot) | None, usum, usq) identical on every rank."""
    if _dist() is None:
        return local_perm, local_dist, usum, usq
    import torch
    slices, totals = [], []
    if local_perm is not None:
        slices.append(torch.from_numpy(np.ascontiguousarray(local_perm.T)))
        totals.append(n_perm)
    if local_dist is not None:
        slices.append(torch.from_numpy(np.ascontiguousarray(np.moveaxis(local_dist, -1, 0))))
        totals.append(n_boot)
    sums = [usum, usq] if usum is not None else []
    full, summed = collect_slices(slices, totals, sums)
    perm = dist_out = None
    k = 0
    if local_perm is not None:
        perm = np.ascontiguousarray(full[k].T)
        k += 1
    if local_dist is not None:
        dist_out = np.ascontiguousarray(np.moveaxis(full[k], 0, -1))
    if usum is not None:
        usum, usq = summed
    return perm, dist_out, usum, usq
