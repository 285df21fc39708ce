"""
Multi-GPU sharding of the resampling loops: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  Resamples are independent given their index arrays -- the
reference runs them in independent joblib workers (pyls/base.py:490-507,
644-650) -- so rank r takes a contiguous slice of the permutations and of the
bootstraps, every rank holds a full replica of X, and there is exactly ONE
collective: an all-gather of a packed per-rank buffer

    [ perm_singval slice | distrib slice | partial sum U | partial sum U^2 ]

after which every rank concatenates the slices in rank order and adds the
partial sums in rank order (fixed order -> deterministic).
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


def _flat_layout(Lp, Tp, L, B, n_perm, n_boot, world, with_boot):
    pmax = shard_bounds(n_perm, 0, world)[1] if n_perm else 0
    rmax = shard_bounds(n_boot, 0, world)[1] if n_boot else 0
    sizes = dict(perm=Lp * pmax, dist=Tp * L * rmax,
                 usum=B * L if with_boot else 0, usq=B * L if with_boot else 0)
    return pmax, rmax, sizes


def collect(local_perm, n_perm, local_dist, n_boot, usum, usq):
    """The one collective.  local_perm (L, p_loc) ndarray or None; local_dist
    (T', L, r_loc) ndarray or None; usum / usq torch tensors (B, L) or None.
    Returns (perm (L, n_perm) | None, dist (T', L, n_boot) | None, usum, usq)
    identical on every rank."""
    rank, world = rank_world()
    if world == 1:
        return local_perm, local_dist, usum, usq
    import torch
    d = _dist()
    with_boot = usum is not None
    home = usum.device if with_boot else None
    if d.get_backend() == 'nccl':           # RCCL: pack and gather on the GPU
        device = home if home is not None else torch.device('cuda', torch.cuda.current_device())
    else:                                   # gloo (CPU tests / single-GPU dry runs)
        device = torch.device('cpu')
        if with_boot:
            usum, usq = usum.to(device), usq.to(device)
    # the permutation block may carry extra rows (split-half nulls ride along)
    Lp = local_perm.shape[0] if local_perm is not None else 0
    L = local_dist.shape[1] if local_dist is not None else (usum.shape[1] if with_boot else 0)
    Tp = local_dist.shape[0] if local_dist is not None else 0
    B = usum.shape[0] if with_boot else 0
    pmax, rmax, sizes = _flat_layout(Lp, Tp, L, B, n_perm if local_perm is not None else 0,
                                     n_boot if local_dist is not None else 0, world, with_boot)
    total = sum(sizes.values())
    flat = torch.zeros(total, dtype=torch.float64, device=device)
    off = 0
    if sizes['perm']:
        blk = np.zeros((Lp, pmax))
        blk[:, :local_perm.shape[1]] = local_perm
        flat[off:off + sizes['perm']] = torch.from_numpy(blk.ravel()).to(device)
    off += sizes['perm']
    if sizes['dist']:
        blk = np.zeros((Tp, L, rmax))
        blk[:, :, :local_dist.shape[2]] = local_dist
        flat[off:off + sizes['dist']] = torch.from_numpy(blk.ravel()).to(device)
    off += sizes['dist']
    if with_boot:
        flat[off:off + sizes['usum']] = usum.reshape(-1)
        off += sizes['usum']
        flat[off:off + sizes['usq']] = usq.reshape(-1)
    gathered = torch.empty((world, total), dtype=torch.float64, device=device)
    d.all_gather_into_tensor(gathered.view(-1), flat)

    perm = dist_out = None
    off = 0
    if sizes['perm']:
        host = gathered[:, off:off + sizes['perm']].cpu().numpy().reshape(world, Lp, pmax)
        perm = np.concatenate([host[r][:, :np.diff(shard_bounds(n_perm, r, world))[0]]
                               for r in range(world)], axis=1)
    off += sizes['perm']
    if sizes['dist']:
        host = gathered[:, off:off + sizes['dist']].cpu().numpy().reshape(world, Tp, L, rmax)
        dist_out = np.concatenate([host[r][:, :, :np.diff(shard_bounds(n_boot, r, world))[0]]
                                   for r in range(world)], axis=2)
    off += sizes['dist']
    if with_boot:
        parts = gathered[:, off:off + sizes['usum']].reshape(world, B, L)
        qarts = gathered[:, off + sizes['usum']:off + 2 * sizes['usum']].reshape(world, B, L)
        usum, usq = parts[0].clone(), qarts[0].clone()
        for r in range(1, world):                      # fixed rank order
            usum += parts[r]
            usq += qarts[r]
        if home is not None and usum.device != home:
            usum, usq = usum.to(home), usq.to(home)
    return perm, dist_out, usum, usq
