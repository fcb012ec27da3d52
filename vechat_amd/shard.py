"""Multi-GPU side of the path: windows are independent units (src/polisher.cpp:497-517 submits one
task per window), so ranks take contiguous window ranges and the only exchange is the final
variable-length gather of corrected sequences to rank 0 (RCCL over xGMI when the process group is
"nccl"; the same code runs over gloo on CPU tensors in the tests)."""
import torch
import torch.distributed as dist


def shard_range(n_windows: int, rank: int, world: int):
    """Contiguous, order-preserving split: concatenating the ranks' outputs in rank order restores
    window order, which the stitching in Polisher::polish (polisher.cpp:525-547) relies on."""
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def estimated_cells(batch):
    """Per-window DP work estimate of SURVEY 8(e): sum over layers of len * (L + 0.3 * sum of the previous
    layers' lengths) -- the graph a layer is aligned to has grown by about a third of what came before."""
    import numpy as np
    lens = np.diff(batch.seq_off.astype(np.int64))
    out = np.zeros(batch.n_windows, dtype=np.float64)
    for w in range(batch.n_windows):
        s0, s1 = int(batch.win_seq_off[w]), int(batch.win_seq_off[w + 1])
        ll = lens[s0 + 1:s1].astype(np.float64)
        prev = np.concatenate([[0.0], np.cumsum(ll)[:-1]]) if ll.size else ll
        out[w] = float(np.sum(ll * (lens[s0] + 0.3 * prev)))
    return out


def shard_range_balanced(cost, rank: int, world: int):
    """Contiguous, order-preserving split with (nearly) equal summed cost per rank: rank r takes the windows
    whose cumulative cost midpoint falls into the r-th of `world` equal slices.  Every window goes to exactly
    one rank and the ranges stay in rank order, as the stitching needs."""
    import numpy as np
    cost = np.asarray(cost, dtype=np.float64)
    n = cost.size
    if n == 0 or float(cost.sum()) <= 0.0:
        return shard_range(n, rank, world)
    mid = np.cumsum(cost) - 0.5 * cost
    owner = np.minimum((mid / cost.sum() * world).astype(np.int64), world - 1)     # non-decreasing
    lo = int(np.searchsorted(owner, rank, side="left"))
    hi = int(np.searchsorted(owner, rank, side="right"))
    return lo, hi


def gather_consensus(cons: torch.Tensor, lens: torch.Tensor, dst=0, force: bool = False):
    """cons: uint8 [sum(lens)] consensus bytes of this rank's windows, lens: int64 [n_local].
    Returns (cons_all, lens_all) on rank `dst` (window order), (None, None) elsewhere; dst=None: on every rank.

    dst = r (the path's one exchange): a small all_gather of (n_windows, n_bytes), then every other rank SENDS
    its exact payload to r and r receives each into its place of the output -- point-to-point over the xGMI link between the
    two GPUs, no padding to the largest shard, nothing delivered to ranks that do not need it.
    dst = None: every rank needs everything (sharing device-aligned CIGAR strings): all_gather of padded payloads."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return cons, lens
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = cons.device
    meta = torch.tensor([lens.numel(), cons.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu()
    nw, nb = [int(x) for x in metas[:, 0]], [int(x) for x in metas[:, 1]]
    if dst is not None:
        if rank != dst:
            ops = []
            if nw[rank]:
                ops.append(dist.P2POp(dist.isend, lens.contiguous(), dst))
            if nb[rank]:
                ops.append(dist.P2POp(dist.isend, cons.contiguous(), dst))
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
            return None, None
        lens_all = torch.empty(sum(nw), dtype=torch.int64, device=dev)
        cons_all = torch.empty(sum(nb), dtype=torch.uint8, device=dev)
        ops, wo, bo = [], 0, 0
        for r in range(world):
            lv, cv = lens_all[wo:wo + nw[r]], cons_all[bo:bo + nb[r]]
            if r == rank:
                lv.copy_(lens); cv.copy_(cons)
            else:
                if nw[r]:
                    ops.append(dist.P2POp(dist.irecv, lv, r))
                if nb[r]:
                    ops.append(dist.P2POp(dist.irecv, cv, r))
            wo += nw[r]; bo += nb[r]
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return cons_all, lens_all
    max_w, max_b = max(nw), max(nb)
    pl = torch.zeros(max(max_w, 1), dtype=torch.int64, device=dev)
    pl[:lens.numel()] = lens
    pc = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    pc[:cons.numel()] = cons
    all_l = [torch.empty_like(pl) for _ in range(world)]
    all_c = [torch.empty_like(pc) for _ in range(world)]
    dist.all_gather(all_l, pl)
    dist.all_gather(all_c, pc)
    lens_all = torch.cat([all_l[r][:nw[r]] for r in range(world)])
    cons_all = torch.cat([all_c[r][:nb[r]] for r in range(world)])
    return cons_all, lens_all
