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


def gather_consensus(cons: torch.Tensor, lens: torch.Tensor, dst: int = 0, force: bool = False):
    """cons: uint8 [sum(lens)] consensus bytes of this rank's windows, lens: int64 [n_local].
    Returns (cons_all, lens_all) on rank `dst` (window order), (None, None) elsewhere.
    Two collectives: all_gather of (n_windows, n_bytes), then an all_gather of payloads padded to
    the largest shard (one large message per peer link; no ring dependency on payload size)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return cons, lens
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = cons.device
    meta = torch.tensor([lens.numel(), cons.numel()], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = torch.stack(metas).cpu()
    max_w, max_b = int(metas[:, 0].max()), int(metas[:, 1].max())
    pl = torch.zeros(max(max_w, 1), dtype=torch.int64, device=dev)
    pl[:lens.numel()] = lens
    pc = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    pc[:cons.numel()] = cons
    all_l = [torch.empty_like(pl) for _ in range(world)]
    all_c = [torch.empty_like(pc) for _ in range(world)]
    dist.all_gather(all_l, pl)
    dist.all_gather(all_c, pc)
    if rank != dst:
        return None, None
    lens_all = torch.cat([all_l[r][:int(metas[r, 0])] for r in range(world)])
    cons_all = torch.cat([all_c[r][:int(metas[r, 1])] for r in range(world)])
    return cons_all, lens_all
