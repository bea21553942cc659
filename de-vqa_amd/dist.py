"""Multi-GPU plumbing for the sharded edit stream: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests).

The data path has NO collective: every rank runs its contiguous block of splits on a full model
replica.  At the end one gather to rank 0 moves a fixed-width fp32 row per cycle
([n_local, 16]: sample id, 12 accuracies, edit_time, steps, final loss); decoded strings for
results.json travel host-side (gather_object) outside the timed path.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None, min_world=2):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if WORLD_SIZE >= min_world (2: a single process needs no group;
    the GPU test passes 1 to drive this very code path with one rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world < min_world or dist.is_initialized():
        return int(os.environ.get("RANK", "0")), world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        # bind the communicator to this rank's GPU up front: no device guessing in barrier(), RCCL set-up happens here
        dist.init_process_group(backend=backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def gather_score_rows(local_rows: np.ndarray, n_total: int, rank: int, world: int, device):
    """All ranks contribute [n_local,16] fp32; rank 0 gets [n_total,16] in rank (== sample) order.
    Block sizes differ by at most one row, so rows are padded to the max block for the collective."""
    from .batched import shard_range
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros((mx, local_rows.shape[1] if local_rows.ndim == 2 else 16), dtype=torch.float32, device=device)
    if len(local_rows):
        buf[:len(local_rows)] = torch.from_numpy(local_rows).to(device)
    if torch.device(device).type == "cuda" and os.environ.get("DEVQA_GATHER_ABI", "1") != "0":
        # the path-level entry point (include/devqa.h, devqa_gather_scores): ONE RCCL all-gather on the library's own communicator.
        # N > 1 on hardware is UNPINNED until a multi-GPU run is recorded (SCALE_r01/r02 were skipped by the driver): this leg has run
        # with one rank on the GPU box and with two ranks on gloo only; DEVQA_GATHER_ABI=0 takes torch.distributed's all_gather
        allrows = _score_comm(rank, world, torch.device(device)).gather_scores(buf.contiguous())
        outs = list(allrows.view(world, mx, -1).unbind(0))
    else:
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)  # gloo (CPU tests): the same collective through torch.distributed
    if rank != 0:
        return None
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)], 0)


_SCORE_COMM = {}


def _score_comm(rank, world, device):
    """lib.ScoreComm for this process group: rank 0 draws the RCCL id, the host-side object broadcast hands it to the others."""
    key = (rank, world, device.index or 0)
    if key not in _SCORE_COMM:
        from . import lib
        uid = [lib.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        _SCORE_COMM[key] = lib.ScoreComm(rank, world, uid[0], device.index or 0)
    return _SCORE_COMM[key]


def close_score_comms():
    """Destroy the library's RCCL communicators of this process (call before dist.destroy_process_group())."""
    for key in list(_SCORE_COMM):
        c = _SCORE_COMM.pop(key)
        try:
            c.close()
        except Exception as e:      # a failed teardown must not mask the run's result
            import warnings
            warnings.warn("ScoreComm close failed: %s" % e)


def gather_results(local_results, local_rows, n_total, rank, world, device):
    """-> on rank 0: list of all result dicts in split order + checks them against the gathered
    score rows; None elsewhere."""
    rows = gather_score_rows(local_rows, n_total, rank, world, device)
    objs = [None] * world if rank == 0 else None
    dist.gather_object(local_results, objs, dst=0)
    if rank != 0:
        return None
    allres = [r for part in objs for r in part]
    assert len(allres) == n_total == len(rows)
    for i, r in enumerate(allres):  # the device gather and the host gather must agree
        assert int(rows[i, 0]) == i
        assert abs(float(rows[i, 1]) - r["reliability"][0]["acc"]) < 1e-6
    return allres


def gather_results_ragged(local_results, local_rows, rank, world, device):
    """As gather_results, for blocks whose sizes are not the shard_range partition of a known total (the generic
    evaluator shards SPLITS, which may hold several samples each): sizes are exchanged first."""
    n_local = torch.tensor([len(local_results)], dtype=torch.int64, device=device)
    sizes_t = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes_t, n_local)
    sizes = [int(t.item()) for t in sizes_t]
    mx = max(max(sizes), 1)
    buf = torch.zeros((mx, 16), dtype=torch.float32, device=device)
    if len(local_rows):
        buf[:len(local_rows)] = torch.from_numpy(np.asarray(local_rows, np.float32)).to(device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    objs = [None] * world if rank == 0 else None
    dist.gather_object(local_results, objs, dst=0)
    if rank != 0:
        return None
    rows = np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)], 0)
    allres = [r for part in objs for r in part]
    assert len(allres) == len(rows)
    for i, r in enumerate(allres):
        assert int(rows[i, 0]) == i and abs(float(rows[i, 1]) - r["reliability"][0]["acc"]) < 1e-6
    return allres


def gather_split_results(local_splits, sizes, lo, rank, world, device):
    """Rank r evaluated the contiguous block of SPLITS starting at split `lo` (each a list of sample results; `sizes` = samples per
    split of the whole run); rank 0 gets all splits in order (the collective of fixed-width score rows + the host gather of the
    dicts, per SAMPLE, regrouped), the others None."""
    from .batched import BatchedEditEval
    first = sum(sizes[:lo])
    flat = [r for sp in local_splits for r in sp]
    rows = BatchedEditEval.score_rows(flat, [(0, 0.0)] * len(flat), first)
    dev = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    allres = gather_results_ragged(flat, rows, rank, world, dev)
    if allres is None:
        return None
    out, i = [], 0
    for sz in sizes:
        out.append(allres[i:i + sz])
        i += sz
    return out
