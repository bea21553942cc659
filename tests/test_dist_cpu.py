"""world_size-2 gloo (CPU) test of the N>1 path: contiguous split sharding and the single gather of
per-cycle score rows (device collective) + result dicts (host gather)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_result(i):
    acc = (i % 5) / 4.0
    loc = {n: [{"acc": acc, "predict_after_edit": "p%d" % i, "predict_before_edit": "b%d" % i}]
           for n in ["text_loc", "t3i3", "t1i4", "t2i4", "t1i2", "t1i3", "t2i1", "t2i2", "t3i1"]}
    return {"reliability": [{"acc": acc, "edit_time": 0.01 * i, "predict_after_edit": "r%d" % i}],
            "generality": {"text_rephrase": [{"acc": acc}], "image_rephrase": [{"acc": 1 - acc}]}, "locality": loc}


def _worker(rank, world, n, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import devqa_amd  # noqa: F401
    from devqa_amd.batched import BatchedEditEval, shard_range
    from devqa_amd.dist import gather_results, init_from_env
    r, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_range(n, rank, world)
    local = [_fake_result(i) for i in range(lo, hi)]
    rows = BatchedEditEval.score_rows(local, [(25, 0.5)] * len(local), lo)
    allres = gather_results(local, rows, n, rank, world, torch.device("cpu"))
    if rank == 0:
        q.put([r_["reliability"][0]["predict_after_edit"] for r_ in allres])
    else:
        assert allres is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_shard_and_gather_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, n, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out == ["r%d" % i for i in range(n)]  # concatenation in rank order == reference sample order


def test_shard_range_partitions():
    from devqa_amd.batched import shard_range
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_generic(rank, world, sizes, port, q):
    """The generic evaluator's split sharding: fake per-split results, real shard / gather / regroup code."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import devqa_amd  # noqa: F401
    from devqa_amd.dist import init_from_env
    from devqa_amd.evaluation.vllm_editor_eval import VLLMEditorEvaluation
    init_from_env("gloo")
    ids, splits = 0, []
    for sz in sizes:
        splits.append(list(range(ids, ids + sz)))
        ids += sz

    class Ed:
        device = "cpu"

        def name_of_editor_and_model(self):
            return "fake", "fake"
    ev = VLLMEditorEvaluation.__new__(VLLMEditorEvaluation)
    ev._run_splits = lambda editor, rd, ed: [[_fake_result(i) for i in sp] for sp in rd]
    out = ev._sequential_generic(Ed(), splits, splits)
    if rank == 0:
        q.put([[r["reliability"][0]["predict_after_edit"] for r in sp] for sp in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[2, 2, 3, 2, 2], [1], [2, 2]])
def test_generic_evaluator_shards_splits_world2(sizes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + len(sizes)) % 2000
    procs = [ctx.Process(target=_worker_generic, args=(r, 2, sizes, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want, i = [], 0
    for sz in sizes:
        want.append(["r%d" % j for j in range(i, i + sz)])
        i += sz
    assert out == want


def _run_bench(extra, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-cpu"] + extra, env=e, capture_output=True, text=True,
                       timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_n_spawns_n_ranks_world2():
    """`python bench.py --gpus 2` with no torchrun environment (the driver's documented form): the parent starts 2 child ranks, they
    form a process group (gloo here, RCCL on GPUs), shard, meet in the single gather, and rank 0's JSON line comes back with
    n_gpus = 2.  Weak: K * E cycles per rank; strong: K * E in total, block-partitioned."""
    rc, j, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--cycles-per-step", "5"])
    assert rc == 0, err[-1500:]
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["scaling"] == "weak" and j["config"]["cycles_total"] == 30 and j["steps"] == 3
    rc, j, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0", "--cycles-per-step", "5", "--scaling", "strong", "--strong-cycles", "15"])
    assert rc == 0, err[-1500:]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["cycles_total"] == 15
    rc, j, err = _run_bench(["--gpus", "1", "--steps", "2", "--cycles-per-step", "4"])
    assert rc == 0 and j["n_gpus"] == 1 and j["config"]["cycles_total"] == 8


def test_bench_strong_stream_is_rank_count_invariant(tmp_path):
    """`bench.py --gpus 2 --scaling strong` end to end (launcher, gloo group, block partition, every rank's block cut into >= 2
    sub-batches, the single gather): rank-order concatenation of the gathered score rows == the rows of the 1-rank run of the same
    stream, bit for bit -- also for a stream length that does not divide by the rank count or the batch size."""
    import numpy as np
    from bench import instrumented_iters, split_batches
    assert instrumented_iters(20, 4) == {2, 6, 10, 14, 18} and instrumented_iters(10, 4) == {2, 6} and instrumented_iters(8, 4) == {2, 6}
    assert instrumented_iters(3, 4) == set() and instrumented_iters(2, 4) == set() and instrumented_iters(20, 1) == set()      # = every step instrumented
    assert instrumented_iters(20, 4, pipelined=False) == set() and 19 not in instrumented_iters(20, 2) and instrumented_iters(7, 4) == {2}
    assert split_batches(125, 127) == [63, 62] and split_batches(1000, 127) == [125] * 8 and split_batches(1, 127) == [1]
    assert split_batches(0, 127) == [] and sum(split_batches(255, 127)) == 255 and max(split_batches(255, 127)) <= 127
    for total in (1000, 37):
        dumps = []
        for n in (1, 2, 3):
            f = str(tmp_path / ("rows_%d_%d.npy" % (total, n)))
            rc, j, err = _run_bench(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--cycles-per-step", "127", "--scaling", "strong",
                                     "--strong-cycles", str(total), "--dump-rows", f])
            assert rc == 0 and j["config"]["cycles_total"] == total and j["n_gpus"] == n, err[-1500:]
            dumps.append(np.load(f))
        assert dumps[0].shape == (total, 16) and [int(v) for v in dumps[0][:, 0]] == list(range(total))
        assert dumps[0].tobytes() == dumps[1].tobytes() == dumps[2].tobytes()


def test_bench_launcher_failure_modes():
    # a rank that exits non-zero makes the parent exit non-zero (and the surviving rank is ended, not left in the collective)
    rc, j, err = _run_bench(["--gpus", "2", "--steps", "2"], {"DEVQA_BENCH_FAIL_RANK": "1"})
    assert rc != 0 and "rank exit codes" in err
    # under a launcher whose WORLD_SIZE disagrees with --gpus the run refuses to print a mislabelled line
    rc, j, err = _run_bench(["--gpus", "2", "--steps", "2"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert rc == 2 and j is None and "WORLD_SIZE=4" in err
