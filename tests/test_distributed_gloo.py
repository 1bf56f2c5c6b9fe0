"""Multi-process path on CPU (gloo, world_size 2): pair sharding + the result gather.

The per-pair compute needs a GPU, so each rank produces its shard's match records with the
oracle (small images); what is under test is the product's distributed layer
(onnx_image_processing_amd.distributed): contiguous sharding, record packing, gather order and
the max-over-ranks timing reduction -- the same code bench.py runs over RCCL."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _records_for(seeds):
    """Match records (len(seeds), Mx, 6) of small pairs via the oracle (CPU checker)."""
    sys.path.insert(0, ROOT)
    from onnx_image_processing_amd import distributed as D
    from onnx_image_processing_amd.synth import synth_batch
    from oracle import numpy_oracle as O
    t = np.load(os.path.join(ROOT, "onnx_image_processing_amd", "data", "bad_tables.npz"))
    recs = []
    for s in seeds:
        a, b = synth_batch(s, 1, 64, 96)
        k1, k2, p = O.match_pair(a, b, t["box_256"], t["thr_256"], 32, binarize=True, soft_binarize=False,
                                 epsilon=0.05, nms_radius=2, sinkhorn_iterations=5)
        mk1, mk2, sc, valid, _ = O.mnn_extract(p, k1, k2, 16, 0.1)
        recs.append(D.pack_records(torch.from_numpy(mk1), torch.from_numpy(mk2), torch.from_numpy(sc),
                                   torch.from_numpy(valid)))
    return torch.cat(recs, 0) if recs else torch.zeros((0, 16, 6), dtype=torch.float32)


def _worker(rank, world, port, total, out_path, collective="gather"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from onnx_image_processing_amd import distributed as D
    r, w, _ = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    begin, end = D.shard_range(total, rank, world)
    rec = _records_for(range(5000 + begin, 5000 + end))
    assert rec.shape[0] == D.shard_sizes(total, world)[rank]
    gathered = D.gather_records(rec, dst=0, total=total, collective=collective)
    if total % world == 0:              # equal shards: the form without `total` gives the same
        again = D.gather_records(rec, dst=0, collective=collective)
        assert (again is None) == (rank != 0) and (rank != 0 or torch.equal(again, gathered))
    else:                               # a wrong shard size is caught before the collective, on the rank that has it
        with pytest.raises(ValueError):
            D.gather_records(torch.cat([rec, rec[:1], torch.zeros(1, 16, 6)]), dst=0, total=total)
    slowest = D.barrier_max_ms(10.0 * (rank + 1), "cpu")
    assert slowest == 10.0 * world
    if rank == 0:
        torch.save(gathered, out_path)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from onnx_image_processing_amd.distributed import shard_range
    for total in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_record_pack_roundtrip():
    from onnx_image_processing_amd.distributed import pack_records, unpack_records
    mk1, mk2 = torch.rand(3, 5, 2), torch.rand(3, 5, 2)
    sc = torch.rand(3, 5)
    valid = sc > 0.5
    a, b, c, d = unpack_records(pack_records(mk1, mk2, sc, valid))
    assert torch.equal(a, mk1) and torch.equal(b, mk2) and torch.equal(c, sc) and torch.equal(d, valid)


@pytest.mark.parametrize("total,collective", [(4, "gather"), (5, "gather"), (5, "all_gather"), (1, "gather")])
def test_two_rank_gather_equals_single_process(tmp_path, total, collective):
    """4 pairs: equal shards.  5 pairs / 2 ranks: shards of 3 and 2 (padded for the collective, trimmed on rank 0),
    through both collectives.  1 pair: rank 1 holds an empty shard."""
    world = 2
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(world, _free_port(), total, out, collective), nprocs=world, join=True)
    gathered = torch.load(out)
    assert gathered.shape == (total, 16, 6)
    assert torch.equal(gathered, _records_for(range(5000, 5000 + total)))     # global pair order preserved


def test_gather_records_argument_checks():
    from onnx_image_processing_amd.distributed import gather_records
    rec = torch.zeros(3, 4, 6)
    assert gather_records(rec, total=3) is rec                                # single process: identity
    with pytest.raises(ValueError):
        gather_records(rec, total=4)
    with pytest.raises(ValueError):
        gather_records(rec, collective="ring")


def _bench_worker(rank, world, port, total, out_path):
    """bench.py's N > 1 control flow on CPU: init -> shard -> timed region (barriers, per-step clock) -> gather ->
    max-over-ranks reduction, with the per-pair compute replaced by oracle-made records."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    from onnx_image_processing_amd import distributed as D
    D.init(backend="gloo")
    begin, end = D.shard_range(total, rank, world)
    rec = _records_for(range(5000 + begin, 5000 + end))
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.002 * (rank + 1))                       # ranks of different speed: the reduction must take the slowest
        return D.gather_records(rec, dst=0, total=total)

    elapsed_ms, per_step, out, own_ms = bench.run_timed(step, steps=4, warmup=2, world=world, device="cpu", sync=lambda: None)
    assert 0 < own_ms <= elapsed_ms
    facts = bench.world_facts(own_ms, 4, "cpu")
    assert facts["ranks_seen"] == world and facts["backend"] == "gloo" and len(facts["ms_per_step_per_rank"]) == world
    assert all(t > 0 for t in facts["ms_per_step_per_rank"])     # (a gather synchronises the ranks: their times are close)
    assert len(calls) == 6 and len(per_step) == 4 and all(t > 0 for t in per_step)
    assert elapsed_ms >= 4 * 2.0 * world * 0.9                # the slowest rank's time, on every rank
    stats = bench.step_stats(per_step)
    assert stats["min"] <= stats["median"] <= stats["max"]
    if rank == 0:
        torch.save(out, out_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _pipelined_worker(rank, world, port, total, out_path):
    """bench.PipelinedGather under gloo: step i returns the gathered records of step i - 1, drain() those of the last
    step; every step's records arrive, in global pair order, ragged shards included."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    from onnx_image_processing_amd import distributed as D
    D.init(backend="gloo")
    begin, end = D.shard_range(total, rank, world)
    base = torch.arange(begin, end, dtype=torch.float32)[:, None, None].expand(end - begin, 4, 6).contiguous()
    gather = bench.PipelinedGather(total=total)
    seen = []

    def step():
        k = len(seen) + 1
        seen.append(gather(base * k))                         # step k's records = k * (global pair index)
        return seen[-1]

    elapsed_ms, per_step, out, own = bench.run_timed(step, steps=3, warmup=2, world=world, device="cpu", sync=lambda: None,
                                                     drain=gather.drain)
    want = torch.arange(total, dtype=torch.float32)[:, None, None].expand(total, 4, 6)
    if rank == 0:
        assert seen[0] is None                                # nothing gathered before the first wait
        for k in (2, 3):                                      # warm-up: step k returned step k - 1's records ...
            assert torch.equal(seen[k - 1], want * (k - 1)) or k == 3
        assert torch.equal(seen[1], want * 1)
        assert torch.equal(seen[3], want * 3) and torch.equal(seen[4], want * 4)   # ... (the drain after warm-up took step 2's)
        assert torch.equal(out, want * 5)                     # drain(): the last step's records, before the clock stopped
        torch.save(out, out_path)
    else:
        assert all(x is None for x in seen) and out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_pipelined_gather_delivers_every_step(tmp_path, total):
    out = str(tmp_path / "pipe.pt")
    mp.spawn(_pipelined_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    assert torch.load(out).shape == (total, 4, 6)
    import bench
    g = bench.PipelinedGather()                               # one process: the identity, nothing deferred
    rec = torch.rand(3, 4, 6)
    assert g(rec) is rec and g.drain() is rec


def test_bench_multi_rank_control_flow_dry_run(tmp_path):
    total, world = 5, 2
    out = str(tmp_path / "bench_out.pt")
    mp.spawn(_bench_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    assert torch.equal(torch.load(out), _records_for(range(5000, 5000 + total)))


# ---------------------------------------------------------------------------------------------- bench.py launcher
def _run_bench(argv, extra_env=None, launcher=()):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (the way the driver calls it): the parent starts two
    fresh ranks, rank 0's single JSON line comes back and says what the process group really was."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs-per-gpu", "5", "--dry-run"])
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1
    line = lines[0]
    assert line["dry_run"] is True and "DRY RUN" in line["data"]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert len(line["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in line["ms_per_step_per_rank"])
    assert line["config"]["global_pairs_per_step"] == 10 and line["gathered_in_global_pair_order"] is True
    assert line["steps"] == 3 and line["warmup"] == 1 and line["ms_per_step"] >= max(line["ms_per_step_per_rank"]) * 0.5
    assert line["value"] == pytest.approx(10 / (line["ms_per_step"] * 1e-3))


def test_bench_under_torchrun_and_single_rank_unchanged():
    """The contract's own launch form still works (ranks exist already: no second launcher level), and --gpus 1 runs
    in-process."""
    port = str(_free_port())
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs-per-gpu", "3", "--dry-run"],
                          launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                    "--master-addr", "127.0.0.1", "--master-port", port))
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1 and lines[0]["ranks_seen"] == 2 and lines[0]["n_gpus"] == 2
    r, lines = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs-per-gpu", "3", "--dry-run"])
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1 and lines[0]["ranks_seen"] == 1 and lines[0]["backend"] is None


def test_bench_launcher_reports_a_dead_rank():
    """A rank that dies: the parent terminates the others (they would wait in a collective), prints no JSON line as a
    result and exits non-zero."""
    t0 = time.time()
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs-per-gpu", "3", "--dry-run"],
                          extra_env={"MI_BENCH_DRY_RUN_FAIL_RANK": "1"})
    assert r.returncode != 0 and not lines
    assert "rank 1 exited with code 3" in r.stderr
    assert time.time() - t0 < 120


def test_bench_parent_makes_no_gpu_call_before_launching():
    """The launcher branch sits before the first torch.cuda call of main() (a process that has initialised the GPU
    must not be the one that spawns / is replaced)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main()"):]
    assert main.index("launch_ranks(") < main.index("torch.cuda.")
    launcher = src[src.index("def launch_ranks"):src.index("def world_facts")]
    assert "torch.cuda" not in launcher and "os.exec" not in src and "execv" not in src
