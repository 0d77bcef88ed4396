"""CPU, world_size 2 over gloo: the data-parallel exchange of the hot path (one all-reduce of the flat gradient
buffer, 1/world fused into the optimiser, rank-local BN buffers, even pair-preserving shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sound_event_detection_dcase2017_task4_amd import parallel
    r, w, lr = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    # parameters: rank 0's values win
    flat = torch.full((1000,), float(rank + 1))
    parallel.broadcast_flat(flat)
    assert torch.all(flat == 1.0)
    # gradients: sum over ranks, in one call and in 3 buckets, sync and async
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    parallel.allreduce_flat_grad(g)
    assert torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 3)
    g2 = torch.ones(1001) * (rank + 1)
    hs = parallel.allreduce_flat_grad(g2, buckets=3, async_op=True)
    for h in hs:
        h.wait()
    assert torch.all(g2 == 3.0)
    # buffers broadcast once at start
    bn = torch.nn.BatchNorm2d(4)
    bn.running_mean.fill_(float(rank))
    parallel.broadcast_buffers(bn)
    assert torch.all(bn.running_mean == 0.0)
    # data-parallel SGD-equivalence on a toy problem: mean of per-rank gradients == gradient of the global batch
    torch.manual_seed(0)
    wgt = torch.randn(5, requires_grad=True)
    X = torch.randn(8, 5); Y = torch.randn(8)
    lo, hi = parallel.shard_range(8, rank, world)
    assert (hi - lo) == 4 and lo % 2 == 0                          # mixup pairs (2i, 2i+1) stay on one rank
    loss = ((X[lo:hi] @ wgt - Y[lo:hi]) ** 2).mean()
    loss.backward()
    gl = wgt.grad.clone()
    parallel.allreduce_flat_grad(gl)
    gl /= world
    full = torch.autograd.grad(((X @ wgt.detach().requires_grad_(True) - Y) ** 2).mean(), [])[0] if False else None
    w2 = wgt.detach().clone().requires_grad_(True)
    ((X @ w2 - Y) ** 2).mean().backward()
    assert torch.allclose(gl, w2.grad, atol=1e-6)
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    out = str(tmp_path / "ok.txt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_single_process_is_a_noop():
    sys.path.insert(0, REPO)
    from sound_event_detection_dcase2017_task4_amd import parallel
    g = torch.ones(10)
    assert parallel.allreduce_flat_grad(g) == [] and torch.all(g == 1)
    assert parallel.world_size() == 1 and parallel.shard_range(10, 0, 1) == (0, 10)


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sound_event_detection_dcase2017_task4_amd import parallel
    parallel.init_from_env(backend="gloo")
    # 6 "parameters" in one flat buffer, cut into 3 buckets; parameter 3 never takes part (like att_block.bn_att)
    numels = [4, 6, 10, 2, 20, 8]
    offsets = list(np.cumsum([0] + numels[:-1]))
    flat = torch.zeros(sum(numels))
    gb = parallel.GradBuckets(flat, offsets, numels, cuts=[offsets[2], offsets[4]])
    assert gb.ranges == [(0, 10), (10, 22), (22, 50)] and gb.bucket_of == [0, 0, 1, 1, 2, 2]
    for step in range(2):
        for i in (0, 1, 2, 4, 5):
            gb.expect(i)                                          # forward pass
        gb.new_gradients()                                        # zero_grad (after the forward, as in main.py)
        flat.zero_()
        for i in (5, 4, 2, 1, 0):                                 # backward: from the end of the buffer
            flat[offsets[i]:offsets[i] + numels[i]] = float((rank + 1) * (i + 1) * (step + 1))
            fired_before = list(gb.issue_order)
            gb.ready(i)
            if i == 4:
                assert gb.issue_order == [2] and fired_before == []       # the tail bucket left as soon as it was complete
            if i == 2:
                assert gb.issue_order == [2, 1]                           # parameter 3 is not expected: does not block
        gb.finish()
        assert gb.issue_order == [2, 1, 0]
        want = torch.zeros_like(flat)
        for i in (0, 1, 2, 4, 5):
            want[offsets[i]:offsets[i] + numels[i]] = 3.0 * (i + 1) * (step + 1)  # (1 + 2) * ...
        assert torch.equal(flat, want), (flat, want)
        gb.begin_step()
    # a gradient that arrives without having been announced is still reduced (at finish)
    flat.fill_(float(rank + 1))
    gb.finish()
    assert torch.all(flat == 3.0)
    gb.begin_step()
    # two backward passes in one cycle overwrite gradients: refused loudly
    gb.expect(5); gb.ready(5); gb.expect(5); gb.ready(5)
    try:
        gb.finish()
        raise AssertionError("second backward was not detected")
    except RuntimeError as e:
        assert "direct_grads=False" in str(e)
    # one global batch, one RNG state: rows of rank r
    assert parallel.shard_rows(64, rank, world) == (32 * rank, 32 * rank + 32)
    torch.manual_seed(100 + rank)
    parallel.broadcast_rng_state()
    v = torch.rand(3)
    vs = [torch.zeros(3) for _ in range(world)]
    dist.all_gather(vs, v)
    assert torch.equal(vs[0], vs[1])
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


def _ranks_seen_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sound_event_detection_dcase2017_task4_amd import parallel
    import bench
    parallel.init_from_env(backend="gloo")
    seen = bench.ranks_seen(torch.device("cpu"))          # all-reduce of a 1 from every rank: what the bench rows report
    assert seen == world, seen
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok %d" % seen)
    dist.destroy_process_group()


def test_bench_rows_prove_the_rank_count_by_an_all_reduce(tmp_path):
    """`dist.n_ranks_seen` of the weak and strong rows of `bench.py --gpus N` (bench.ranks_seen): the sum over ranks of 1."""
    import bench
    assert bench.ranks_seen(torch.device("cpu")) == 1      # no process group: one rank
    out = str(tmp_path / "ok.txt")
    mp.spawn(_ranks_seen_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok 2"


def test_grad_buckets_world_size_2_gloo(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _forced_one_rank_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SED_FORCE_DIST="1")
    from sound_event_detection_dcase2017_task4_amd import parallel
    r, w, lr = parallel.init_from_env(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and parallel.collectives_on()
    flat = torch.arange(50, dtype=torch.float32)
    parallel.broadcast_flat(flat)                                 # a real broadcast among one rank: values unchanged
    assert torch.equal(flat, torch.arange(50, dtype=torch.float32))
    numels = [10, 15, 25]
    offsets = [0, 10, 25]
    gb = parallel.GradBuckets(flat, offsets, numels, cuts=[offsets[1]])
    for i in range(3):
        gb.expect(i)
    gb.new_gradients()
    for i in (2, 1, 0):
        gb.ready(i)
    assert gb.issue_order == [1] or gb.issue_order == [1, 0]     # the tail bucket went to the backend from inside "backward"
    gb.finish()
    assert gb.issue_order == [1, 0] and torch.equal(flat, torch.arange(50, dtype=torch.float32))   # sum over one rank
    gb.begin_step()
    parallel.broadcast_rng_state()
    parallel.barrier()
    parallel.shutdown()
    open(out, "w").write("ok")


def test_forced_one_rank_group_runs_every_exchange_step(tmp_path):
    """SED_FORCE_DIST=1 (tests only): a ONE-rank job builds its process group and sends every broadcast, bucket and barrier
    through it -- the switch that lets the RCCL branch execute on the 1-GPU boxes (tests/test_gpu_parallel.py); here over gloo."""
    out = str(tmp_path / "ok.txt")
    mp.spawn(_forced_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert open(out).read() == "ok"


def test_shard_rows_keeps_mixup_pairs_together():
    sys.path.insert(0, REPO)
    from sound_event_detection_dcase2017_task4_amd import parallel
    assert [parallel.shard_rows(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        parallel.shard_rows(64, 0, 3)                 # does not split evenly
    with pytest.raises(ValueError):
        parallel.shard_rows(12, 0, 4)                 # 3 waveforms per rank: a mixup pair would straddle ranks
    assert parallel.shard_rows(12, 1, 4, pair=False) == (3, 6)


def test_bucket_cuts_of_the_real_models():
    """[bn0 + block1 + block2 | block3 | block4 (+ fc)] and an extra [gru + att_block] bucket for configs[3]/[4]."""
    sys.path.insert(0, REPO)
    from sound_event_detection_dcase2017_task4_amd.optim import bucket_cuts
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    for mt, want in (("Cnn_9layers_FrameAvg", ["conv_block3.conv1.weight", "conv_block4.conv1.weight"]),
                     ("Cnn_9layers_Gru_FrameAtt", ["conv_block3.conv1.weight", "conv_block4.conv1.weight", "gru.weight_ih_l0"])):
        m = getattr(models, mt)(32000, 1024, 320, 64, 50, 14000, 17)
        named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
        numels = [p.numel() for _, p in named]
        offsets = [int(v) for v in np.cumsum([0] + numels[:-1])]
        cuts = bucket_cuts([n for n, _ in named], offsets, numels)
        assert [named[offsets.index(c)][0] for c in cuts] == want, (mt, cuts)


def test_sharded_sampler_is_a_slice_of_one_stream():
    """N ranks walking ONE seed-1234 sampler stream and taking their rows reproduce the single-process batches
    (reference data_generator.py:52-101 feeds one batch to DataParallel, which scatters it)."""
    sys.path.insert(0, REPO)
    from sound_event_detection_dcase2017_task4_amd.utils.data_generator import ShardedBatchSampler, TrainSampler
    path = "synthetic:50:3200"
    whole = iter(TrainSampler(path, 16, random_seed=1234))
    parts = [iter(ShardedBatchSampler(TrainSampler(path, 16, random_seed=1234), 4 * r, 4 * r + 4)) for r in range(4)]
    for _ in range(9):                                            # crosses the reshuffle at the wrap-around
        full = [m["index_in_hdf5"] for m in next(whole)]
        got = sum(([m["index_in_hdf5"] for m in next(p)] for p in parts), [])
        assert got == full


def test_bench_gpus_n_never_degrades_to_one_rank():
    """`python bench.py --gpus 8` with no launcher and fewer than 8 GPUs (none here) must fail loudly, not print n_gpus 1."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode != 0 and "needs 8 GPUs" in r.stderr and '"n_gpus"' not in r.stdout


def test_bench_roofline_traffic_lookup_matches_the_kernels_that_exist():
    """bench.py fills roofline.traffic / roofline_frontend.traffic from the committed PMC digest by kernel-name substring:
    the substrings must name kernels that exist in csrc/ AND in the newest profiles/rNN/pmc_traffic.json."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    csrc = os.path.join(REPO, "sound_event_detection_dcase2017_task4_amd", "csrc")
    source = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith(".hip"))
    text = open(os.path.join(REPO, "bench.py")).read()
    import re
    front = re.search(r'pmc_traffic\(\["([a-z0-9_]+)"\]', text).group(1)
    for sub in [front] + [s for subs in bench.FAMILY_KERNELS.values() for s in subs]:
        assert sub in source, sub
    # a digest is quoted only while the kernel's sources are the ones it was collected on (bench.pmc_fresh); otherwise the
    # figure is null and the source string says why
    tr, src = bench.pmc_traffic([front])
    assert (tr and tr > 786e6 and src.startswith("profiles/r")) or (tr is None and "NOT quoted" in src), (tr, src)   # >= the algorithmic 786.6 MB per launch
    tr, src = bench.pmc_traffic(bench.FAMILY_KERNELS["conv3x3_sf16_mfma(fwd+dgrad)"])      # the default convolution path
    assert (tr and tr > 1e9) or (tr is None and "NOT quoted" in src), (tr, src)


def test_bucket_refuses_to_fire_before_its_side_stream_gradients_are_joined():
    """Ordering contract of the bucketed all-reduce (one process, no backend needed): a bucket is handed to the backend only
    behind the MAIN stream, so a weight gradient still running on the side stream must have been joined first.
    GradBuckets asks its owner right before firing (`pre_fire_check`); FusedAdamAmsgrad answers from ops._PENDING."""
    from sound_event_detection_dcase2017_task4_amd import ops, parallel
    flat = torch.zeros(100)
    gb = parallel.GradBuckets(flat, offsets=[0, 40, 70], numels=[40, 30, 30], cuts=[70])
    seen = []

    class FakeSink(object):
        def __init__(self, index):
            self.index, self.opt = index, "opt"

    def check(bucket, indices):
        seen.append((bucket, list(indices)))
        late = ops.pending_sink_indices("opt").intersection(indices)
        if late:
            raise RuntimeError("bucket %d fired before parameters %s were joined" % (bucket, sorted(late)))

    gb.pre_fire_check = check
    for i in range(3):
        gb.expect(i)
    gb.ready(2)                                       # bucket 1 = {param 2}: complete, fires, nothing pending
    assert seen == [(1, [2])] and gb.issue_order == [1]
    ops._PENDING.append((None, FakeSink(1), []))      # parameter 1's weight gradient is still on the side stream
    try:
        gb.ready(0)
        with pytest.raises(RuntimeError, match="before parameters \\[1\\] were joined"):
            gb.ready(1)                               # would complete bucket 0 = {0, 1} and fire it
    finally:
        del ops._PENDING[:]


def _flag_worker(rank, world, port, out, overlap):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), SED_ALLREDUCE_OVERLAP="1" if overlap else "0")
    from sound_event_detection_dcase2017_task4_amd import parallel
    parallel.init_from_env(backend="gloo")
    numels = [4, 6, 10, 20]
    offsets = list(np.cumsum([0] + numels[:-1]))
    pad = 4
    store = torch.zeros(pad + sum(numels))
    flat = store[pad:]
    gb = parallel.GradBuckets(flat, offsets, numels, cuts=[offsets[2], offsets[3]], store=store, pad=pad)
    assert gb.deferred == (not overlap)
    published = []

    def publish():                                   # what optim.FusedAdamAmsgrad._publish_flag does on the device
        published.append(list(gb.issue_order))
        store[0] = float("nan") if (rank == 1 and len(published) == 2) else 0.0

    gb.publish_flag = publish
    for step in range(3):
        for i in range(4):
            gb.expect(i)
        gb.new_gradients()
        store.zero_()
        for i in (3, 2, 1, 0):
            flat[offsets[i]:offsets[i] + numels[i]] = float(rank + 1)
            gb.ready(i)
            if not overlap:
                assert gb.issue_order == []              # nothing leaves from inside "backward"
        gb.finish()
        assert gb.issue_order == [2, 1, 0]
        assert published[-1] == [2, 1, 0]                # published right before the LAST bucket (bucket 0) went out
        assert torch.all(flat == 3.0)
        # the flag of step 1 (rank 1 only) reaches BOTH ranks on the last bucket's all-reduce; steps 0 and 2 are clean
        assert bool(torch.isnan(store[0])) == (step == 1), (rank, step, store[:pad])
        gb.begin_step()
    dist.barrier()
    if rank == 0:
        open(out, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_rank_flag_rides_on_the_last_bucket(tmp_path, overlap):
    """The found-non-finite rank flag (store[0], in front of the flat gradient) is published right before the last bucket
    of a step is issued and summed with it: one rank's NaN flag is every rank's after the exchange.  With
    SED_ALLREDUCE_OVERLAP=0 no bucket leaves from inside backward; all go out from finish(), tail first."""
    out = str(tmp_path / "ok.txt")
    mp.spawn(_flag_worker, args=(2, _free_port(), out, overlap), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_bench_strong_row_watchdog_keeps_the_headline_line():
    """`bench.py --gpus N`: the strong-scaling row runs AFTER the headline line is complete, under a watchdog on every rank.  A
    row that never comes back (a stuck collective) must cost the row only: rank 0 prints the headline with `strong.error`, every
    rank leaves with exit code 0 (bench.StrongRowWatchdog)."""
    import json
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "line = {'metric': 'm', 'value': 1.0, 'n_gpus': 2}\n"
            "w = bench.StrongRowWatchdog(line if sys.argv[1] == '0' else None, 0.5)\n"
            "time.sleep(30)\n"
            "print('never')\n" % REPO)
    for rank in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code, rank], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "never" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if rank == "0":
            assert len(lines) == 1
            d = json.loads(lines[0])
            assert d["value"] == 1.0 and d["strong"]["value"] is None and "watchdog" in d["strong"]["error"]
        else:
            assert lines == []
    # cancelled in time: nothing happens
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "w = bench.StrongRowWatchdog({'a': 1}, 0.5); w.cancel(); time.sleep(1.5); print('done')\n" % REPO)
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "done"
