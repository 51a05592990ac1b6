"""CPU, world_size 2, gloo: the N>1 path -- pair sharding, gradient all-reduce(SUM) + identical SGD on every rank, MAX timing."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lib.utils.dist_utils import shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lib.utils.dist_utils import allreduce_sum_, barrier, max_over_ranks
    from oracle import train as otrain

    rng = np.random.RandomState(100 + rank)
    shapes = {"conv_weight": (8, 4, 3, 3), "conv_bias": (8,), "upsampling_weight": (2, 1, 4, 4)}
    params = {k: np.random.RandomState(7).randn(*s).astype(np.float32) for k, s in shapes.items()}  # same init on every rank
    grads = {k: rng.randn(*s) for k, s in shapes.items()}                                            # rank-local gradients
    flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in shapes]))
    allreduce_sum_(flat)
    off, summed = 0, {}
    for k, s in shapes.items():
        n = int(np.prod(s))
        summed[k] = flat[off:off + n].numpy().reshape(s)
        off += n
    moms = {k: np.zeros(s) for k, s in shapes.items()}
    params, moms = otrain.sgd_step(params, summed, moms, lr=1e-2, momentum=0.975, wd=5e-4)
    t = max_over_ranks(1.0 + rank)
    barrier()
    np.savez(os.path.join(out_dir, "rank{}.npz".format(rank)), tmax=t, **{"g_" + k: grads[k] for k in shapes}, **{"s_" + k: summed[k] for k in shapes},
             **{"p_" + k: params[k] for k in shapes})
    dist.destroy_process_group()


def test_allreduce_sum_sgd_and_timing_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank{}.npz".format(i))) for i in range(world)]
    for k in ("conv_weight", "conv_bias", "upsampling_weight"):
        np.testing.assert_allclose(r[0]["s_" + k], r[0]["g_" + k] + r[1]["g_" + k], rtol=1e-12)  # SUM, not mean
        np.testing.assert_array_equal(r[0]["s_" + k], r[1]["s_" + k])
        np.testing.assert_array_equal(r[0]["p_" + k], r[1]["p_" + k])                              # replicas stay identical
    # every rank draws from its own RandomState(7): identical initial values; the frozen (lr_mult 0) tensor must not move
    np.testing.assert_array_equal(r[0]["p_upsampling_weight"], np.random.RandomState(7).randn(2, 1, 4, 4).astype(np.float32))
    assert float(r[0]["tmax"]) == 2.0 and float(r[1]["tmax"]) == 2.0


def test_shard_range_partitions_pairs():
    for total in (1, 7, 16, 128, 129):
        for world in (1, 2, 4, 8):
            cover = []
            for rank in range(world):
                b, e = shard_range(total, rank, world)
                assert 0 <= b <= e <= total and (e - b) in (total // world, total // world + 1)
                cover += list(range(b, e))
            assert cover == list(range(total))


def _uneven_worker(rank, world, port, n_batches, out_dir):
    import datetime

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from lib.utils.dist_utils import allreduce_sum_, even_shard_range

    lo, hi = even_shard_range(n_batches, rank, world)
    acc = torch.zeros(4, dtype=torch.float64)
    for i in range(lo, hi):           # one blocking collective per optimizer step, like MutableModule.update
        g = torch.full((4,), float(i), dtype=torch.float64)
        allreduce_sum_(g)
        acc += g
    np.save(os.path.join(out_dir, "acc{}.npy".format(rank)), np.append(acc.numpy(), hi - lo))
    dist.destroy_process_group()


def test_training_shards_are_equal_when_batches_do_not_divide_world3(tmp_path):
    """16 batches on 3 ranks: shard_range would give 6/5/5 and the rank with the extra batch would block in an all-reduce the
    others never post; the training split gives 5/5/5 (remainder dropped) and every rank runs the same number of collectives."""
    from lib.utils.dist_utils import even_shard_range

    world, n_batches, port = 3, 16, _free_port()
    assert sorted(shard_range(n_batches, r, world)[1] - shard_range(n_batches, r, world)[0] for r in range(world)) == [5, 5, 6]
    spans = [even_shard_range(n_batches, r, world) for r in range(world)]
    assert [e - b for b, e in spans] == [5, 5, 5] and spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    mp.spawn(_uneven_worker, args=(world, port, n_batches, str(tmp_path)), nprocs=world, join=True)
    accs = [np.load(os.path.join(str(tmp_path), "acc{}.npy".format(r))) for r in range(world)]
    for a in accs:
        np.testing.assert_array_equal(a, accs[0])
    assert accs[0][4] == 5 and accs[0][0] == sum(range(15))


def test_bench_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` is ONE command (reference: `--gpus 0,1,2,3`, deepim/train.py:425-438): with WORLD_SIZE unset it starts a
    torch.distributed.run job of 2 ranks as a child process.  No GPU here, so the ranks meet over gloo, say so, and then every rank
    stops at the GPU check -- the job, and with it the parent, must leave with a non-zero code (never a silent 1-GPU run)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert "rank 0 of 2 joined (gloo)" in r.stdout and "rank 1 of 2 joined (gloo)" in r.stdout, r.stdout[-3000:]
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stdout, r.stdout[-3000:]
    # an external launcher whose world size contradicts --gpus is an error, also when that world size is 1
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env1, stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert r1.returncode != 0 and "WORLD_SIZE is 1" in r1.stdout, r1.stdout[-2000:]


def _ident_worker(rank, world, port, out_dir):
    import json

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lib.utils.dist_utils import gather_floats, gather_rank_identities

    # what rank_identity() returns on a GPU box, with a fake card per rank (no GPU here); ranks 0 and 1 share a uuid
    ident = {"rank": rank, "local_rank": rank, "host": "node0", "device": rank, "name": "AMD Instinct MI355X", "arch": "gfx950", "pci_bus_id": None,
             "uuid": "GPU-{:02d}".format(min(rank, 1) if world == 3 else rank), "cus": 256, "hbm_gb": 288.0, "pid": os.getpid()}
    res = {"floats": gather_floats(1.5 + rank)}
    for backend in ("gloo", "nccl"):
        try:
            parts = gather_rank_identities(ident, backend=backend)
            res[backend] = [p["rank"] for p in parts]
        except RuntimeError as e:
            res[backend] = "error: {}".format(e)
    with open(os.path.join(out_dir, "ident{}.json".format(rank)), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_rank_identities_are_gathered_and_a_shared_card_is_an_error_under_rccl(tmp_path):
    """the `ranks` object of the bench line (which devices joined): gathered in rank order on every rank; two ranks on one card are
    refused when the backend is nccl (= RCCL) and accepted in a gloo rehearsal; the per-rank step times travel the same way"""
    import json

    for world in (2, 3):
        d = tmp_path / "w{}".format(world)
        d.mkdir()
        mp.spawn(_ident_worker, args=(world, _free_port(), str(d)), nprocs=world, join=True)
        for rank in range(world):
            res = json.load(open(str(d / "ident{}.json".format(rank))))
            assert res["floats"] == [1.5 + r for r in range(world)]
            assert res["gloo"] == list(range(world))
            if world == 2:   # distinct cards
                assert res["nccl"] == [0, 1]
            else:            # ranks 1 and 2 report the same uuid
                assert res["nccl"].startswith("error: ranks 1 and 2 drive the same device GPU-01 on node0")
