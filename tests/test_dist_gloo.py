"""world_size-2 test of the N>1 path on CPU (gloo): block ownership map, the
per-layer residual all-reduce and the single fused [mean | var] reduce.  The
per-block arithmetic is the oracle here (tests may use it as the checker); the
sharding and reduction code is the product's (cimrgp_amd.dist)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    import oracle
    rng = np.random.default_rng(5)
    n, ns = 256, 96
    x = np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x), np.cos(5 * x)]) + 0.1 * rng.normal(size=(n, 2))
    xs = np.sort(rng.uniform(-1.7, 1.7, size=(ns, 1)), axis=0)
    bounds = oracle.index_bounds_uniform(n, 2, 2)
    tbounds = oracle.index_bounds_uniform(ns, 2, 2)
    specs = [oracle.DenseLayerSpec(1.0 / 2 ** j, 1.0, None) for j in range(3)]
    return x, y, xs, bounds, tbounds, specs


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import oracle
    from cimrgp_amd import dist
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    x, y, xs, bounds, tbounds, specs = _problem()
    r, w = dist.world()
    assert (r, w) == (rank, world)
    f_bar = np.zeros_like(y)
    model = []
    for j, layer in enumerate(bounds):
        owner = dist.assign_blocks([b - a for a, b in layer], w)
        layer_pred = torch.zeros(y.shape, dtype=torch.float64)
        blocks = {}
        for l, (a, b) in enumerate(layer):
            if owner[l] != r:
                continue
            resid = (y - f_bar)[a:b]
            bias = resid.mean(axis=0)
            noise = oracle.mrgp._noise_from_targets(resid, specs[j])
            fit = oracle.block_fit(x[a:b], resid - bias, specs[j].ell, specs[j].sf2, noise)
            layer_pred[a:b] = torch.from_numpy(resid - bias - noise * fit["alpha"] + bias)
            blocks[l] = dict(a=a, b=b, bias=bias, noise=noise, alpha=fit["alpha"], L=fit["L"])
        dist.allreduce_sum_(layer_pred)                 # per-layer residual exchange
        f_bar = f_bar + layer_pred.numpy()
        model.append((owner, blocks))
    fused = torch.zeros((3, xs.shape[0]), dtype=torch.float64)
    for j, (owner, blocks) in enumerate(model):
        for l, blk in blocks.items():
            ta, tb = tbounds[j][l]
            m, v = oracle.block_predict(x[blk["a"]:blk["b"]], blk, xs[ta:tb], specs[j].ell, specs[j].sf2, True)
            fused[:2, ta:tb] += torch.from_numpy((m + blk["bias"]).T)
            fused[2, ta:tb] += torch.from_numpy(v + (blk["noise"] if j == len(model) - 1 else 0.0))
    dist.allreduce_sum_(fused)                          # ONE reduce for the sum over resolutions
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), f_bar=f_bar, fused=fused.numpy())
    td.destroy_process_group()


def test_two_rank_sharded_chain_matches_single_process(tmp_path):
    import oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x, y, xs, bounds, tbounds, specs = _problem()
    model, f_bar = oracle.mrgp_fit(x, y, bounds, specs)
    mean, var = oracle.mrgp_predict(x, model, specs, xs, tbounds, True, True)
    for rank in range(2):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        np.testing.assert_allclose(g["f_bar"], f_bar, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(g["fused"][:2].T, mean, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(g["fused"][2], var, rtol=1e-10, atol=1e-13)


def test_single_process_world_is_identity():
    from cimrgp_amd import dist
    assert dist.world() == (0, 1)
    t = torch.arange(4.0)
    assert dist.allreduce_sum_(t) is t


def _async_reduce_worker(rank, world, port, out_dir):
    """bench.py's pattern: every rank writes its slice of a zeroed fused buffer, the reduce is started and only
    waited for when the buffer is touched again one step later."""
    sys.path.insert(0, ROOT)
    from cimrgp_amd import dist
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ns = 5
    fused = torch.zeros((3, ns * world), dtype=torch.float64)
    pending = None
    seen = []
    for step in range(3):
        if pending is not None:
            pending.wait()
            seen.append(fused.clone())
        fused.zero_()
        fused[:, rank * ns:(rank + 1) * ns] = float(10 * step + rank + 1)
        pending = dist.allreduce_sum_begin(fused)
        assert pending is not None
    pending.wait()
    seen.append(fused.clone())
    np.save(os.path.join(out_dir, "async%d.npy" % rank), torch.stack(seen).numpy())
    td.destroy_process_group()


def test_async_reduce_of_the_fused_buffer_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_async_reduce_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "async%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(got[0], got[1])
    for step in range(3):
        for r in range(world):
            assert np.all(got[0][step][:, r * 5:(r + 1) * 5] == 10 * step + r + 1)   # no stale slices from earlier steps


def test_async_reduce_is_a_no_op_for_one_process():
    from cimrgp_amd import dist
    assert dist.allreduce_sum_begin(torch.ones(3)) is None


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from cimrgp_amd import dist
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    # phase 1: nobody fails -> nobody raises
    dist.raise_together(None)
    # phase 2: rank 1 fails in its local phase; rank 0 must not be left waiting in the next collective
    err = np.linalg.LinAlgError("Matrix is not positive definite (leading minor of order 3)") if rank == 1 else None
    try:
        dist.raise_together(err)
        seen = "none"
    except np.linalg.LinAlgError as exc:
        seen = "own:" + str(exc)
    except dist.RemoteRankError as exc:
        seen = "remote:" + str(exc)
    # both ranks are still in step: a collective after the failure completes
    t = torch.ones(1)
    dist.allreduce_sum_(t)
    with open(os.path.join(out_dir, "seen%d.txt" % rank), "w") as fh:
        fh.write("%s|%d" % (seen, int(t.item())))
    td.destroy_process_group()


def test_rank_local_failure_is_raised_on_every_rank(tmp_path):
    """a failure on one rank (e.g. a non-PD block) must surface on all of them instead of
    hanging the healthy ranks in the layer's all-reduce."""
    port = _free_port()
    mp.spawn(_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    seen = [open(os.path.join(str(tmp_path), "seen%d.txt" % r)).read() for r in range(2)]
    assert seen[0].startswith("remote:") and seen[0].endswith("|2")
    assert seen[1].startswith("own:Matrix is not positive definite") and seen[1].endswith("|2")


# ---------------------------------------------------------------------------------- nested ownership
def _nested_problem(world):
    import oracle
    rng = np.random.default_rng(9)
    n, res = (256, 2) if world == 2 else (512, 4)
    x = np.sort(rng.uniform(-1.7, 1.7, size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x), np.cos(5 * x)]) + 0.1 * rng.normal(size=(n, 2))
    bounds = oracle.index_bounds_uniform(n, res, 2)
    specs = [oracle.DenseLayerSpec(1.0 / 2 ** j, 1.0, None) for j in range(res + 1)]
    return x, y, bounds, specs


def _nested_worker(rank, world, port, out_dir):
    """MRGP._fit's communication schedule (cimrgp_amd/MRGP.py) with the oracle as the per-block arithmetic:
    layers below dist.plan_layers' `first_local` exchange per layer, the others not at all; ONE all-reduce after
    the sweep assembles the local layers' predictions."""
    sys.path.insert(0, ROOT)
    import oracle
    from cimrgp_amd import dist
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    x, y, bounds, specs = _nested_problem(world)
    owners, first_local = dist.plan_layers(bounds, world)
    f_bar = np.zeros_like(y)
    f_layers, tail, n_coll = [], [], 0
    for j, layer in enumerate(bounds):
        f_layers.append(f_bar.copy())
        layer_pred = torch.zeros(y.shape, dtype=torch.float64)
        for l, (a, b) in enumerate(layer):
            if owners[j][l] != rank:
                continue
            resid = (y - f_bar)[a:b]
            bias = resid.mean(axis=0)
            noise = oracle.mrgp._noise_from_targets(resid, specs[j])
            fit = oracle.block_fit(x[a:b], resid - bias, specs[j].ell, specs[j].sf2, noise)
            layer_pred[a:b] = torch.from_numpy(resid - noise * fit["alpha"])
        if j < first_local:
            dist.allreduce_sum_(layer_pred)
            n_coll += 1
        else:
            tail.append(layer_pred)
        f_bar = f_bar + layer_pred.numpy()
    if tail:
        stacked = torch.stack(tail)
        dist.allreduce_sum_(stacked)
        n_coll += 1
        f_bar = f_layers[first_local].copy()
        for k, j in enumerate(range(first_local, len(bounds))):
            f_layers[j] = f_bar.copy()
            f_bar = f_bar + stacked[k].numpy()
    np.savez(os.path.join(out_dir, "nested%d.npz" % rank), f_bar=f_bar, f_layers=np.stack(f_layers),
             owners=np.concatenate([np.asarray(o) for o in owners]), first_local=first_local, n_coll=n_coll)
    td.destroy_process_group()


def _check_nested(tmp_path, world, want_first_local, want_blocks_rank0):
    import oracle
    port = _free_port()
    mp.spawn(_nested_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x, y, bounds, specs = _nested_problem(world)
    _, f_bar = oracle.mrgp_fit(x, y, bounds, specs)
    got = [np.load(os.path.join(str(tmp_path), "nested%d.npz" % r)) for r in range(world)]
    for g in got:
        assert int(g["first_local"]) == want_first_local
        assert int(g["n_coll"]) == want_first_local + 1            # one per exchanged layer + ONE for all the local ones
        np.testing.assert_array_equal(g["owners"], got[0]["owners"])          # every rank computes the same plan
        np.testing.assert_allclose(g["f_bar"], f_bar, rtol=1e-12, atol=1e-13)    # = the single-process chain
        np.testing.assert_array_equal(g["f_layers"], got[0]["f_layers"])      # every layer's latent function, everywhere
    from cimrgp_amd import dist
    owners, _ = dist.plan_layers(bounds, world)
    assert [int((np.asarray(o) == 0).sum()) for o in owners] == want_blocks_rank0


def test_nested_ownership_two_ranks_skips_the_per_layer_exchange(tmp_path):
    _check_nested(tmp_path, 2, 1, [1, 1, 2])


def test_nested_ownership_eight_ranks(tmp_path):
    _check_nested(tmp_path, 8, 3, [1, 1, 1, 1, 2])


def test_plan_layers_blocks_per_rank_of_the_baseline_configs():
    """BASELINE configs[2] (N = 65536, the reference's index set: 1, 2, 4, 8, 16 regions) and configs[3] (N = 262144,
    root policy: 8 ... 128 regions) on 1, 2, 4 and 8 ranks: blocks per rank and layer, the first local layer, and that
    every finer block sits on the rank owning the range around it (src/IndexSetGenerator.py:51-65 nests the ranges)."""
    from cimrgp_amd import dist
    from cimrgp_amd.IndexSetGenerator import IndexSetUniform
    idx3 = IndexSetUniform(65536, 4, 2)
    idx4 = IndexSetUniform(262144, 4, 2, first_divider_power=3)
    for world, first3 in [(1, 5), (2, 1), (4, 2), (8, 3)]:
        owners, first = dist.plan_layers(idx3.bounds, world)
        assert first == first3
        per_rank = [[int((np.asarray(o) == r).sum()) for o in owners] for r in range(world)]
        if world == 8:
            assert per_rank[0] == [1, 1, 1, 1, 2] and per_rank[1] == [0, 1, 1, 1, 2] and per_rank[4] == [0, 0, 0, 1, 2]
        for j in range(first, 4):                       # nesting: a block's children live on its rank
            for l, (a, b) in enumerate(idx3.bounds[j + 1]):
                parent = next(k for k, (pa, pb) in enumerate(idx3.bounds[j]) if pa <= a and b <= pb)
                assert owners[j + 1][l] == owners[j][parent]
    for world in (1, 2, 4, 8):
        owners, first = dist.plan_layers(idx4.bounds, world)
        assert first == (5 if world == 1 else 0)
        per_rank = [[int((np.asarray(o) == r).sum()) for o in owners] for r in range(world)]
        assert all(p == [8 // world, 16 // world, 32 // world, 64 // world, 128 // world] for p in per_rank)
    # a sample count the divider does not divide: the ranges do not nest -> every layer by LPT, every layer exchanged
    owners, first = dist.plan_layers(IndexSetUniform(1003, 2, 2).bounds, 2)
    assert first == 3
