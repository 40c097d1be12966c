"""Host logic of the multi-GPU path on CPU: partitioning, LCG skip-ahead, and world_size-2 runs of
the sharded matmul / Monte-Carlo over gloo with an oracle-backed provider double."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def test_partition_covers_exactly_once():
    from runmat_amd.sharding import partition

    for total in (0, 1, 7, 128, 1000, 8192, 100_000_001):
        for world in (1, 2, 3, 8):
            for gran in (1, 2, 128):
                spans = [partition(total, world, r, gran) for r in range(world)]
                assert spans[0][0] == 0 and spans[-1][1] == total
                for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                    assert a1 == b0 and a0 <= a1
                for s0, s1 in spans:
                    if s1 > s0:  # non-empty spans start on a granule boundary
                        assert s0 % gran == 0 and (s1 % gran == 0 or s1 == total)
                sizes = [s1 - s0 for s0, s1 in spans]
                assert max(sizes) - min(sizes) < 2 * gran or total < world * gran  # one granule + ragged tail


def test_lcg_advance_matches_oracle(oracle):
    from runmat_amd.sharding import lcg_advance

    s0 = oracle.rng_default_seed()
    for d in (0, 1, 2, 63, 64, 1000, 2**33 + 5, 2**63 + 11):
        assert lcg_advance(s0, d) == oracle.rng_advance(s0, d)
    _, s = oracle.rng_uniform(s0, 1234)
    assert lcg_advance(s0, 1234) == s


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from oracle import oracle
    from oracle_provider import OracleProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        assert group.rank == rank and group.world == world and group.device == "cpu"
        prov = OracleProvider(oracle)
        # --- row-sharded matmul: each rank builds ITS rows of A from the global generator ---
        m, k, n = 384, 96, 160
        A = oracle.fill_uniform(11, -1.0, 1.0, m * k).reshape(m, k, order="F")
        B = oracle.fill_uniform(12, -1.0, 1.0, k * n).reshape(k, n, order="F")
        r0, r1 = sh.row_block(m, group)
        c_rows = sh.matmul_row_sharded(prov, prov.upload(A[r0:r1, :]), prov.upload(B))
        C = sh.gather_row_blocks(group, c_rows.arr, m)
        # --- Monte-Carlo with skip-ahead ---
        M, T = 20001, 3  # odd M: the last pair is half used
        price, state = sh.monte_carlo_price_sharded(prov, group, M, T, rng_state=oracle.rng_default_seed())
        price_ev, state_ev = sh.monte_carlo_price_evolved(prov, group, M, T, rng_state=oracle.rng_default_seed())
        # --- ordered sum ---
        total = group.ordered_sum(0.1 * (rank + 1))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), C=C, price=price, state=np.uint64(state), total=total,
                 r0=r0, r1=r1, price_ev=price_ev, state_ev=np.uint64(state_ev))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo_sharded_matmul_and_monte_carlo(oracle, tmp_path):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    m, k, n = 384, 96, 160
    A = oracle.fill_uniform(11, -1.0, 1.0, m * k).reshape(m, k, order="F")
    B = oracle.fill_uniform(12, -1.0, 1.0, k * n).reshape(k, n, order="F")
    ref = oracle.matmul(A, B)
    assert (res[0]["r0"], res[0]["r1"], res[1]["r0"], res[1]["r1"]) == (0, 256, 256, 384)  # 128-row granules
    for r in res:
        assert np.array_equal(r["C"], ref)  # row blocks are computed exactly as the unsharded rows
    # Monte-Carlo: same stream as one device (skip-ahead), same price up to the summation grouping
    want, want_state = oracle.monte_carlo_price(oracle.rng_default_seed(), 20001, 3)
    for r in res:
        assert abs(float(r["price"]) - want) <= 1e-12 * want
        assert int(r["state"]) == want_state
        # the one-call time loop (stochastic_evolution with the global per-step stride) walks the same stream
        assert float(r["price_ev"]) == float(r["price"]) and int(r["state_ev"]) == want_state
    assert res[0]["price"] == res[1]["price"]  # ordered sum: bit-identical on every rank
    assert res[0]["total"] == res[1]["total"] == 0.1 + 0.2


def test_single_process_group_is_identity(oracle):
    from oracle_provider import OracleProvider
    from runmat_amd import sharding as sh

    g = sh.Group()
    assert g.ordered_sum(3.5) == 3.5 and sh.row_block(1000, g) == (0, 1000)
    price, state = sh.monte_carlo_price_sharded(OracleProvider(oracle), g, 5000, 2, rng_state=oracle.rng_default_seed())
    want, want_state = oracle.monte_carlo_price(oracle.rng_default_seed(), 5000, 2)
    assert price == want and state == want_state


def _lu_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from numpy_block_provider import NumpyBlockProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = NumpyBlockProvider()
        n, nb, nrhs = 150, 32, 2  # 5 blocks (the last one ragged) over 2 ranks
        rng = np.random.default_rng(5)
        A = rng.uniform(-1, 1, (n, n)) + 0.5 * np.eye(n)
        X = np.stack([np.ones(n), np.arange(n) / n], axis=1)
        B = A @ X
        cols = np.concatenate([np.arange(p * nb, min((p + 1) * nb, n)) for p in sh.owned_blocks(n, nb, group)])
        a_local = prov.upload(A[:, cols])
        y = sh.mldivide_block_cyclic(prov, group, a_local, n, prov.upload(B), nb=nb)
        np.savez(os.path.join(out_dir, f"lu_rank{rank}.npz"), x=prov.download(y).reshape(n, nrhs, order="F"), X=X)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo_block_cyclic_solve(tmp_path):
    """Host logic of the distributed A\\b (ownership, offsets, broadcast order, back substitution)."""
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_lu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"lu_rank{r}.npz") for r in range(world)]
    for r in res:
        assert np.max(np.abs(r["x"] - r["X"])) < 1e-9
    assert np.array_equal(res[0]["x"], res[1]["x"])  # replicated result, bit-identical on every rank


def test_block_cyclic_solve_single_rank_double():
    from numpy_block_provider import NumpyBlockProvider
    from runmat_amd import sharding as sh

    prov, g = NumpyBlockProvider(), sh.Group()
    n = 70
    rng = np.random.default_rng(6)
    A = rng.standard_normal((n, n))
    b = rng.standard_normal((n, 1))
    y = sh.mldivide_block_cyclic(prov, g, prov.upload(A.copy()), n, prov.upload(b), nb=16)
    assert np.allclose(prov.download(y), np.linalg.solve(A, b).reshape(-1), atol=1e-10)
    assert sh.owned_blocks(100, 32, sh.Group(1, 3)) == [1] and sh.local_col_offset(5, 32, sh.Group(1, 2)) == 64


def _rows_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from numpy_block_provider import NumpyBlockProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = NumpyBlockProvider()
        n, rb, nrhs = 150, 32, 2  # 5 row blocks (the last one ragged) over 2 ranks: 3 direct panels, 2 gathered blocks
        rng = np.random.default_rng(5)
        A = rng.uniform(-1, 1, (n, n))
        X = np.stack([np.ones(n), np.arange(n) / n], axis=1)
        AB = np.hstack([A, A @ X])
        rows = np.concatenate([np.arange(q * rb, min((q + 1) * rb, n)) for q in sh.owned_row_blocks(n, rb, group)])
        x = sh.mldivide_row_partitioned(prov, group, prov.upload(AB[rows, :]), n, nrhs, rb=rb, tau=64.0)
        # rows shuffled so that the big entries are NOT in the owners' domains: every rank must refuse, at the same point
        bad = (rng.uniform(-1, 1, (n, n)) + n * np.eye(n))[np.roll(np.arange(n), 2 * rb + 1)]
        refused = False
        try:
            sh.mldivide_row_partitioned(prov, group, prov.upload(np.hstack([bad, A @ X])[rows, :]), n, nrhs, rb=rb)
        except sh.PivotGrowth:
            refused = True
        # a singular diagonal domain on ONE rank (column 3 is zero on every row rank 0 owns: panel 0 finds no pivot there): the owner
        # must not leave the others blocked in the broadcast - it sends a poisoned tile, stays in step, and every rank raises after
        # the guard's exchange
        sing = A.copy()
        own0 = np.concatenate([np.arange(q * rb, min((q + 1) * rb, n)) for q in range(0, (n + rb - 1) // rb, world)])
        sing[own0, 3] = 0.0
        collective = False
        try:
            sh.mldivide_row_partitioned(prov, group, prov.upload(np.hstack([sing, A @ X])[rows, :]), n, nrhs, rb=rb, tau=64.0)
        except sh.PivotGrowth:
            collective = True
        np.savez(os.path.join(out_dir, f"rows_rank{rank}.npz"), x=prov.download(x).reshape(n, nrhs, order="F"), X=X, refused=refused,
                 collective=collective)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world2_gloo_row_partitioned_solve(tmp_path):
    """Host logic of the row-partitioned A\\b (row ownership, the per-panel tile-row broadcast, the gathered tail, the guard)."""
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_rows_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rows_rank{r}.npz") for r in range(world)]
    for r in res:
        assert np.max(np.abs(r["x"] - r["X"])) < 1e-9 and bool(r["refused"])
        assert bool(r["collective"])  # the failure of one rank's domain reached every rank (no hang: the test has a timeout)
    assert np.array_equal(res[0]["x"], res[1]["x"])


def _rows8_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from numpy_block_provider import NumpyBlockProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = NumpyBlockProvider()
        n, rb, nrhs = 300, 16, 1  # 19 row blocks (the last one ragged) over 8 ranks: 11 direct panels, 8 gathered blocks
        rng = np.random.default_rng(8)
        A = rng.uniform(-1, 1, (n, n)) + 4.0 * np.eye(n)
        X = np.ones((n, 1))
        AB = np.hstack([A, A @ X])
        rows = np.concatenate([np.arange(q * rb, min((q + 1) * rb, n)) for q in sh.owned_row_blocks(n, rb, group)])
        x = sh.mldivide_row_partitioned(prov, group, prov.upload(AB[rows, :]), n, nrhs, rb=rb, tau=64.0)
        nb = 16
        mine = sh.owned_blocks(n, nb, group)
        cols = np.concatenate([np.arange(p * nb, min((p + 1) * nb, n)) for p in mine])
        y = sh.mldivide_block_cyclic(prov, group, prov.upload(A[:, cols]), n, prov.upload(A @ X), nb=nb)
        np.savez(os.path.join(out_dir, f"rows8_rank{rank}.npz"), x=prov.download(x), y=prov.download(y))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world8_gloo_solvers_block_ownership(tmp_path):
    """The target machine has 8 GPUs: block ownership, the gathered tail (world row blocks) and the panel broadcasts of both sharded
    solvers with world = 8 (gloo, numpy block double)."""
    import torch.multiprocessing as mp

    world, port = 8, _free_port()
    mp.spawn(_rows8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rows8_rank{r}.npz") for r in range(world)]
    for r in res:
        assert np.max(np.abs(r["x"] - 1.0)) < 1e-9 and np.max(np.abs(r["y"] - 1.0)) < 1e-9
        assert np.array_equal(r["x"], res[0]["x"]) and np.array_equal(r["y"], res[0]["y"])


def test_row_partitioned_solve_single_rank_double():
    from numpy_block_provider import NumpyBlockProvider
    from runmat_amd import sharding as sh

    prov, g = NumpyBlockProvider(), sh.Group()
    for n, rb, nrhs in ((70, 16, 1), (64, 16, 2), (33, 64, 1)):
        rng = np.random.default_rng(n)
        A = rng.standard_normal((n, n))
        B = rng.standard_normal((n, nrhs))
        x = sh.mldivide_row_partitioned(prov, g, prov.upload(np.hstack([A, B])), n, nrhs, rb=rb)
        assert np.allclose(prov.download(x).reshape(n, nrhs, order="F"), np.linalg.solve(A, B), atol=1e-10)
    assert sh.owned_row_blocks(100, 32, sh.Group(1, 3)) == [1] and sh.local_row_offset(5, 32, sh.Group(1, 2)) == 64

