"""Two ranks sharing cuda:0: the multi-GPU drivers (sample-range Monte-Carlo with LCG skip-ahead, the one-call
stochastic_evolution form, row-sharded matmul, block-column cyclic LU with panel broadcasts and depth-1 look-ahead, the
row-partitioned solve with diagonal-domain pivoting)
through the REAL provider, checked against the oracle.  Run twice: with the data path on torch.distributed (gloo,
host-staged) and with every exchange going through the C-ABI collectives (`rmhip_comm_*`) on the host shared-memory
transport - RCCL refuses two ranks on one device, so on this single-GPU box RCCL itself is exercised with a one-rank
communicator; on a multi-GPU node the same entry points run over RCCL / xGMI, one rank per GPU."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, native):
    import torch.distributed as dist

    from runmat_amd import HipProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = HipProvider(0)
        if native:  # control plane stays gloo (it carries the 128-byte id); the data path is rmhip_comm_*
            group.with_native_comm(prov, transport="shm")
            assert prov.comm_rank() == (rank, world)
        from planner_requests import monte_carlo_shaders

        shaders = monte_carlo_shaders(100.0)
        seed = 0x9E3779B97F4A7C15
        M, T = 200001, 3  # odd M: the last pair is half used
        p_fused, s_fused = sh.monte_carlo_price_fused(prov, group, M, T, shaders, rng_state=seed)
        p_evol, s_evol = sh.monte_carlo_price_evolved(prov, group, M, T, rng_state=seed, payoff_shader=shaders[1])
        # row-sharded matmul: this rank's rows of A from the global generator, B replicated
        m, k, n = 384, 96, 160
        rng = np.random.default_rng(11)
        A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
        r0, r1 = sh.row_block(m, group)
        hc = sh.matmul_row_sharded(prov, prov.upload(A[r0:r1, :]), prov.upload(B))
        if native:
            full, _ = sh.gather_row_blocks_device(group, prov, hc, m)  # rmhip_comm_allgather_rows
            C = prov.download_matrix(full)
        else:
            C = sh.gather_row_blocks(group, prov.download_matrix(hc), m)
        # block-column cyclic LU: every rank builds the same A and keeps the column blocks it owns
        nn, nb, nrhs = 1000, 128, 2
        rng2 = np.random.default_rng(12)
        AA = rng2.standard_normal((nn, nn))
        BB = rng2.standard_normal((nn, nrhs))
        blocks = sh.owned_blocks(nn, nb, group)
        cols = [c for p in blocks for c in range(p * nb, min(nn, (p + 1) * nb))]
        x = prov.download_matrix(sh.mldivide_block_cyclic(prov, group, prov.upload(AA[:, cols]), nn, prov.upload(BB), nb=nb))
        # row-partitioned solve: every rank keeps the row blocks of [A | b] it owns; diagonal-domain pivoting, one tile-row broadcast
        # per panel, the last `world` blocks gathered (sharding.mldivide_row_partitioned).  Also: a matrix whose large entries sit
        # outside the owners' domains must be refused (PivotGrowth), on every rank alike.
        rbk = 128
        rows = [r for q in sh.owned_row_blocks(nn, rbk, group) for r in range(q * rbk, min(nn, (q + 1) * rbk))]
        xr = prov.download_matrix(sh.mldivide_row_partitioned(prov, group, prov.upload(np.hstack([AA, BB])[rows, :]), nn, nrhs, rb=rbk))
        bad = (rng2.uniform(-1, 1, (nn, nn)) + nn * np.eye(nn))[np.roll(np.arange(nn), 3 * rbk)]  # the diagonal moved three blocks down
        refused = False
        try:
            sh.mldivide_row_partitioned(prov, group, prov.upload(np.hstack([bad, BB])[rows, :]), nn, nrhs, rb=rbk)
        except sh.PivotGrowth:
            refused = True
        # the same two sharded forms through the C-ABI drivers (rmhip_matmul_row_sharded, rmhip_mldivide_row_partitioned: csrc/sharded.cpp),
        # what a host that is not Python calls - they need the native communicator
        xc = cc = None
        refused_c = collective_c = False
        if native:
            from runmat_amd import ProviderError

            cc = prov.download_matrix(prov.matmul_row_sharded(prov.upload(A[r0:r1, :]), prov.upload(B), m, gather=True))
            xc = prov.download_matrix(prov.mldivide_row_partitioned(prov.upload(np.hstack([AA, BB])[rows, :]), nn, nrhs, rb=rbk))
            try:
                prov.mldivide_row_partitioned(prov.upload(np.hstack([bad, BB])[rows, :]), nn, nrhs, rb=rbk)
            except ProviderError as e:
                refused_c = e.code == 10  # RMHIP_ERR_GROWTH, on every rank
            # a singular diagonal domain on rank 0 only: nobody hangs, everybody gets RMHIP_ERR_GROWTH
            sing = AA.copy()
            own0 = [r for q in range(0, (nn + rbk - 1) // rbk, world) for r in range(q * rbk, min(nn, (q + 1) * rbk))]
            sing[own0, 5] = 0.0
            try:
                prov.mldivide_row_partitioned(prov.upload(np.hstack([sing, BB])[rows, :]), nn, nrhs, rb=rbk)
            except ProviderError as e:
                collective_c = e.code == 10
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p_fused=p_fused, s_fused=np.uint64(s_fused), p_evol=p_evol,
                 s_evol=np.uint64(s_evol), C=C, x=x, xr=xr, refused=refused, xc=xc if xc is not None else np.zeros(0),
                 cc=cc if cc is not None else np.zeros(0), refused_c=refused_c, collective_c=collective_c)
        prov.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("native", [False, True], ids=["torch-gloo", "c-abi-collectives"])
def test_two_ranks_on_one_gpu(oracle, tmp_path, native):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), native), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    want, want_state = oracle.monte_carlo_price(oracle.rng_default_seed(), 200001, 3)
    for r in res:
        assert int(r["s_fused"]) == want_state and int(r["s_evol"]) == want_state  # integer stream position: exact
        assert abs(float(r["p_fused"]) - want) <= 1e-10 * want and abs(float(r["p_evol"]) - want) <= 1e-10 * want
    assert res[0]["p_fused"] == res[1]["p_fused"] and res[0]["p_evol"] == res[1]["p_evol"]  # ordered sums
    rng = np.random.default_rng(11)
    A, B = rng.uniform(-1, 1, (384, 96)), rng.uniform(-1, 1, (96, 160))
    ref = oracle.matmul(A, B)
    for r in res:
        assert np.max(np.abs(r["C"] - ref)) <= 98 * 2.3e-16 * np.max(np.abs(A) @ np.abs(B))
    assert np.array_equal(res[0]["C"], res[1]["C"])
    rng2 = np.random.default_rng(12)
    AA = rng2.standard_normal((1000, 1000))
    BB = rng2.standard_normal((1000, 2))
    xr = oracle.mldivide_lu(AA, BB)
    for r in res:
        assert np.max(np.abs(r["x"] - xr)) <= 1e-9 * max(1.0, np.abs(xr).max())
        assert np.max(np.abs(r["xr"] - xr)) <= 1e-9 * max(1.0, np.abs(xr).max())  # row-partitioned form vs the oracle's LU solve
        assert bool(r["refused"])
    assert np.array_equal(res[0]["x"], res[1]["x"])
    assert np.array_equal(res[0]["xr"], res[1]["xr"])  # replicated, bit-identical
    if native:  # the C-ABI drivers beside the Python ones
        for r in res:
            assert np.array_equal(r["cc"], r["C"])  # same kernel on the same rows, same gather: identical bits
            assert np.max(np.abs(r["xc"] - xr)) <= 1e-9 * max(1.0, np.abs(xr).max())
            # the look-ahead splits each trailing update into column / row pieces (same products, possibly another tile kernel):
            # agreement with the Python driver to rounding, identity where the kernels coincide
            assert np.max(np.abs(r["xc"] - r["xr"])) <= 1e-11 * max(1.0, np.abs(xr).max())
            assert bool(r["refused_c"]) and bool(r["collective_c"])
        assert np.array_equal(res[0]["xc"], res[1]["xc"])  # replicated, bit-identical across ranks


def _abort_worker(rank, world, port, out_dir):
    import time

    import torch.distributed as dist

    from runmat_amd import HipProvider, ProviderError
    from runmat_amd import sharding as sh

    os.environ["RMHIP_RP_TEST_FAIL_RANK"] = "1"  # rank 1 leaves the driver before its first collective (a local failure)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = HipProvider(0)
        group.with_native_comm(prov, transport="shm")
        nn, nrhs, rbk = 1000, 2, 128
        rng = np.random.default_rng(12)
        AB = np.hstack([rng.standard_normal((nn, nn)), rng.standard_normal((nn, nrhs))])
        rows = [r for q in sh.owned_row_blocks(nn, rbk, group) for r in range(q * rbk, min(nn, (q + 1) * rbk))]
        t0 = time.perf_counter()
        code, msg = 0, ""
        try:
            prov.mldivide_row_partitioned(prov.upload(AB[rows, :]), nn, nrhs, rb=rbk)
        except ProviderError as e:
            code, msg = e.code, str(e)
        dt = time.perf_counter() - t0
        again = ""
        try:  # the communicator stays unusable until it is destroyed and made again
            prov.comm_barrier()
        except ProviderError as e:
            again = str(e)
        prov.comm_destroy()
        np.savez(os.path.join(out_dir, f"abort{rank}.npz"), code=code, msg=msg, dt=dt, again=again)
        prov.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_local_failure_aborts_the_communicator_instead_of_hanging_the_peer(tmp_path):
    """a rank that leaves rmhip_mldivide_row_partitioned between two collectives (allocation / device failure; here the
    test hook) must not leave the others blocked in the next broadcast: it aborts the communicator, the peer's barrier fails at once."""
    import torch.multiprocessing as mp

    mp.spawn(_abort_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"abort{r}.npz") for r in range(2))
    assert int(r1["code"]) != 0 and "injected local failure" in str(r1["msg"])
    assert int(r0["code"]) != 0 and "aborted the communicator" in str(r0["msg"]), str(r0["msg"])
    assert float(r0["dt"]) < 20.0, float(r0["dt"])  # not the 60 s barrier timeout
    assert "abort" in str(r0["again"]) and "abort" in str(r1["again"])


def _expire_worker(rank, world, port, out_dir):
    import time

    import torch.distributed as dist

    from runmat_amd import HipProvider, ProviderError
    from runmat_amd import sharding as sh

    if rank == 1:
        os.environ["RMHIP_COMM_TEST_EXPIRE"] = "1"  # rank 1's bounded wait behaves as if its time had run out
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = HipProvider(0)
        group.with_native_comm(prov, transport="shm")
        h = prov.upload(np.arange(6.0).reshape(3, 2) + rank)
        prov.comm_bcast(h, 0)
        t0 = time.perf_counter()
        first, second = "", ""
        try:
            prov.comm_wait_bounded(5.0)
        except ProviderError as e:
            first = str(e)
        dist.barrier()  # rank 1 has aborted by now
        try:
            prov.comm_barrier()
        except ProviderError as e:
            second = str(e)
        dt = time.perf_counter() - t0
        prov.comm_destroy()
        np.savez(os.path.join(out_dir, f"expire{rank}.npz"), first=first, second=second, dt=dt)
        prov.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bounded_wait_aborts_the_communicator_on_expiry(tmp_path):
    """on expiry (here a test hook on rank 1) rmhip_comm_wait_bounded aborts the local communicator and reports an error;
    the peer's next collective fails at once instead of blocking (host transport: the shared abort flag; on RCCL the local
    ncclCommAbort lets this rank's own collective kernels exit, and the peers' bounded waits expire in turn)."""
    import torch.multiprocessing as mp

    mp.spawn(_expire_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"expire{r}.npz") for r in range(2))
    assert "timed out" in str(r1["first"]) and str(r0["first"]) == ""
    assert "abort" in str(r0["second"]) and "abort" in str(r1["second"])
    assert float(r0["dt"]) < 20.0 and float(r1["dt"]) < 20.0


def test_rccl_one_rank_communicator(prov, oracle):
    """The RCCL transport itself (librccl through dlopen, ncclCommInitRank / ncclAllGather / ncclBroadcast on the
    context's streams) with the one-rank communicator a single-GPU box allows, plus the sharded drivers on it."""
    from runmat_amd import HipProvider
    from runmat_amd import sharding as sh

    p2 = HipProvider(0)
    try:
        g = sh.Group().with_native_comm(p2, transport="rccl")
        assert p2.comm_rank() == (0, 1)
        rng = np.random.default_rng(4)
        X = rng.uniform(-1, 1, (300, 7))
        h = p2.upload(X)
        full = p2.comm_allgather_rows(h, 300, 128)
        assert np.array_equal(p2.download_matrix(full), X)
        v = p2.upload(np.array([[1.5], [-2.0], [3.25]]))
        gv = p2.comm_allgather_f64(v)
        assert gv.shape == (3, 1) and np.array_equal(p2.download(gv), [1.5, -2.0, 3.25])
        p2.comm_bcast(h, 0)                      # whole buffer
        p2.comm_bcast((h, 10, 2, 50, 3), 0)      # strided sub-block (packed)
        p2.comm_bcast(h, 0, async_=True)         # communication stream
        p2.comm_wait()
        p2.comm_barrier()
        assert np.array_equal(p2.download_matrix(h), X)
        want, want_state = oracle.monte_carlo_price(oracle.rng_default_seed(), 50001, 2)
        from planner_requests import monte_carlo_shaders

        price, state = sh.monte_carlo_price_fused(p2, g, 50001, 2, monte_carlo_shaders(100.0), rng_state=oracle.rng_default_seed())
        assert state == want_state and abs(price - want) <= 1e-10 * want
        n, nb = 700, 128
        A = rng.standard_normal((n, n))
        B = rng.standard_normal((n, 2))
        x = p2.download_matrix(sh.mldivide_block_cyclic(p2, g, p2.upload(A), n, p2.upload(B), nb=nb))
        assert np.max(np.abs(x - oracle.mldivide_lu(A, B))) <= 1e-9 * max(1.0, np.abs(oracle.mldivide_lu(A, B)).max())
        p2.comm_destroy()
    finally:
        p2.close()
