"""Two ranks sharing cuda:0 over gloo: the multi-GPU drivers (sample-range Monte-Carlo with LCG skip-ahead, the
one-call stochastic_evolution form, row-sharded matmul, block-column cyclic LU with panel broadcasts) through the
REAL provider, checked against the oracle.  The data-path collectives are host-staged here (gloo); on a multi-GPU
node the same drivers run over RCCL, one rank per GPU."""
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


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from runmat_amd import HipProvider
    from runmat_amd import sharding as sh

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        group = sh.Group.from_env()
        prov = HipProvider(0)
        seed = 0x9E3779B97F4A7C15
        M, T = 200001, 3  # odd M: the last pair is half used
        p_fused, s_fused = sh.monte_carlo_price_fused(prov, group, M, T, rng_state=seed)
        p_evol, s_evol = sh.monte_carlo_price_evolved(prov, group, M, T, rng_state=seed)
        # row-sharded matmul: this rank's rows of A from the global generator, B replicated
        m, k, n = 384, 96, 160
        rng = np.random.default_rng(11)
        A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
        r0, r1 = sh.row_block(m, group)
        c_rows = prov.download_matrix(sh.matmul_row_sharded(prov, prov.upload(A[r0:r1, :]), prov.upload(B)))
        C = sh.gather_row_blocks(group, c_rows, m)
        # block-column cyclic LU: every rank builds the same A and keeps the column blocks it owns
        nn, nb, nrhs = 1000, 128, 2
        rng2 = np.random.default_rng(12)
        AA = rng2.standard_normal((nn, nn))
        BB = rng2.standard_normal((nn, nrhs))
        blocks = sh.owned_blocks(nn, nb, group)
        cols = [c for p in blocks for c in range(p * nb, min(nn, (p + 1) * nb))]
        x = prov.download_matrix(sh.mldivide_block_cyclic(prov, group, prov.upload(AA[:, cols]), nn, prov.upload(BB), nb=nb))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p_fused=p_fused, s_fused=np.uint64(s_fused), p_evol=p_evol,
                 s_evol=np.uint64(s_evol), C=C, x=x)
        prov.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu(oracle, tmp_path):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
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
    assert np.array_equal(res[0]["x"], res[1]["x"])
