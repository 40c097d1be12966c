"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests) for the few real exchange steps.

The reference has no multi-device code at all; this is new work, designed for how
the path shards:

  * elementwise / independent matrices : contiguous slabs of the column-major index, no exchange;
  * C = A*B                            : row-block -- rank g computes C[rows_g,:] = A[rows_g,:]*B
                                         with B replicated; all-gather of the row blocks only when
                                         a replicated C is asked for;
  * sum / mean / Monte-Carlo           : local partial + a ONE-value exchange, summed in rank order
                                         so results do not depend on collective reduction order;
  * randn                              : every rank skips ahead in the same 64-bit LCG stream
                                         (random.rs:238-256), so the sharded stream IS the
                                         single-device / CPU stream.

Everything here is host logic over a provider object (`HipProvider`, or any object with the same
methods -- the CPU tests use an oracle-backed double), so it is testable with world_size 2 on gloo.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): the only bulk collective (all-gather of C)
moves each row block once to every peer; everything else is a few bytes.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

LCG_MULT = 6364136223846793005
LCG_INC = 1
MASK64 = (1 << 64) - 1


def partition(total: int, world: int, rank: int, granule: int = 1) -> Tuple[int, int]:
    """Balanced contiguous split of `total` items in units of `granule`: [start, stop) of `rank`.
    The first (units % world) ranks get one extra unit; the last rank absorbs the ragged tail."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = (total + granule - 1) // granule
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * granule, total), min(u1 * granule, total)


def lcg_advance(state: int, delta: int) -> int:
    """advance_state of crates/runmat-runtime/src/builtins/common/random.rs:238-256."""
    cur_mult, cur_plus, acc_mult, acc_plus = LCG_MULT, LCG_INC, 1, 0
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & MASK64
            acc_plus = (acc_plus * cur_mult + cur_plus) & MASK64
        cur_plus = (cur_plus * (cur_mult + 1)) & MASK64
        cur_mult = (cur_mult * cur_mult) & MASK64
        delta >>= 1
    return (acc_mult * state + acc_plus) & MASK64


def _flush_c_stdio() -> None:
    """Flush the C library's stdio buffers (librccl writes its banner with printf; on a pipe that text would otherwise
    appear at exit, after whatever Python printed last - e.g. bench.py's one JSON line)."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


@dataclass
class Group:
    """Thin view of the process group: rank, world and an ordered all-gather of small f64 vectors.

    Two data paths: `native` (a provider whose context holds a communicator, `rmhip_comm_init`: every exchange below is
    one C-ABI collective - RCCL over xGMI, or the host shared-memory transport when ranks share a GPU), or
    `torch.distributed` tensors (the CPU tests' oracle-backed doubles on gloo).  `dist` stays the control plane
    (rendezvous, the bench's timing reduction) in both."""
    rank: int = 0
    world: int = 1
    dist: Optional[object] = None  # torch.distributed module when world > 1
    device: str = "cpu"            # "cuda" for nccl, "cpu" for gloo
    native: Optional[object] = None  # HipProvider with a communicator of this rank / world

    def with_native_comm(self, prov, transport: str = "rccl") -> "Group":
        """Create the C-ABI communicator on `prov`: rank 0 makes the id, the control plane distributes its 128 bytes."""
        if self.world > 1:
            # rank 0 may fail to make the id (librccl not loadable, ...): it still broadcasts - an error sentinel - so that every
            # rank leaves this collective and raises the same way (a rank 0 that raised before the broadcast left the others in it)
            payload = [None]
            if self.rank == 0:
                try:
                    payload = [("ok", prov.comm_unique_id(transport))]
                except Exception as e:  # noqa: BLE001
                    payload = [("error", str(e)[:300])]
            self.dist.broadcast_object_list(payload, src=0)
            status, uid = payload[0]
            if status != "ok":
                raise RuntimeError(f"native communicator: rank 0 could not create the id: {uid}")
        else:
            uid = prov.comm_unique_id(transport)
        prov.comm_init(uid, self.rank, self.world)
        _flush_c_stdio()  # RCCL prints a version banner through C stdio on rank 0: out now, not at process exit
        self.native = prov
        return self

    def try_native_comm(self, prov, transport: str = "rccl"):
        """`with_native_comm`, then the ranks AGREE on the outcome through the control plane (a MIN over one flag): either every rank
        holds the native communicator afterwards, or none does (a rank whose init succeeded while a peer's failed - or timed out waiting
        for it, rmhip_comm_init's bounded wait - destroys its half) and the group keeps exchanging through torch.distributed.
        Returns (ok, note): the same on every rank."""
        ok, why = 1, ""
        try:
            self.with_native_comm(prov, transport=transport)
        except Exception as e:  # noqa: BLE001 - any failure means "fall back"
            ok, why = 0, str(e)[:200]
        if self.world > 1:
            import torch

            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
            agreed = int(flag.item()) == 1
        else:
            agreed = ok == 1
        if agreed:
            return True, f"rmhip_comm_* ({'RCCL' if transport == 'rccl' else 'host shared memory'})"
        if self.native is not None:
            try:
                prov.comm_destroy()
            except Exception:  # noqa: BLE001
                pass
            self.native = None
        return False, "torch.distributed (native communicator unavailable" + (f": {why}" if why else " on another rank") + ")"

    @staticmethod
    def from_env() -> "Group":
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            return Group(dist.get_rank(), dist.get_world_size(), dist, dev)
        return Group()

    def all_gather_f64(self, values: Sequence[float]) -> np.ndarray:
        """Returns a [world, len(values)] array, row r = rank r's values (identical on all ranks)."""
        local = np.asarray(values, dtype=np.float64).reshape(1, -1)
        if self.world == 1:
            return local
        if self.native is not None:
            p = self.native
            h = p.upload(local.reshape(-1, 1))
            g = p.comm_allgather_f64(h)
            out = p.download_matrix(g).T.copy()  # [k, world] -> [world, k]
            p.free(h)
            p.free(g)
            return out
        import torch

        t = torch.from_numpy(local.copy()).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.concatenate([o.cpu().numpy() for o in out], axis=0)

    def ordered_sum(self, value: float) -> float:
        """Sum of one value per rank, accumulated in rank order (deterministic, same on every rank)."""
        parts = self.all_gather_f64([value])[:, 0]
        s = 0.0
        for v in parts:
            s += float(v)
        return s

    def barrier(self) -> None:
        if self.world > 1:
            if self.native is not None:
                self.native.comm_barrier()
            else:
                self.dist.barrier()


# ---- matmul ------------------------------------------------------------------------------------
def row_block(rows: int, group: Group, granule: int = 128) -> Tuple[int, int]:
    """Rows of A / C owned by this rank. Multiples of the dgemm tile height keep every rank on the
    unguarded kernel when rows % (128*world) == 0."""
    return partition(rows, group.world, group.rank, granule)


def matmul_row_sharded(prov, a_rows, b):
    """C[rows_g,:] = A[rows_g,:] * B. `a_rows` is this rank's row block (rows_g x k, its own
    column-major buffer), `b` the replicated k x n operand. No collective."""
    return prov.matmul(a_rows, b)


def gather_row_blocks(group: Group, local_block: np.ndarray, rows_total: int) -> np.ndarray:
    """Host-side reassembly of a replicated C from per-rank row blocks (used by tests and by callers
    that need C on the host). `local_block` is rows_g x n."""
    if group.world == 1:
        return local_block
    import torch

    n = local_block.shape[1]
    counts = [partition(rows_total, group.world, r, 128) for r in range(group.world)]
    # column-major rows_g x n == row-major n x rows_g: gather the transposes, concatenate along dim 1
    # all_gather needs equal sizes: ragged blocks are padded to the widest and trimmed afterwards
    width = max(c1 - c0 for c0, c1 in counts)
    mine = torch.zeros((width, n), dtype=torch.float64, device=group.device)
    mine[: local_block.shape[0], :] = torch.from_numpy(np.ascontiguousarray(local_block)).to(group.device)
    outs = [torch.empty_like(mine) for _ in counts]
    group.dist.all_gather(outs, mine)
    return torch.cat([o[: c1 - c0, :] for o, (c0, c1) in zip(outs, counts)], dim=0).cpu().numpy()


def gather_row_blocks_device(group: Group, prov, c_rows, rows_total: int):
    """Device-side all-gather (RCCL over xGMI) of C row blocks into a replicated rows_total x n
    buffer owned by the provider. Zero-copy: torch views the provider's memory through
    `rmhip_device_ptr`, the result is adopted with `rmhip_wrap_external`."""
    import torch

    rows_g, n = c_rows.shape
    if group.world == 1:
        return c_rows, None
    _require_f64(prov, "gather_row_blocks_device")  # the gathered buffer is adopted with rmhip_wrap_external (f64 memory)
    if group.native is not None:  # one C-ABI collective (rmhip_comm_allgather_rows): no torch tensor on the data path
        return group.native.comm_allgather_rows(c_rows, rows_total, 128), None
    counts = [partition(rows_total, group.world, r, 128) for r in range(group.world)]
    width = max(c1 - c0 for c0, c1 in counts)
    view = _torch_view(prov, c_rows, (n, rows_g))  # column-major rows_g x n == row-major n x rows_g
    prov.synchronize()
    if rows_g == width:
        mine = view
    else:  # ragged last block: pad to the widest so all_gather sees equal sizes
        mine = torch.zeros((n, width), dtype=torch.float64, device="cuda")
        mine[:, :rows_g] = view
    outs = [torch.empty((n, width), dtype=torch.float64, device="cuda") for _ in counts]
    group.dist.all_gather(outs, mine)
    full = torch.cat([o[:, : c1 - c0] for o, (c0, c1) in zip(outs, counts)], dim=1).contiguous()
    # (n, rows_total) row-major == column-major rows_total x n
    torch.cuda.synchronize()
    handle = prov.wrap_external(full.data_ptr(), (rows_total, n))
    return handle, full  # keep `full` alive as long as the handle is used


class _CudaArray:
    def __init__(self, ptr: int, shape: Tuple[int, ...], typestr: str = "<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 3,
                                         "strides": None}


def _require_f64(prov, what: str) -> None:
    """The exchange helpers below move f64 device memory; a precision-32 provider's shards are gathered on the host
    (`gather_row_blocks`) or not at all (sharded results stay sharded)."""
    if getattr(prov, "precision", lambda: "F64")() != "F64":
        from .provider import ProviderError
        raise ProviderError(2, f"{what}: needs a precision-64 provider")


def _torch_view(prov, handle, shape):
    """Zero-copy torch view of a library buffer for the collectives; the element type follows the buffer's storage
    (f32 on a precision-32 provider, `rmhip_buffer_bits`)."""
    import torch

    typestr = "<f4" if hasattr(prov, "buffer_bits") and prov.buffer_bits(handle) == 32 else "<f8"
    return torch.as_tensor(_CudaArray(prov.device_ptr(handle), shape, typestr), device="cuda")


# ---- reductions --------------------------------------------------------------------------------
def sum_all_sharded(prov, group: Group, local) -> float:
    """sum(x,'all') of a tensor sharded in contiguous slabs: local device sum + ordered exchange."""
    part = float(prov.download(prov.reduce_sum(local))[0])
    return group.ordered_sum(part)


# ---- Monte-Carlo (benchmarks/monte-carlo-analysis/runmat_rng.m, f64) ----------------------------
def monte_carlo_price_sharded(prov, group: Group, M: int, T: int, S0=100.0, mu=0.05, sigma=0.2, dt=1.0 / 252.0,
                              K=100.0, rng_state: Optional[int] = None) -> Tuple[float, int]:
    """price = mean(max(S_T - K, 0)) * exp(-mu*T*dt) with S evolved by T randn steps.

    Rank g owns paths [start, stop), start even (pair boundary of Box-Muller). At step t the global
    generator would be at state_t = advance(state_0, t * 2*ceil(M/2)); rank g starts its draw at
    advance(state_t, start). Returns (price, final global rng state) -- identical on every rank."""
    if rng_state is None:
        rng_state = prov.get_rng_state()
    start, stop = partition(M, group.world, group.rank, granule=2)
    count = stop - start
    per_step = 2 * ((M + 1) // 2)
    drift = (mu - 0.5 * sigma * sigma) * dt
    scale = sigma * math.sqrt(dt)
    partial = 0.0
    if count > 0:
        S = prov.fill((count, 1), S0)
        for t in range(T):
            prov.set_rng_state(lcg_advance(rng_state, t * per_step + start))
            Z = prov.random_normal((count, 1))
            zs = prov.scalar_mul(Z, scale)
            zd = prov.scalar_add(zs, drift)
            e = prov.unary_exp(zd)
            S_next = prov.elem_mul(S, e)
            for h in (Z, zs, zd, e, S):
                prov.free(h)
            S = S_next
        diff = prov.scalar_sub(S, K)
        payoff = prov.scalar_max(diff, 0.0)
        psum = prov.reduce_sum(payoff)
        partial = float(prov.download(psum)[0])
        for h in (S, diff, payoff, psum):
            prov.free(h)
    total = group.ordered_sum(partial)
    final_state = lcg_advance(rng_state, T * per_step)
    prov.set_rng_state(final_state)
    return (total / float(M)) * math.exp(-mu * T * dt), final_state


def monte_carlo_price_fused(prov, group: Group, M: int, T: int, shaders: Tuple[str, str], S0=100.0, mu=0.05, sigma=0.2, dt=1.0 / 252.0,
                            rng_state: Optional[int] = None) -> Tuple[float, int]:
    """Same workload as `monte_carlo_price_sharded`, issued the way RunMat's planner would: one
    fused elementwise kernel per time step (`S = S .* exp(drift + scale .* Z)`, constants as
    1-element inputs) and one fused reduction for `sum(max(S - K, 0))`.  Materialised traffic per
    path and step: randn write 8 B + fused update 24 B, plus 8 B for the final reduction
    ((32*T + 8) * M bytes).

    `shaders` = (step, payoff): the WGSL text of `S .* exp(drift + scale .* Z)` (inputs S, Z, scale, drift) and of the reduction
    `sum(max(S - K, 0))`, as RunMat's planner emits them once when it compiles the script's fusion groups (fusion.rs:679-682).
    This module does not generate requests - the caller brings them (tests/workloads.py `monte_carlo_shaders(K)` for tests and bench)."""
    from .provider import ReductionFlavor

    if rng_state is None:
        rng_state = prov.get_rng_state()
    start, stop = partition(M, group.world, group.rank, granule=2)
    count = stop - start
    per_step = 2 * ((M + 1) // 2)
    drift = (mu - 0.5 * sigma * sigma) * dt
    scale = sigma * math.sqrt(dt)
    partial = 0.0
    if count > 0:
        step_shader, red_shader = shaders
        # constants are 1-element tensors created on the device (no host copy, no synchronisation); the initial price
        # S0 is one too and broadcasts into the first update (`S = S0 .* exp(...)`: the planner hands scalars to the
        # kernel as [1,1] inputs, fusion_exec.rs:279,305-326), so no M-element fill precedes the time loop
        h_scale = h_drift = S = None
        for t in range(T):
            prov.set_rng_state(lcg_advance(rng_state, t * per_step + start))
            Z = prov.random_normal((count, 1))
            if S is None:  # the three scalar fills are enqueued BEHIND the first randn: the host's work for them hides under that kernel
                h_scale = prov.fill((1, 1), scale)
                h_drift = prov.fill((1, 1), drift)
                S = prov.fill((1, 1), S0)
            S_next = prov.fused_elementwise(step_shader, [S, Z, h_scale, h_drift], (count, 1), count)
            prov.free(Z)
            prov.free(S)
            S = S_next
        if S is None:  # T == 0
            h_scale, h_drift, S = prov.fill((1, 1), scale), prov.fill((1, 1), drift), prov.fill((count, 1), S0)
        psum = prov.fused_reduction(red_shader, [S], (1,), count, 1, 256, ReductionFlavor.Sum())
        partial = float(prov.download(psum)[0])
        for h in (S, psum, h_scale, h_drift):
            prov.free(h)
    total = group.ordered_sum(partial)
    final_state = lcg_advance(rng_state, T * per_step)
    prov.set_rng_state(final_state)
    return (total / float(M)) * math.exp(-mu * T * dt), final_state


def monte_carlo_price_evolved(prov, group: Group, M: int, T: int, S0=100.0, mu=0.05, sigma=0.2, dt=1.0 / 252.0, K=100.0,
                              rng_state: Optional[int] = None, payoff_shader: Optional[str] = None) -> Tuple[float, int]:
    """Same workload and the same random stream again, with the whole time loop as ONE provider call
    (`stochastic_evolution`, the idiom RunMat's VM recognises: crates/runmat-vm/src/accel/idioms/
    stochastic_evolution.rs) followed by one fused reduction.  HBM traffic per path: fill 8 B + evolve 16 B +
    payoff sum 8 B = 32 B for any T (the materialised plan moves (32*T + 8) B).  `payoff_shader`: the planner's fused reduction
    `sum(max(S - K, 0))` (see `monte_carlo_price_fused`); without it the payoff is the three per-op calls."""
    from .provider import ReductionFlavor

    if rng_state is None:
        rng_state = prov.get_rng_state()
    start, stop = partition(M, group.world, group.rank, granule=2)
    count = stop - start
    per_step = 2 * ((M + 1) // 2)
    drift = (mu - 0.5 * sigma * sigma) * dt
    scale = sigma * math.sqrt(dt)
    partial = 0.0
    if count > 0:
        S = prov.fill((count, 1), S0)
        prov.set_rng_state(lcg_advance(rng_state, start))
        S_end = prov.stochastic_evolution(S, drift, scale, T, draws_per_step=per_step)
        temps = [S, S_end]
        if payoff_shader is not None:
            psum = prov.fused_reduction(payoff_shader, [S_end], (1,), count, 1, 256, ReductionFlavor.Sum())
        else:  # per-op form (max(S - K, 0) then sum), as monte_carlo_price_sharded
            d = prov.scalar_sub(S_end, K)
            pay = prov.scalar_max(d, 0.0)
            psum = prov.reduce_sum(pay)
            temps += [d, pay]
        partial = float(prov.download(psum)[0])
        for h in temps + [psum]:
            prov.free(h)
    total = group.ordered_sum(partial)
    final_state = lcg_advance(rng_state, T * per_step)
    prov.set_rng_state(final_state)
    return (total / float(M)) * math.exp(-mu * T * dt), final_state


# ---- x = A\\b across GPUs: 1-D block-column cyclic LU with one panel broadcast per block -----------
def owned_blocks(n: int, nb: int, group: Group) -> List[int]:
    """Global column-block ids owned by this rank (block p -> rank p % world)."""
    nblocks = (n + nb - 1) // nb
    return [p for p in range(nblocks) if p % group.world == group.rank]


def local_col_offset(p: int, nb: int, group: Group) -> int:
    """First local column of global block p on its owner (blocks are stored in ownership order)."""
    return (p // group.world) * nb


def _bcast(group: Group, prov, handle, shape, src: int) -> None:
    """In-place broadcast of a provider buffer. nccl: zero-copy torch view of the device memory;
    gloo (CPU tests): the test double exposes `.arr`."""
    if group.world == 1:
        return
    if group.native is not None:
        group.native.comm_bcast(handle, src)
        return
    import torch

    if group.device == "cuda":
        prov.synchronize()
        t = _torch_view(prov, handle, tuple(reversed(shape)))  # column-major (r, c) == row-major (c, r)
        group.dist.broadcast(t, src)
        torch.cuda.synchronize()
    elif hasattr(handle, "arr"):
        t = torch.from_numpy(handle.arr)
        group.dist.broadcast(t, src)
    else:
        # a device buffer with a CPU backend (several ranks sharing one GPU in a control-flow test): stage
        # through the host -- download, broadcast, write back in place on the receivers
        host = np.ascontiguousarray(prov.download(handle), dtype=np.float64)
        t = torch.from_numpy(host)
        group.dist.broadcast(t, src)
        if group.rank != src:
            rows = int(shape[0]) if len(shape) else 1
            cols = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            tmp = prov.upload(host.reshape((rows, cols), order="F"))
            prov.blk_assign((handle, 0, 0, rows, cols), tmp)
            prov.free(tmp)


def mldivide_block_cyclic(prov, group: Group, a_local, n: int, b, nb: int = 512):
    """Solve A x = b with A distributed by column blocks (block p of width nb on rank p % world,
    stored contiguously in ownership order in `a_local`, an n x ncols_local buffer that is
    OVERWRITTEN with its part of the LU factors) and b (n x nrhs) replicated.  Returns the replicated
    solution handle.

    Needs a precision-64 provider (the block views update f64 storage in place).

    Per block p: the owner factors its panel (rows j.., the host_lu.rs pivot rule), broadcasts the
    factored panel plus the interchanges (one RCCL broadcast of (n-j) x nb doubles: 64 MiB for the
    first panel at n = 16384, nb = 512, shrinking linearly), then every rank applies interchanges,
    triangular solve and the MFMA dgemm update to the trailing blocks IT owns and, redundantly, to its
    copy of b.  The back substitution walks the blocks in reverse: the owner solves with its U_pp,
    updates y[0:j] and broadcasts the finished prefix."""
    _require_f64(prov, "mldivide_block_cyclic")
    from .provider import ProviderError

    nrhs = b.shape[1] if len(b.shape) > 1 else 1
    nblocks = (n + nb - 1) // nb
    mine = owned_blocks(n, nb, group)
    ncols_loc = sum(min(nb, n - p * nb) for p in mine)
    y = prov.blk_copy((b, 0, 0, n, nrhs))
    overlap = group.native is not None and group.world > 1  # asynchronous broadcasts on the communication stream

    def post_panel(p):
        """Owner: factor block p (its columns are up to date) and send it; everyone else: post the receive.  With a
        native communicator both are asynchronous: the transfer runs under whatever is enqueued next."""
        j, w, owner = p * nb, min(nb, n - p * nb), p % group.world
        if group.rank == owner:
            lq = local_col_offset(p, nb, group)
            ipiv, info = prov.blk_lu((a_local, j, lq, n - j, w))
            panel = prov.blk_copy((a_local, j, lq, n - j, w))
            aux = prov.upload(np.concatenate([prov.download(ipiv), [float(info)]]).reshape(-1, 1))
            prov.free(ipiv)
        else:
            panel = prov.zeros((n - j, w))
            aux = prov.zeros((w + 1, 1))
        if overlap:
            group.native.comm_bcast(panel, owner, async_=True)
            group.native.comm_bcast(aux, owner, async_=True)
        else:
            _bcast(group, prov, panel, (n - j, w), owner)
            _bcast(group, prov, aux, (w + 1, 1), owner)
        return panel, aux

    def update(panel, ipiv, j, w, lc0, ncl):
        """interchanges, U block row and Schur update of local columns [lc0, lc0 + ncl) by panel (j, w)"""
        if ncl <= 0:
            return
        prov.blk_swap_rows((a_local, j, lc0, n - j, ncl), ipiv)
        prov.blk_trsm(False, (panel, 0, 0, w, w), (a_local, j, lc0, w, ncl))
        if n - j - w > 0:
            prov.blk_gemm(-1.0, (panel, w, 0, n - j - w, w), (a_local, j, lc0, w, ncl), 1.0, (a_local, j + w, lc0, n - j - w, ncl))

    cur = post_panel(0)
    for p in range(nblocks):
        j = p * nb
        w = min(nb, n - j)
        panel, aux = cur
        if overlap:
            group.native.comm_wait()
        auxh = prov.download(aux)
        if float(auxh[w]) > 0:
            for h in (panel, aux, y):
                prov.free(h)
            raise ProviderError(7, "mldivide: pivot <= 1e-12; matrix is numerically singular, use the CPU SVD path")
        ipiv = prov.upload(auxh[:w].reshape(-1, 1))
        later = [q for q in mine if q > p]
        nxt = None
        if p + 1 < nblocks:
            # depth-1 look-ahead: block p+1 is brought up to date first and its factorisation / broadcast posted, so the
            # transfer of panel p+1 overlaps the bulk of update p on every rank
            if later and later[0] == p + 1:
                lc = local_col_offset(p + 1, nb, group)
                update(panel, ipiv, j, w, lc, min(nb, n - (p + 1) * nb))
                later = later[1:]
            nxt = post_panel(p + 1)
        if later:
            lc0 = local_col_offset(later[0], nb, group)
            update(panel, ipiv, j, w, lc0, ncols_loc - lc0)
        prov.blk_swap_rows((y, j, 0, n - j, nrhs), ipiv)
        prov.blk_trsm(False, (panel, 0, 0, w, w), (y, j, 0, w, nrhs))
        if n - j - w > 0:
            prov.blk_gemm(-1.0, (panel, w, 0, n - j - w, w), (y, j, 0, w, nrhs), 1.0, (y, j + w, 0, n - j - w, nrhs))
        for h in (panel, aux, ipiv):
            prov.free(h)
        cur = nxt
    for p in reversed(range(nblocks)):
        j = p * nb
        w = min(nb, n - j)
        owner = p % group.world
        if group.rank == owner:
            lq = local_col_offset(p, nb, group)
            prov.blk_trsm(True, (a_local, j, lq, w, w), (y, j, 0, w, nrhs))
            if j > 0:
                prov.blk_gemm(-1.0, (a_local, 0, lq, j, w), (y, j, 0, w, nrhs), 1.0, (y, 0, 0, j, nrhs))
        if group.world > 1:
            pre = prov.blk_copy((y, 0, 0, j + w, nrhs))
            _bcast(group, prov, pre, (j + w, nrhs), owner)
            prov.blk_assign((y, 0, 0, j + w, nrhs), pre)
            prov.free(pre)
    return y


# ---- row-partitioned A \\ b (BASELINE.json configs[4]: "row-partitioned ... RCCL all-gather") ------------------------------------
class PivotGrowth(RuntimeError):
    """A multiplier outside the diagonal domain exceeded the bound: solve with `mldivide_block_cyclic` (grid-wide pivot rule)."""


def owned_row_blocks(n: int, rb: int, group: Group) -> List[int]:
    """Row blocks of height `rb` owned by this rank (block q on rank q % world), ascending = local storage order."""
    return [q for q in range((n + rb - 1) // rb) if q % group.world == group.rank]


def local_row_offset(q: int, rb: int, group: Group) -> int:
    """First local row of row block q on its owner (blocks are stored in ownership order, all but the last full)."""
    return (q // group.world) * rb


def mldivide_row_partitioned(prov, group: Group, ab_local, n: int, nrhs: int, rb: int = 512, tau: float = 8.0):
    """Solve A x = b with [A | b] distributed BY ROWS: row block q (height rb) of the n x (n + nrhs) augmented matrix lives on rank
    q % world, stored in ownership order in `ab_local` (OVERWRITTEN with this rank's rows of the factors).  Returns the replicated
    solution handle (n x nrhs), identical on every rank.

    Pivots never leave a solve, so - as on one GPU (lu.hip, solve path) - pivoting is restricted to a DIAGONAL DOMAIN and verified:
    panel p (columns of row block p) is factored by its owner with partial pivoting among the owner's OWN rows from the diagonal tile
    down ((n - j) / world of them): every interchange is local to one rank, no row ever crosses the fabric.  What does:
      * ONE broadcast per panel of the owner's tile row [L11\\U11 | U12 | y-part] - rb x (n + nrhs - j) doubles, 64 MiB for the first
        panel at n = 16384, rb = 512, shrinking linearly: the same volume the block-column form moves, but every rank then
        computes ITS rows' multipliers (L21 = A21 U11^-1) and trailing update (A22 -= L21 U12) with no further exchange;
      * the last `world` row blocks, whose owners have no rows left below the diagonal tile to pivot among, are ALL-GATHERED
        (<= world * rb rows) and finished by every rank redundantly with the single-GPU solve;
      * the back substitution broadcasts one rb x nrhs solution block per panel.
    The right-hand sides ride along as extra columns, so the forward substitution is part of the trailing update.
    Guard: the largest multiplier a rank computes for rows outside the owner's domain; beyond `tau` (one exchange at the end) the
    factorisation is not trusted and `PivotGrowth` is raised - callers then use `mldivide_block_cyclic` (the grid-wide rule).
    Needs a precision-64 provider."""
    _require_f64(prov, "mldivide_row_partitioned")
    from .provider import ProviderError

    world, rank = group.world, group.rank
    ncols = n + nrhs
    nblocks = (n + rb - 1) // rb
    mine = owned_row_blocks(n, rb, group)
    nloc = sum(min(rb, n - q * rb) for q in mine)
    # phase 1 covers the panels whose owner still has a block below the tile; the rest is gathered
    n_direct = max(0, nblocks - world)
    growth = 0.0

    def bcast(handle, shape, src):
        if group.native is not None and world > 1:
            group.native.comm_bcast(handle, src)
        elif world > 1:
            _bcast(group, prov, handle, shape, src)

    def first_local_row_at_or_after(q):
        """local row offset of this rank's first block with index >= q (nloc when there is none)"""
        for b in mine:
            if b >= q:
                return local_row_offset(b, rb, group)
        return nloc

    tiles = []  # (j, w, tile-row handle [w x (ncols - j)]) of every direct panel, kept for the back substitution
    # A failure on ONE rank (a pivot at the singular cut-off inside its domain, a ProviderError from a block kernel) must not leave the
    # others blocked in the panel broadcast: the failing rank sends a NaN-poisoned tile, keeps taking part in every collective and
    # reports NaN to the guard, so every rank raises PivotGrowth together after the one exchange at the end (the same protocol as
    # rmhip_mldivide_row_partitioned, csrc/sharded.cpp).
    failed = None

    def note(v):
        nonlocal growth
        if v != v or v > growth:  # NaN sticks: max(0.0, nan) is 0.0 in Python
            growth = v

    for p in range(n_direct):
        j, w, owner = p * rb, rb, p % world
        width = ncols - j
        tile = None
        if rank == owner and failed is None:
            lr = local_row_offset(p, rb, group)
            try:
                ipiv, info = prov.blk_lu((ab_local, lr, j, nloc - lr, w))  # partial pivoting among this rank's rows from the tile down
                if info > 0:
                    prov.free(ipiv)
                    raise PivotGrowth(f"panel {p}: {info} pivot(s) at the singular cut-off inside the diagonal domain")
                if j > 0:
                    prov.blk_swap_rows((ab_local, lr, 0, nloc - lr, j), ipiv)               # the interchanges on the L part ...
                prov.blk_swap_rows((ab_local, lr, j + w, nloc - lr, width - w), ipiv)       # ... and on everything to the right (b included)
                prov.free(ipiv)
                prov.blk_trsm(False, (ab_local, lr, j, w, w), (ab_local, lr, j + w, w, width - w))  # U12 and the y part: L11^-1 [A12 | b]
                tile = prov.blk_copy((ab_local, lr, j, w, width))
            except (PivotGrowth, ProviderError) as e:
                failed = str(e)
            below = lr + w
        else:
            below = first_local_row_at_or_after(p + 1)
        if tile is None:
            tile = prov.fill((w, width), float("nan")) if rank == owner else prov.zeros((w, width))
        bcast(tile, (w, width), owner)
        mb = nloc - below
        if mb > 0 and failed is None:
            try:
                if rank != owner:  # the owner's rows below the tile were factored with it
                    prov.blk_trsm(2, (tile, 0, 0, w, w), (ab_local, below, j, mb, w))       # L21 = A21 U11^-1
                    note(prov.blk_absmax((ab_local, below, j, mb, w)))
                prov.blk_gemm(-1.0, (ab_local, below, j, mb, w), (tile, 0, w, w, width - w), 1.0, (ab_local, below, j + w, mb, width - w))
            except ProviderError as e:
                failed = str(e)
        tiles.append((j, w, tile))
    # ---- the guard: one exchange, every rank decides the same way (a failed rank reports NaN)
    gathered = group.all_gather_f64([float("nan") if failed is not None else growth])
    worst = 0.0
    for v in np.asarray(gathered, dtype=np.float64).reshape(-1):
        if v != v or v > worst:
            worst = float(v)
    if failed is not None or not worst <= tau:
        for _, _, t in tiles:
            prov.free(t)
        raise PivotGrowth(failed if failed is not None else f"largest multiplier outside the diagonal domains {worst:.3g} > {tau:g} (or another rank failed)")
    # ---- the remaining rows: gathered, then the single-GPU solve on every rank
    j0 = n_direct * rb
    m_rem = n - j0
    x = prov.zeros((n, nrhs))
    if m_rem > 0:
        trailing = prov.zeros((m_rem, m_rem + nrhs))
        for q in range(n_direct, nblocks):
            h = min(rb, n - q * rb)
            owner = q % world
            if rank == owner:
                blk = prov.blk_copy((ab_local, local_row_offset(q, rb, group), j0, h, m_rem + nrhs))
            else:
                blk = prov.zeros((h, m_rem + nrhs))
            bcast(blk, (h, m_rem + nrhs), owner)
            prov.blk_assign((trailing, q * rb - j0, 0, h, m_rem + nrhs), blk)
            prov.free(blk)
        a_rem = prov.blk_copy((trailing, 0, 0, m_rem, m_rem))
        b_rem = prov.blk_copy((trailing, 0, m_rem, m_rem, nrhs))
        try:
            x_rem = prov.mldivide(a_rem, b_rem)
        except ProviderError:
            for h in (trailing, a_rem, b_rem, x):
                prov.free(h)
            for _, _, t in tiles:
                prov.free(t)
            raise
        prov.blk_assign((x, j0, 0, m_rem, nrhs), x_rem)
        for h in (trailing, a_rem, b_rem, x_rem):
            prov.free(h)
    # ---- back substitution over the direct panels: every rank holds every tile row, so it is redundant and needs no exchange at all
    for j, w, tile in reversed(tiles):
        width = ncols - j
        rhs = prov.blk_copy((tile, 0, width - nrhs, w, nrhs))                # y_p
        later = n - j - w
        if later > 0:
            prov.blk_gemm(-1.0, (tile, 0, w, w, later), (x, j + w, 0, later, nrhs), 1.0, (rhs, 0, 0, w, nrhs))  # y_p - U12 x_later
        prov.blk_trsm(True, (tile, 0, 0, w, w), (rhs, 0, 0, w, nrhs))        # U11^-1
        prov.blk_assign((x, j, 0, w, nrhs), rhs)
        prov.free(rhs)
        prov.free(tile)
    return x

