"""Lazy `random_normal` handles (rmhip.h: rmhip_set_lazy_random; round 6): a handle without storage whose normals the consuming
streaming fused elementwise kernel generates in registers - bit for bit what an eager `random_normal` writes - and which every other
consumer sees materialised under the same id.  The CPU stream being followed: crates/runmat-runtime/src/builtins/common/random.rs
:271-288 (LCG step, uniform), :530-543 (Box-Muller pairs), :238-256 (skip-ahead)."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from planner_requests import FusionGroupPlan, monte_carlo_shaders

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"
S0, MU, SIGMA, DT = 100.0, 0.05, 0.2, 1.0 / 252.0
DRIFT, SCALE = (MU - 0.5 * SIGMA * SIGMA) * DT, SIGMA * math.sqrt(DT)


@pytest.fixture()
def lazy(prov):
    """Lazy handles from 2 elements on for the test, the library's default restored afterwards."""
    prov.set_lazy_random(True, 2)
    yield prov
    prov.set_lazy_random(True, 1024)


def _step(prov, n, state, lazy_on, shader):
    prov.set_lazy_random(lazy_on, 2)
    prov.set_rng_state(state)
    before = prov.lazy_random_stats()
    z = prov.random_normal((n, 1))
    hs, hsc, hdr = prov.fill((1, 1), S0), prov.fill((1, 1), SCALE), prov.fill((1, 1), DRIFT)
    out = prov.fused_elementwise(shader, [hs, z, hsc, hdr], (n, 1), n)
    s = prov.download(out).ravel()
    after = prov.lazy_random_stats()
    end_state = prov.get_rng_state()
    for h in (z, hs, hsc, hdr, out):
        prov.free(h)
    return s, end_state, {k: after[k] - before[k] for k in after}


@pytest.mark.parametrize("n", [2, 3, 1024, 1025, 4097, 100001, (1 << 20) + 3, 3_000_000])
def test_fused_update_on_a_lazy_handle_is_bit_identical_to_the_eager_one(lazy, oracle, n):
    prov = lazy
    shader = monte_carlo_shaders(100.0)[0]
    state = 0x9E3779B97F4A7C15 ^ (n * 0x1234567)
    eager, st_e, d_e = _step(prov, n, state, False, shader)
    lz, st_l, d_l = _step(prov, n, state, True, shader)
    assert d_e == {"created": 0, "fused": 0, "materialised": 0}
    assert d_l == {"created": 1, "fused": 1, "materialised": 0}  # consumed in registers, never written
    assert st_e == st_l == oracle.rng_advance(state, 2 * ((n + 1) // 2))
    assert np.array_equal(eager.view(np.uint64), lz.view(np.uint64))
    # and both follow the CPU stream (libm tolerance of the normals, amplified by S0 * scale)
    z, _ = oracle.rng_normal(state, n)
    assert np.max(np.abs(lz - S0 * np.exp(DRIFT + SCALE * z))) <= 1e-11


def test_any_other_consumer_sees_the_tensor_materialised_under_the_same_id(lazy, oracle):
    prov = lazy
    n, state = 50001, 777
    prov.set_lazy_random(False, 2)
    prov.set_rng_state(state)
    he = prov.random_normal((n, 1))
    want = prov.download(he).ravel()
    prov.set_lazy_random(True, 2)
    for touch in ("download", "per_op", "reduction", "reshape"):
        prov.set_rng_state(state)
        b = prov.lazy_random_stats()
        h = prov.random_normal((n, 1))
        assert prov.lazy_random_stats()["created"] == b["created"] + 1
        if touch == "download":
            got = prov.download(h).ravel()
        elif touch == "per_op":
            t = prov.scalar_mul(h, 1.0)
            got = prov.download(t).ravel()
            prov.free(t)
        elif touch == "reduction":
            t = prov.reduce_sum(h)
            assert abs(prov.download(t).ravel()[0] - math.fsum(want)) <= 1e-9
            prov.free(t)
            got = prov.download(h).ravel()
        else:
            h = prov.reshape(h, (1, n))
            got = prov.download(h).ravel()
        assert prov.lazy_random_stats()["materialised"] == b["materialised"] + 1, touch
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), touch
        # once materialised it is an ordinary tensor: a fused kernel now reads it from memory
        sq = FusionGroupPlan()
        v = sq.input()
        o = sq.primitive("ElemMul", v, v)
        shape = (1, n) if touch == "reshape" else (n, 1)
        r = prov.fused_elementwise(sq.generate_wgsl_for_output(o, "f64"), [h], shape, n)
        assert np.array_equal(prov.download(r).ravel(), want * want)
        assert prov.lazy_random_stats()["fused"] == b["fused"]
        prov.free(r)
        prov.free(h)
    prov.free(he)


def test_a_lazy_handle_read_twice_and_two_lazy_handles_in_one_kernel(lazy):
    prov = lazy
    n, state = 70001, 4242
    prov.set_lazy_random(False, 2)
    prov.set_rng_state(state)
    z1e, z2e = prov.random_normal((n, 1)), prov.random_normal((n, 1))
    a, b = prov.download(z1e).ravel(), prov.download(z2e).ravel()
    prov.set_lazy_random(True, 2)
    prov.set_rng_state(state)
    z1, z2 = prov.random_normal((n, 1)), prov.random_normal((n, 1))
    p = FusionGroupPlan()
    v1, v2 = p.input(), p.input()
    o = p.primitive("Add", p.primitive("ElemMul", v1, v2), v1)
    shader = p.generate_wgsl_for_output(o, "f64")
    before = prov.lazy_random_stats()
    for _ in range(2):  # the same handles again: generated again, same values
        r = prov.fused_elementwise(shader, [z1, z2], (n, 1), n)
        assert np.array_equal(prov.download(r).ravel(), a * b + a)
        prov.free(r)
    r = prov.fused_elementwise(shader, [z1, z1e], (n, 1), n)  # one lazy, one resident
    assert np.array_equal(prov.download(r).ravel(), a * a + a)
    prov.free(r)
    r = prov.fused_elementwise(shader, [z1, z1], (n, 1), n)  # the SAME lazy handle bound to two inputs of one kernel
    assert np.array_equal(prov.download(r).ravel(), a * a + a)
    prov.free(r)
    after = prov.lazy_random_stats()
    assert after["fused"] - before["fused"] == 7 and after["materialised"] == before["materialised"]
    for h in (z1, z2, z1e, z2e):
        prov.free(h)


def test_requests_the_streaming_kernel_does_not_serve_materialise_first(lazy):
    prov = lazy
    n, state = 4096, 99
    prov.set_lazy_random(False, 2)
    prov.set_rng_state(state)
    ze = prov.random_normal((n, 1))
    z = prov.download(ze).ravel()
    prov.set_lazy_random(True, 2)
    row = prov.upload(np.arange(1.0, 4.0), (1, 3))
    p = FusionGroupPlan()
    v1, v2 = p.input(), p.input()
    shader = p.generate_wgsl_for_output(p.primitive("ElemMul", v1, v2), "f64")
    prov.set_rng_state(state)
    h = prov.random_normal((n, 1))
    b = prov.lazy_random_stats()
    r = prov.fused_elementwise(shader, [h, row], (n, 3), 3 * n)  # implicit expansion: the broadcast kernel
    got = prov.download_matrix(r)
    assert np.array_equal(got, z.reshape(-1, 1) * np.arange(1.0, 4.0).reshape(1, 3))
    a = prov.lazy_random_stats()
    assert a["materialised"] == b["materialised"] + 1 and a["fused"] == b["fused"]
    for x in (ze, row, h, r):
        prov.free(x)
    # a precision-32 provider never creates lazy handles
    from runmat_amd import HipProvider

    p32 = HipProvider(0, precision="F32")
    try:
        p32.set_lazy_random(True, 2)
        hz = p32.random_normal((n, 1))
        assert p32.lazy_random_stats()["created"] == 0
        p32.free(hz)
    finally:
        p32.close()


def test_monte_carlo_price_lazy_equals_materialised_bit_for_bit(prov):
    """The Monte-Carlo step of BASELINE configs[3] with the update reading a lazy Z (16 B per sample moved) and a materialised Z
    (32 B): same price bits, same final stream state; against tests/golden/monte_carlo_rng_oracle.json (the oracle's own output,
    tests/golden/make_oracle_numbers.py) within SURVEY 8(d)'s rel 1e-10."""
    from runmat_amd import sharding as sh

    case = [c for c in json.loads((GOLDEN / "monte_carlo_rng_oracle.json").read_text())["cases"] if c["M"] == 10000001][0]
    g = sh.Group()
    shaders = monte_carlo_shaders(100.0)
    out = {}
    try:
        for on in (False, True):
            prov.set_lazy_random(on, 1024)
            b = prov.lazy_random_stats()
            out[on] = sh.monte_carlo_price_fused(prov, g, case["M"], case["T"], shaders, rng_state=case["seed_state"])
            a = prov.lazy_random_stats()
            assert a["created"] - b["created"] == (case["T"] if on else 0)
            assert a["fused"] - b["fused"] == (case["T"] if on else 0) and a["materialised"] == b["materialised"]
    finally:
        prov.set_lazy_random(True, 1024)
    assert out[True][1] == out[False][1] == case["final_state"]
    assert np.float64(out[True][0]).view(np.uint64) == np.float64(out[False][0]).view(np.uint64)
    assert abs(out[True][0] - case["price"]) <= 1e-10 * case["price"]
