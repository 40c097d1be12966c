"""Randomised fused-elementwise programs: random op DAGs over the IEEE-exact vocabulary, random broadcast patterns.
The device result must be BIT-identical to evaluating the same plan op by op with the CPU oracle (front-end, code
generator, broadcast addressing and CPU-policy helpers all in one property)."""
import numpy as np
import pytest

from planner_requests import FusionGroupPlan

pytestmark = pytest.mark.gpu

PRIM_BIN = {"Add": "add", "Sub": "sub", "ElemMul": "mul", "ElemDiv": "div"}
PRIM_UN = {"Neg": "neg", "UPlus": "uplus"}
BUILTIN_UN = ["abs", "floor", "ceil", "round", "fix", "sign", "heaviside", "sqrt"]
BUILTIN_BIN = ["max", "min", "mod", "rem"]


def _random_plan(rng, n_inputs, n_ops):
    plan = FusionGroupPlan()
    vals = [plan.input() for _ in range(n_inputs)]
    for _ in range(n_ops):
        kind = rng.integers(0, 4)
        if kind == 0:
            name = rng.choice(list(PRIM_BIN))
            a, b = rng.choice(vals, 2)
            vals.append(plan.primitive(name, int(a), int(b)))
        elif kind == 1:
            name = rng.choice(list(PRIM_UN))
            vals.append(plan.primitive(name, int(rng.choice(vals))))
        elif kind == 2:
            vals.append(plan.builtin(str(rng.choice(BUILTIN_UN)), int(rng.choice(vals))))
        else:
            a, b = rng.choice(vals, 2)
            vals.append(plan.builtin(str(rng.choice(BUILTIN_BIN)), int(a), int(b)))
    return plan, vals[-1]


def _eval_oracle(plan, out_id, arrays, oracle):
    env = {vid: arrays[i] for i, vid in enumerate(plan.inputs)}
    for op in plan.operations:
        args = [env[i] for i in op.inputs]
        if op.kind == "primitive" and op.name in PRIM_BIN:
            env[op.output] = oracle.binary(PRIM_BIN[op.name], *args)
        elif op.kind == "primitive":
            env[op.output] = oracle.unary(PRIM_UN[op.name], args[0])
        elif len(args) == 1:
            env[op.output] = oracle.unary(op.name, args[0])
        else:
            env[op.output] = oracle.binary(op.name, *args)
    return env[out_id]


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("seed", range(40))
def test_random_exact_programs_bitwise(prov, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    rank = int(rng.integers(1, 4))
    out_shape = tuple(int(rng.choice([1, 2, 3, 5, 8, 17, 64, 130])) for _ in range(rank))
    if rank == 1:
        out_shape = (out_shape[0], 1)  # tensors are at least 2-D at the boundary
    n_inputs = int(rng.integers(1, 5))
    arrays = []
    for k in range(n_inputs):
        mode = rng.integers(0, 4) if k else 0          # input 0 is always full size, so the output shape is out_shape
        if mode == 0:
            shp = out_shape
        elif mode == 1:
            shp = tuple(1 for _ in out_shape)           # scalar operand (1-element tensor)
        else:
            shp = tuple(d if rng.random() < 0.5 else 1 for d in out_shape)
        vals = rng.choice([-2.5, -1.0, -0.0, 0.0, 0.5, 1.0, 3.0, 7.25, np.inf, -np.inf, np.nan], size=shp,
                          p=[.14, .14, .05, .1, .14, .14, .1, .1, .03, .03, .03])
        arrays.append(np.asarray(vals, dtype=np.float64))
    plan, out_id = _random_plan(rng, n_inputs, int(rng.integers(2, 11)))
    shader = plan.generate_wgsl_for_output(out_id, "f64")
    want = _eval_oracle(plan, out_id, arrays, oracle)
    want = np.broadcast_to(want, out_shape)  # the final value may depend on broadcast operands only
    handles = [prov.upload(a.reshape(-1, order="F"), a.shape) for a in arrays]
    got = prov.download(prov.fused_elementwise(shader, handles, out_shape, int(np.prod(out_shape))))
    assert _bits_equal(got, want.reshape(-1, order="F")), (seed, out_shape, [a.shape for a in arrays], [o.name for o in plan.operations])
