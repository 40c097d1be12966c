"""TEST / BENCH INFRASTRUCTURE (not product code; the product is librmhip.so behind include/rmhip.h).

Host-side executors: mirror of `execute_elementwise` / `execute_reduction`
(crates/runmat-accelerate/src/fusion_exec.rs:196-628), i.e. what sits between the VM and the
provider call: resolve the output shape (plan shape or runtime broadcast, trailing-aligned,
:216-277), upload host operands and scalars (scalars become 1-element tensors shaped [1,1,...],
:279-353), generate the request text, call the provider, free the temporaries it uploaded itself
(:415-419).  Values may be `GpuTensorHandle`s (resident operands), numpy arrays (host tensors) or
Python floats/ints (Value::Num / Value::Int).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from planner_requests import FusionGroupPlan
from runmat_amd.provider import GpuTensorHandle, ProviderError, ReductionFlavor

ERR_UNSUPPORTED = 2


def runtime_broadcast_shape(values: Sequence) -> Optional[Tuple[int, ...]]:
    """fusion_exec.rs:216-245: scalars contribute an empty shape; shapes align on trailing dims."""
    shapes: List[Tuple[int, ...]] = []
    for v in values:
        if isinstance(v, GpuTensorHandle):
            shapes.append(tuple(v.shape))
        elif isinstance(v, np.ndarray):
            shapes.append(tuple(v.shape))
        elif isinstance(v, (int, float)):
            shapes.append(())
        else:
            return None
    rank = max((len(s) for s in shapes), default=0)
    out = [1] * rank
    for shape in shapes:
        offset = rank - len(shape)
        for i, dim in enumerate(shape):
            j = offset + i
            a, b = out[j], dim
            if a == 1:
                out[j] = max(b, 1)
            elif b == 1 or a == b:
                pass
            else:
                return None
    return tuple(out)


def normalize_scalar_shape(shape: Sequence[int]) -> Tuple[int, ...]:
    """Scalars are at least 2-D ([1,1]) like MATLAB values."""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 0:
        return (1, 1)
    if len(shape) == 1:
        return (shape[0], 1) if shape[0] != 1 else (1, 1)
    return shape


def _scalar_ty(prov) -> str:
    """`scalar_ty` follows the provider's precision (fusion_exec.rs:262-266)."""
    return "f32" if prov.precision() == "F32" else "f64"


def _prepare(prov, values, scalar_shape):
    prepared, owned = [], []
    for v in values:
        if isinstance(v, GpuTensorHandle):
            prepared.append(v)
        elif isinstance(v, np.ndarray):
            h = prov.upload(np.asarray(v, dtype=np.float64).reshape(-1, order="F"), v.shape if v.ndim else (1, 1))
            prepared.append(h)
            owned.append(h)
        elif isinstance(v, (int, float)):
            h = prov.upload(np.array([float(v)]), scalar_shape)
            prepared.append(h)
            owned.append(h)
        else:
            raise ProviderError(ERR_UNSUPPORTED, "fusion: unsupported value type")
    return prepared, owned


def execute_elementwise(prov, plan: FusionGroupPlan, output_ids: Sequence[int], values: Sequence,
                        plan_shape: Optional[Sequence[Optional[int]]] = None) -> List[GpuTensorHandle]:
    """Returns one resident handle per requested output id (first = the plan's final output)."""
    if len(values) != len(plan.inputs):
        raise ProviderError(1, f"fusion input mismatch: expected {len(plan.inputs)}, got {len(values)}")
    rt = runtime_broadcast_shape(values)
    if plan_shape and all(d is not None for d in plan_shape):
        out_shape = tuple(int(d) for d in plan_shape)
    elif plan_shape and rt is not None and len(rt) == len(plan_shape):
        out_shape = tuple(int(p) if p is not None else r for p, r in zip(plan_shape, rt))
    else:
        if rt is None:
            raise ProviderError(ERR_UNSUPPORTED, "fusion: unknown output shape")
        out_shape = rt
    length = int(np.prod(out_shape, dtype=np.int64)) if len(out_shape) else 1
    if length == 0:
        raise ProviderError(ERR_UNSUPPORTED, "fusion: zero-length execution not supported")
    out_shape = normalize_scalar_shape(out_shape)
    scalar_shape = normalize_scalar_shape([1] * len(out_shape))
    prepared, owned = _prepare(prov, values, scalar_shape)
    try:
        shader = plan.generate_wgsl_for_outputs(list(output_ids), _scalar_ty(prov))
        if len(output_ids) == 1:
            outs = [prov.fused_elementwise(shader, prepared, out_shape, length)]
        else:
            outs = prov.fused_elementwise_multi(shader, prepared, out_shape, length, len(output_ids))
    finally:
        for h in owned:
            prov.free(h)
    return outs


def execute_reduction(prov, plan: FusionGroupPlan, data_vid: int, values: Sequence, reduce_len: int, num_slices: int,
                      axis: int = 0, omitnan: bool = False, flavor: Optional[ReductionFlavor] = None,
                      workgroup_size: int = 256) -> GpuTensorHandle:
    """fusion_exec.rs:464-628: output shape is [num_slices]; geometry comes from the VM
    (crates/runmat-vm/src/accel/fusion.rs:540-915)."""
    if reduce_len * num_slices == 0:
        raise ProviderError(ERR_UNSUPPORTED, "fusion: zero-length execution not supported")
    flavor = flavor or ReductionFlavor.Sum()
    prepared, owned = _prepare(prov, values, (1, 1))
    try:
        shader = plan.generate_reduction_wgsl(data_vid, _scalar_ty(prov), axis=axis, omitnan=omitnan, is_mean=flavor.kind == "mean")
        wg = workgroup_size or prov.default_reduction_workgroup_size()
        return prov.fused_reduction(shader, prepared, (num_slices,), reduce_len, num_slices, wg, flavor)
    finally:
        for h in owned:
            prov.free(h)
