"""CPU-side tests of the boundary: the C-ABI library loads and exports every symbol
include/rmhip.h declares, the WGSL front-end accepts exactly the reference planner's text
(crates/runmat-accelerate/src/fusion.rs:1525-2077, 2874-3026) and the generated HIP cross-compiles
for gfx950 with hipRTC -- no GPU and no compute calls involved."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(built):
    from runmat_amd import _lib

    header = (ROOT / "include" / "rmhip.h").read_text()
    declared = set(re.findall(r"RMHIP_API\s+[\w\s\*]+?\b(rmhip_\w+)\s*\(", header))
    assert len(declared) >= 35
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"librmhip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in lib.rmhip_version()


def test_init_fails_loudly_without_gpu(built):
    """No CPU fallback: on a box without a gfx950 device construction raises (on a GPU box it works)."""
    import torch

    from runmat_amd import HipProvider, ProviderError

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ProviderError) as e:
        HipProvider(0)
    assert e.value.code in (9, 4)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_package_never_touches_oracle():
    for f in (ROOT / "runmat_amd").rglob("*"):
        if f.suffix in (".py", ".cpp", ".h", ".hip") and f.is_file():
            text = f.read_text()
            assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f
            # ... nor the request emitter: what RunMat's planner sends is reproduced under tests/ only
            assert "planner_requests" not in text and "planner_exec" not in text and "FusionGroupPlan" not in text, f


def test_rust_float_display():
    from planner_requests import rust_f64_display as d

    assert d(2.0) == "2" and d(0.25) == "0.25" and d(-0.1) == "-0.1" and d(10.0) == "10"
    assert d(1e21) == "1000000000000000000000" and d(1e-7) == "0.0000001"
    assert d(float("inf")) == "inf" and d(float("nan")) == "NaN" and d(1.5e300).startswith("15")


def test_sin_mul_add_shader_text_and_translation(built):
    from runmat_amd import wgsl_translate
    from planner_requests import sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    sh = plan.generate_wgsl_for_output(out, "f64")
    # the exact body lines the reference emits (fusion.rs:1730,1754)
    assert "    let tmp0: f64 = sin(input0.data[i0]);\n" in sh
    assert "    let tmp1: f64 = (tmp0 * input1.data[i1]);\n" in sh
    assert "    let tmp2: f64 = (tmp1 + input2.data[i2]);\n" in sh
    assert "    output.data[g] = tmp2;\n" in sh
    assert "@group(0) @binding(3) var<storage, read_write> output: Tensor;" in sh
    assert "@group(0) @binding(4) var<uniform> params: Params;" in sh
    src = wgsl_translate(sh)
    assert "const double tmp0 = rm_sin(x0);" in src  # rm_sin = sin for f64 storage, the short form for f32 (skel_common.h)
    assert "const double tmp1 = (tmp0 * x1);" in src and "const double tmp2 = (tmp1 + x2);" in src
    assert "rm_ew_fast" in src and "rm_ew_bcast" in src


def test_parity_rewrites_log10_log1p_expm1(built):
    """fusion.rs:3005-3019 emits lossy forms; the CPU builtins use libm log10/ln_1p/exp_m1."""
    from runmat_amd import wgsl_translate
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    x = p.input()
    a = p.builtin("log10", x)
    b = p.builtin("log1p", a)
    c = p.builtin("expm1", b)
    sh = p.generate_wgsl_for_output(c)
    assert "(log(input0.data[i0]) * f64(0.4342944819032518))" in sh
    assert "log(tmp0 + f64(1.0))" in sh and "(exp(tmp1) - f64(1.0))" in sh
    src = wgsl_translate(sh)
    assert "tmp0 = log10(x0);" in src and "tmp1 = log1p(tmp0);" in src and "tmp2 = expm1(tmp1);" in src
    # a user-written log(x+1) arrives as two tmps and must NOT be rewritten
    q = FusionGroupPlan()
    x, one = q.input(), q.input()
    s = q.primitive("Add", x, one)
    l = q.builtin("log", s)
    src2 = wgsl_translate(q.generate_wgsl_for_output(l))
    assert "log1p" not in src2 and "tmp1 = log(tmp0);" in src2


def test_full_vocabulary_translates_and_compiles(built):
    """Every function of builtin_expr / primitive_expr (fusion.rs:2874-3026) in one plan."""
    from runmat_amd import wgsl_compile_check, wgsl_translate
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    x, y = p.input(), p.input()
    vals = []
    for f in ("sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh", "exp", "log", "log2", "sqrt",
              "abs", "exp2", "floor", "ceil", "round", "trunc", "asinh", "acosh", "atanh", "isfinite", "isinf",
              "isnan", "fix", "sign", "pow2", "heaviside", "single", "double", "log10", "log1p", "expm1"):
        vals.append(p.builtin(f, x))
    for f in ("atan2", "hypot", "max", "min", "mod", "rem"):
        vals.append(p.builtin(f, x, y))
    for op in ("Add", "Sub", "Mul", "ElemMul", "ElemDiv", "ElemLeftDiv", "Pow", "ElemPow"):
        vals.append(p.primitive(op, x, y))
    vals.append(p.primitive("Neg", x))
    vals.append(p.primitive("UPlus", x))
    acc = vals[0]
    for v in vals[1:]:
        acc = p.primitive("Add", acc, v)
    sh = p.generate_wgsl_for_output(acc)
    src = wgsl_translate(sh)
    for needle in ("rm_sign(", "rm_max(", "rm_min(", "rm_isnan(", "rm_isinf(", "rm_isfinite(", "hypot(", "atan2(",
                   "pow(", "trunc(", "round(", "exp2(", "log10(", "log1p(", "expm1("):
        assert needle in src, needle
    wgsl_compile_check(sh)  # hipRTC, --offload-arch=gfx950


def test_multi_output_shader(built):
    from runmat_amd import wgsl_compile_check, wgsl_translate
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    a, b = p.input(), p.input()
    s = p.primitive("Add", a, b)
    m = p.primitive("ElemMul", s, b)
    sh = p.generate_wgsl_for_outputs([m, s])
    assert "output0.data[g] = tmp1;" in sh and "output1.data[g] = tmp0;" in sh
    src = wgsl_translate(sh)
    assert "double& o0, double& o1" in src and "o0 = tmp1;" in src and "o1 = tmp0;" in src
    wgsl_compile_check(sh)


@pytest.mark.parametrize("axis", [0, 1])
def test_reduction_shader(built, axis):
    from runmat_amd import wgsl_compile_check, wgsl_translate
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    x, w = p.input(), p.input()
    two = p.constant(2.0)
    v = p.primitive("Add", p.primitive("ElemMul", p.builtin("sin", x), w), two)
    sh = p.generate_reduction_wgsl(v, "f64", axis=axis, omitnan=(axis == 1), is_mean=True)
    assert "let val: f64 = ((sin(v) * v1) + f64(2));" in sh  # constants inlined via Display (fusion.rs:1839-1873)
    assert ("const OMITNAN: bool = true" in sh) == (axis == 1)
    src = wgsl_translate(sh, "reduction")
    assert "return ((rm_sin(v0) * v1) + (0x1p+1));" in src
    assert "rm_red_contig" in src and "rm_red_strided" in src and "rm_red_final" in src
    wgsl_compile_check(sh, "reduction")


@pytest.mark.parametrize("mutate,needle", [
    (lambda s: s.replace("array<f64>", "array<f32>"), "unsupported scalar type"),  # storage and let types disagree
    (lambda s: s.replace("sin(input0.data[i0])", "frobnicate(input0.data[i0])"), "unsupported function"),
    (lambda s: s.replace("(tmp0 * input1.data[i1])", "(tmp7 * input1.data[i1])"), "tmp used before definition"),
    (lambda s: s.replace("input2.data[i2]", "input9.data[i9]"), "input index out of range"),
    (lambda s: s.replace("    output.data[g] = tmp2;\n", ""), "no output store"),
    (lambda s: s.replace("(tmp1 + input2.data[i2]);", "(tmp1 + input2.data[i2]) extra;"), "trailing tokens"),
])
def test_front_end_is_strict(built, mutate, needle):
    """Anything outside the subset is an error (-> the caller's CPU fallback), never a guess."""
    from runmat_amd import ProviderError, wgsl_translate
    from planner_requests import sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    sh = mutate(plan.generate_wgsl_for_output(out))
    with pytest.raises(ProviderError) as e:
        wgsl_translate(sh)
    assert e.value.code == 6 and needle in str(e.value)


def test_literals_are_exact(built):
    from runmat_amd import wgsl_translate
    from planner_requests import FusionGroupPlan

    p = FusionGroupPlan()
    x = p.input()
    c1, c2, c3 = p.constant(0.1), p.constant(-3.0), p.constant(float("inf"))
    v = p.primitive("Add", p.primitive("ElemMul", x, c1), p.primitive("Sub", c2, c3))
    sh = p.generate_reduction_wgsl(v, "f64", axis=0)
    assert "((v * f64(0.1)) + (f64(-3) - f64(inf)))" in sh
    src = wgsl_translate(sh, "reduction")
    assert (0.1).hex() in src and "(-0x1.8p+1)" in src and "__builtin_inf()" in src


def test_runtime_broadcast_shape_rules():
    """fusion_exec.rs:216-245"""
    import numpy as np

    from planner_exec import normalize_scalar_shape, runtime_broadcast_shape
    from runmat_amd.provider import GpuTensorHandle

    h = GpuTensorHandle((4, 1), 1, 1)
    assert runtime_broadcast_shape([h, np.zeros((1, 3)), 2.0]) == (4, 3)
    assert runtime_broadcast_shape([np.zeros((2, 3, 4)), np.zeros((4,))]) == (2, 3, 4)  # trailing alignment
    assert runtime_broadcast_shape([np.zeros((2, 3)), np.zeros((3, 2))]) is None
    assert runtime_broadcast_shape([1.0, 2]) == ()
    assert runtime_broadcast_shape(["x"]) is None
    assert normalize_scalar_shape(()) == (1, 1) and normalize_scalar_shape((5,)) == (5, 1) and normalize_scalar_shape((2, 3)) == (2, 3)


def test_f32_shaders_lower_to_f32_storage_with_f64_arithmetic(built):
    """The planner emits `scalar_ty = f32` for an F32 provider (fusion.rs:1525-1533): the generated kernels read and
    write `float` (four per 16-byte vector) while the body stays `double` -- the CPU path computes `single` arrays in
    f64 and rounds once (elementwise/times.rs:750-760)."""
    from runmat_amd import wgsl_compile_check, wgsl_translate
    from planner_requests import FusionGroupPlan, elementwise_math_plan, sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    sh = plan.generate_wgsl_for_output(out, "f32")
    assert "data: array<f32>" in sh and "let tmp0: f32 = sin(input0.data[i0]);" in sh
    src = wgsl_translate(sh, "elementwise")
    assert "const float* __restrict__ in0" in src and "float* __restrict__ out0" in src
    assert "typedef float rm_v4f" in src and "const rm_v4f a0_t" in src
    assert "const double tmp0 = rm_sin(x0);" in src  # rm_sin = sin for f64 storage, the short form for f32 (skel_common.h)           # arithmetic unchanged
    assert "(double)a0_t.w" in src and "r0_t.w = (float)q0;" in src
    assert "threadIdx.x < (n & 3)" in src                  # up to three tail elements
    wgsl_compile_check(sh, "elementwise")
    plan2, outs2 = elementwise_math_plan()
    wgsl_compile_check(plan2.generate_wgsl_for_output(outs2, "f32"), "elementwise")

    p = FusionGroupPlan()
    x, w = p.input(), p.input()
    v = p.primitive("Add", p.primitive("ElemMul", p.builtin("sin", x), w), p.constant(2.0))
    for axis in (0, 1):
        rs = p.generate_reduction_wgsl(v, "f32", axis=axis, omitnan=False, is_mean=True)
        assert "let val: f32 = ((sin(v) * v1) + 2.0);" in rs  # constants print as `{:?}` of the f32 value (fusion.rs:1839-1873)
        rsrc = wgsl_translate(rs, "reduction")
        assert "const float* __restrict__ in0" in rsrc and "const double v0 = (double)in0[idx * m0];" in rsrc
        wgsl_compile_check(rs, "reduction")


def test_front_end_survives_mangled_shaders(built):
    """`Err`, never a crash: truncations, deletions, duplications and byte flips of valid requests (elementwise and
    reduction, f64 and f32) either translate or come back as a COMPILE error (the caller then takes its CPU path)."""
    import random

    from runmat_amd import ProviderError, wgsl_translate
    from planner_requests import FusionGroupPlan, elementwise_math_plan, sin_mul_add_plan

    plan, out = sin_mul_add_plan()
    plan2, out2 = elementwise_math_plan()
    q = FusionGroupPlan()
    x, w = q.input(), q.input()
    v = q.primitive("Add", q.primitive("ElemMul", q.builtin("sin", x), w), q.constant(2.0))
    seeds = [(plan.generate_wgsl_for_output(out, "f64"), "elementwise"), (plan2.generate_wgsl_for_output(out2, "f32"), "elementwise"),
             (q.generate_reduction_wgsl(v, "f64", axis=0), "reduction"), (q.generate_reduction_wgsl(v, "f32", axis=1, omitnan=True), "reduction")]
    rng = random.Random(20260927)
    outcomes = {"ok": 0, "err": 0}
    for text, kind in seeds:
        body_at = text.index("let ")  # mutate around the statements the front-end actually reads
        for _ in range(120):
            s = text
            op = rng.randrange(5)
            pos = rng.randrange(body_at, len(s))
            if op == 0:
                s = s[:pos]
            elif op == 1:
                s = s[:pos] + s[pos + rng.randrange(1, 40):]
            elif op == 2:
                s = s[:pos] + s[pos:pos + rng.randrange(1, 60)] * 2 + s[pos:]
            elif op == 3:
                s = s[:pos] + rng.choice("()[];,.:=+-*/<>&|!0123456789abcxyz \n") + s[pos + 1:]
            else:
                a, b = sorted((pos, rng.randrange(body_at, len(s))))
                s = s[:a] + s[b:] + s[a:b]
            try:
                src = wgsl_translate(s, kind)
                assert "rm_" in src
                outcomes["ok"] += 1
            except ProviderError as e:
                assert e.code == 6, (e.code, str(e))  # RMHIP_ERR_COMPILE
                outcomes["err"] += 1
    assert outcomes["err"] > 100 and outcomes["ok"] > 0, outcomes


def _split_params(text: str):
    """Top-level comma split of a C / Rust parameter list (no nested parentheses in this ABI, brackets may occur)."""
    text = text.strip()
    if not text or text == "void":
        return []
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def test_bindings_agree_with_the_header_on_every_arity():
    """The ctypes table (runmat_amd/_abi.py) and the Rust FFI block (shim/rmhip_sys.rs) are generated from include/rmhip.h by
    scripts/gen_bindings.py; this is the independent check of that generator: parameter counts must match the header for
    every function a mirror declares (the Rust side cannot be compiled in this image)."""
    from runmat_amd import _lib

    header = (ROOT / "include" / "rmhip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {m.group(1): _split_params(m.group(2))
              for m in re.finditer(r"RMHIP_API\s+[\w\s\*]+?\b(rmhip_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.S)}
    assert len(protos) >= 55, len(protos)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        assert len(argtypes) == len(protos[name]), (name, len(argtypes), protos[name])
    shim = (ROOT / "shim" / "rmhip_sys.rs").read_text()
    block = shim[shim.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    rust = {m.group(1): _split_params(m.group(2)) for m in re.finditer(r"fn\s+(rmhip_\w+)\s*\(([^;]*?)\)\s*(?:->\s*[\w\s\*]+)?;", block, flags=re.S)}
    assert len(rust) >= 25, len(rust)
    for name, params in rust.items():
        assert name in protos, f"the Rust shim declares {name}, which include/rmhip.h does not"
        assert len(params) == len(protos[name]), (name, params, protos[name])


def _c_struct_fields(header: str, tag: str):
    """Field names, in declaration order, of `typedef struct <tag> { ... }` in the (comment-stripped) header."""
    body = re.search(r"typedef\s+struct\s+" + tag + r"\s*\{(.*?)\}", header, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.sub(r"\[.*?\]", "", first).split()[-1].lstrip("*"))
        names += [re.sub(r"\[.*?\]", "", r).strip().lstrip("*") for r in rest]
    return names


def test_struct_mirrors_have_the_headers_fields_in_order():
    """The by-value structs of the ABI are mirrored in ctypes and in the Rust shim: same fields, same order."""
    from runmat_amd import _lib

    header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "rmhip.h").read_text(), flags=re.S)
    pairs = {"rmhip_device_info": _lib.DeviceInfo, "rmhip_telemetry": _lib.Telemetry, "rmhip_matmul_epilogue": _lib.MatmulEpilogue,
             "rmhip_linsolve_options": _lib.LinsolveOptions, "rmhip_image_normalize": _lib.ImageNormalize, "rmhip_view": _lib.View}
    for tag, cls in pairs.items():
        assert [n for n, *_ in cls._fields_] == _c_struct_fields(header, tag), tag
    shim = (ROOT / "shim" / "rmhip_sys.rs").read_text()
    for tag, rust_name in (("rmhip_image_normalize", "RmhipImageNormalize"), ("rmhip_linsolve_options", "RmhipLinsolveOptions")):
        body = re.search(r"struct\s+" + rust_name + r"\s*\{(.*?)\}", shim, flags=re.S).group(1)
        fields = [f.split(":")[0].replace("pub ", "").strip() for f in body.replace("\n", " ").split(",") if ":" in f]
        assert fields == _c_struct_fields(header, tag), (tag, fields)


def test_rust_shim_op_codes_are_the_header_enums():
    """shim/rmhip_sys.rs (generated) repeats the op enums as constants (Rust cannot include the C header): every value must match."""
    header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "rmhip.h").read_text(), flags=re.S)
    values = {}
    for body in re.findall(r"enum\s+rmhip_\w+\s*\{(.*?)\}", header, flags=re.S):
        v = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = (x.strip() for x in item.split("="))
                v = int(val)
            else:
                name, v = item, v + 1
            values[name] = v
    shim = (ROOT / "shim" / "rmhip_sys.rs").read_text()
    consts = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"const\s+(RMHIP_\w+)\s*:\s*c_int\s*=\s*(\d+)\s*;", shim))
    assert len(consts) >= 60, len(consts)
    for name, v in consts.items():
        assert values.get(name) == v, (name, v, values.get(name))
    # every constant the hook tables use exists, and every hook name is a method of the reference trait's families
    hooks = (ROOT / "shim" / "hip_provider.rs").read_text()
    used = set(re.findall(r"=>\s*(RMHIP_\w+)", hooks)) | set(re.findall(r"self\.(?:unary|truth)\((RMHIP_\w+)", hooks))
    assert len(used) >= 70, len(used)
    assert used <= set(consts), used - set(consts)


def test_every_environment_knob_is_documented():
    """Every RMHIP_* variable the library reads with getenv is named in INTEGRATION.md, DESIGN.md or docs/*.md (developer knobs included:
    a maintainer who meets one in a bug report must be able to look it up)."""
    import re

    src = ROOT / "runmat_amd" / "csrc"
    knobs = set()
    for f in list(src.glob("*.cpp")) + list(src.glob("*.hip")) + list(src.glob("*.h")):
        knobs.update(re.findall(r'getenv\("(RMHIP_[A-Z0-9_]+)"\)', f.read_text()))
    assert len(knobs) > 20
    docs = (ROOT / "INTEGRATION.md").read_text() + (ROOT / "DESIGN.md").read_text() + "".join(f.read_text() for f in sorted((ROOT / "docs").glob("*.md")))
    missing = sorted(k for k in knobs if k not in docs)
    assert not missing, f"undocumented environment variables: {missing}"
