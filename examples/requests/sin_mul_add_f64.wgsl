const MAX_RANK: u32 = 128u;
struct PackedValue { value: u32, _pad0: u32, _pad1: u32, _pad2: u32 };
alias PackedArray = array<PackedValue, MAX_RANK>;

struct Tensor { data: array<f64>, };
struct Params {
    len: u32,
    offset: u32,
    rank: u32,
    _pad: u32,
    out_shape: PackedArray,
    in0_shape: PackedArray,
    in0_stride: PackedArray,
    in1_shape: PackedArray,
    in1_stride: PackedArray,
    in2_shape: PackedArray,
    in2_stride: PackedArray,
}

fn isNan(x: f64) -> bool { let bits = bitcast<u64>(x); return (bits & 0x7ff0000000000000u) == 0x7ff0000000000000u && (bits & 0x000fffffffffffffu) != 0u; }
fn isFinite(x: f64) -> bool { return (x == x) && (abs(x) < f64(1.7976931348623157e308)); }
fn isInf(x: f64) -> bool { return (x == x) && !(abs(x) < f64(1.7976931348623157e308)); }
fn hypot(a: f64, b: f64) -> f64 {
    let lo = min(abs(a), abs(b));
    let hi = max(abs(a), abs(b));
    if hi == f64(0.0) { return f64(0.0); }
    if isInf(hi) { return hi; }
    let r = lo / hi;
    return hi * sqrt(f64(1.0) + r * r);
}

@group(0) @binding(0) var<storage, read> input0: Tensor;
@group(0) @binding(1) var<storage, read> input1: Tensor;
@group(0) @binding(2) var<storage, read> input2: Tensor;
@group(0) @binding(3) var<storage, read_write> output: Tensor;
@group(0) @binding(4) var<uniform> params: Params;

@compute @workgroup_size(@WG@)
fn main(@builtin(global_invocation_id) gid: vec3<u32>) {
    let idx = gid.x;
    if (idx >= params.len) { return; }
    let g = idx + params.offset;
    var coord: array<u32, MAX_RANK>;
    var tmp: u32 = g;
    var d: u32 = 0u;
    loop { if d >= params.rank { break; } let dim = params.out_shape[d].value; if dim == 0u { coord[d] = 0u; } else { coord[d] = tmp % dim; tmp = tmp / dim; } d = d + 1u; }
    var i0: u32 = 0u; d = 0u; loop { if d >= params.rank { break; } let sd = params.in0_shape[d].value; let st = params.in0_stride[d].value; let c = select(coord[d], 0u, sd == 1u); i0 = i0 + c * st; d = d + 1u; }
    var i1: u32 = 0u; d = 0u; loop { if d >= params.rank { break; } let sd = params.in1_shape[d].value; let st = params.in1_stride[d].value; let c = select(coord[d], 0u, sd == 1u); i1 = i1 + c * st; d = d + 1u; }
    var i2: u32 = 0u; d = 0u; loop { if d >= params.rank { break; } let sd = params.in2_shape[d].value; let st = params.in2_stride[d].value; let c = select(coord[d], 0u, sd == 1u); i2 = i2 + c * st; d = d + 1u; }
    let tmp0: f64 = sin(input0.data[i0]);
    let tmp1: f64 = (tmp0 * input1.data[i1]);
    let tmp2: f64 = (tmp1 + input2.data[i2]);
    output.data[g] = tmp2;
}
