"""Register / scratch budgets the design depends on, checked at compile time (hipcc cross-compiles gfx950 without a GPU).

The performance claims in DESIGN.md rest on occupancy facts a later edit or compiler can silently break: the eight-wave pipelined
GEMM tiles run FOUR waves per SIMD (two 512-thread blocks per CU) only while they stay within 128 VGPRs; the LU panel keeps 64
doubles per row in registers and must not spill; none of the hot kernels may touch scratch.  hipcc's
`-Rpass-analysis=kernel-resource-usage` remarks give the numbers."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "runmat_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resources(unit: str) -> dict:
    if not Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fvisibility=hidden",
           f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", str(SRC / unit), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=SRC)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    pat = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", re.S)
    for m in pat.finditer(r.stderr):
        out[m.group(1)] = {"vgpr": int(m.group(2)), "agpr": int(m.group(3)), "scratch": int(m.group(4)), "occupancy": int(m.group(5))}
    assert out, "no resource remarks in the compiler output"
    return out


def _pick(res: dict, *needles: str) -> dict:
    hits = {k: v for k, v in res.items() if all(n in k for n in needles)}
    assert hits, f"no kernel matching {needles}"
    return hits


def test_dgemm_kernels_keep_their_register_budgets():
    res = _resources("dgemm.hip")
    # plain, transposed-A and short-epilogue (ELi1E) eight-wave tiles: two blocks per CU need <= 128 VGPRs (four waves per SIMD)
    for name, r in {**_pick(res, "k_dgemm_w8ILb0ELb0ELb0ELi0E"), **_pick(res, "k_dgemm_w8ILb0ELb0ELb0ELi1E"), **_pick(res, "k_dgemm_w8ILb0ELb1ELb0")}.items():
        assert r["vgpr"] + r["agpr"] <= 128 and r["scratch"] == 0 and r["occupancy"] >= 4, (name, r)
    # the guarded forms of the same three (shapes that are not whole tiles) must fit two blocks per CU as well
    # (k_dgemm_w8g asks for four waves per SIMD; a few spilled dwords outside the k loop are the price)
    for name, r in {**_pick(res, "k_dgemm_w8gILb0ELb0ELi0E"), **_pick(res, "k_dgemm_w8gILb0ELb0ELi1E"), **_pick(res, "k_dgemm_w8gILb0ELb1ELi0E"),
                    **_pick(res, "k_dgemm_w8gILb1ELb0ELi0E")}.items():
        assert r["vgpr"] + r["agpr"] <= 128 and r["scratch"] <= 64 and r["occupancy"] >= 4, (name, r)
    # every eight-wave variant: no scratch - except the one that calls the out-of-line epilogue (pow step, ELi2E)
    for name, r in _pick(res, "10k_dgemm_w8I").items():
        if "ELi2E" not in name:
            assert r["scratch"] == 0, (name, r)
    # four-wave kernels: two blocks per CU (<= 256 registers); only the epilogue variant (a noinline call) may use scratch
    for name, r in _pick(res, "7k_dgemmIL").items():
        assert r["vgpr"] + r["agpr"] <= 256, (name, r)
        if "k_dgemmILb1ELb1" not in name:
            assert r["scratch"] == 0, (name, r)
    for name, r in _pick(res, "k_dgemm_small").items():
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 128, (name, r)


def test_sgemm_kernels_keep_their_register_budgets():
    res = _resources("sgemm.hip")
    for name, r in _pick(res, "k_sgemm_w8").items():
        assert r["vgpr"] + r["agpr"] <= 128 and r["scratch"] == 0 and r["occupancy"] >= 4, (name, r)
    for name, r in _pick(res, "7k_sgemmIL").items():
        assert r["scratch"] == 0, (name, r)


def test_lu_kernels_do_not_spill():
    res = _resources("lu.hip")
    for key in ("k_lu_panel2", "k_trsm_lower_2p", "k_trsm_fused", "k_subst_chain", "k_laswp_lists"):
        for name, r in _pick(res, key).items():
            assert r["scratch"] == 0, (name, r)
    # the panel's register window: one wave per SIMD, everything in the 512-register file
    for name, r in _pick(res, "k_lu_panel2").items():
        assert r["vgpr"] + r["agpr"] <= 512, (name, r)
    # the two-pass solve must fit beside an update-stream dgemm block: few registers, 66 KiB of LDS (checked at launch)
    for name, r in _pick(res, "k_trsm_lower_2p").items():
        assert r["vgpr"] + r["agpr"] <= 96, (name, r)


def test_solve_path_chain_kernels_fit_beside_an_update_block():
    """Round 5's occupancy facts: the update streams' eight-wave block - the YIELD instantiation included - holds two
    waves of <= 144 registers per SIMD (288 of 512), and every chain kernel of the solve path must fit into the 224 that are left on
    the same SIMD, or it waits for a 260-us deep tile to retire: k_rp_top (one wave per SIMD), the matrix-core rows-below kernel and
    triangular solve (one-wave workgroups).  None may spill."""
    gemm = _resources("dgemm.hip")
    for name, r in {**_pick(gemm, "k_dgemm_w8ILb1ELb0ELb0ELi0ELb1E"), **_pick(gemm, "k_dgemm_w8ILb1ELb0ELb0ELi0ELb0E")}.items():
        assert r["vgpr"] + r["agpr"] <= 144 and r["scratch"] == 0, (name, r)
    lu = _resources("lu.hip")
    for key in ("8k_rp_topILb0E", "k_rp_below_mfma", "k_trsm_lower_mfma"):
        for name, r in _pick(lu, key).items():
            assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 224, (name, r)


def test_special_kernels_keep_their_register_budgets():
    res = _resources("special.hip")
    # the tall-skinny Gram kernel holds an 8 x 8 block of products and two rows of operands per thread: no spills, two workgroups per CU
    for name, r in _pick(res, "13k_gram_skinnyI").items():
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 256 and r["occupancy"] >= 2, (name, r)
    # the image passes stream: the statistics and the plain apply pass without scratch (the gamma step may call the out-of-line pow)
    for name, r in _pick(res, "k_plane_moments").items():
        assert r["scratch"] == 0, (name, r)


def test_reduction_and_small_solver_kernels_do_not_spill():
    res = _resources("reduce2.hip")
    # two-stage reductions with wide accumulators (arg-min / max: value + index; moments: four doubles) over f64 and f32 storage, and the
    # staged scan whose first wave keeps a tile of 64 steps in registers: none may touch scratch
    # (the moments accumulator's element-by-element path for a batch that holds a NaN keeps three to five doubles in scratch: 24 - 40 bytes, known since round 3 introduced it)
    for key in ("k_r2_contig_v2", "k_r2_strided_v2", "k_r2_short", "k_scan_lines_staged", "k_scan_chunks"):
        for name, r in _pick(res, key).items():
            assert r["scratch"] <= (48 if "MomAcc" in name else 0), (name, r)
    res = _resources("small_solve.hip")
    for name, r in _pick(res, "k_small_solve").items():
        assert r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 128, (name, r)
