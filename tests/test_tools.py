"""CPU tests of the build / measurement tooling of round 4: the exec-prologue lint (the gate behind the root cause of round 3's GPU fault) and the
roofline record of bench.py."""
import importlib.util
import json
import os
import sys

import pytest

from conftest import ROOT


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rows_of(lines):
    """(address, text, raw) rows like tools/check_exec_prologue.py's disassembly parser makes them; 4 bytes per instruction."""
    out = []
    for i, t in enumerate(lines):
        out.append((0x1000 + 4 * i, t.split("<")[0].strip(), t))
    return out


def test_lint_finds_a_vector_copy_ahead_of_the_exec_restore():
    """The defect of round 3: the block an s_cbranch_execz skips to STARTS with a vector instruction, ahead of the s_or_b64 that re-opens the mask."""
    lint = load("check_exec_prologue", "tools/check_exec_prologue.py")
    faulty = rows_of([
        "s_and_saveexec_b64 s[2:3], s[4:5]",
        "s_cbranch_execz 3 <k+0x10>",            # -> row 4
        "v_div_scale_f64 v[16:17], s[4:5], v[0:1], v[0:1], 1.0",
        "v_mul_f64 v[50:51], v[50:51], v[0:1]",
        "v_mov_b32_e32 v66, v100",               # +0x10: the misplaced live-range-split copy
        "s_or_b64 exec, exec, s[2:3]",
        "s_endpgm",
    ])
    hits = lint.check("k", faulty)
    assert len(hits) == 1 and "v_mov_b32_e32 v66, v100" in hits[0][1]
    # the same block with the copy behind the restore, and with only scalar work / lane reads ahead of it: clean
    clean = rows_of([
        "s_and_saveexec_b64 s[2:3], s[4:5]",
        "s_cbranch_execz 3 <k+0x10>",
        "v_div_scale_f64 v[16:17], s[4:5], v[0:1], v[0:1], 1.0",
        "v_mul_f64 v[50:51], v[50:51], v[0:1]",
        "v_readlane_b32 s2, v163, 39",
        "s_or_b64 exec, exec, s[2:3]",
        "v_mov_b32_e32 v66, v100",
        "s_endpgm",
    ])
    assert lint.check("k", clean) == []
    # a vector instruction ahead of the restore of ANOTHER region's mask (the tail of a then-block that falls into an outer join) is not the pattern
    outer = rows_of([
        "s_and_saveexec_b64 s[6:7], s[8:9]",
        "s_cbranch_execz 2 <k+0xc>",
        "v_add_f64 v[0:1], v[2:3], v[4:5]",
        "v_mov_b32_e32 v7, v68",                 # +0xc: join of the s[6:7] region ... whose restore comes first
        "s_or_b64 exec, exec, s[0:1]",
        "s_endpgm",
    ])
    assert lint.check("k", outer) == []


def test_lint_passes_the_shipped_objects():
    lint = load("check_exec_prologue", "tools/check_exec_prologue.py")
    objs = [os.path.join(ROOT, "build", f) for f in sorted(os.listdir(os.path.join(ROOT, "build"))) if f.endswith(".o")] if os.path.isdir(os.path.join(ROOT, "build")) else []
    if not objs:
        pytest.skip("no build/*.o in this checkout")
    for f in objs:
        for name, rows in lint.kernels(lint.disassemble(f)):
            assert lint.check(name, rows) == [], (f, name)


def test_roofline_record_is_a_fraction():
    """bench.py's roofline: useful FP64 flop (FMA = 2) x lane utilisation / kernel time against 78.6 TFLOP/s; HBM traffic with FETCH_SIZE doubled."""
    bench = load("bench", "bench.py")
    pmc = {"SQ_ACTIVE_INST_VALU": 2.7e9, "SQ_THREAD_CYCLES_VALU": 8.64e10, "SQ_INSTS_VALU": 2.64e9, "SQ_INSTS_SALU": 1.83e9, "SQ_INSTS_VALU_FMA_F64": 5.1e8,
           "SQ_INSTS_VALU_MUL_F64": 4.0e8, "SQ_INSTS_VALU_ADD_F64": 2.0e8, "SQ_INSTS_VALU_TRANS_F64": 2.1e7, "GRBM_GUI_ACTIVE": 1.24e8, "FETCH_SIZE": 2.2e6, "WRITE_SIZE": 4.8e6}
    r = bench.roofline_record(pmc, "test", 6.45, 61.09e9, 33177600)
    lane_util = 8.64e10 / (2.7e9 * 64)
    useful = (2 * 5.1e8 + 4.0e8 + 2.0e8) * 64 * lane_util
    assert r["bound"] == "fp64-valu" and r["peak"] == 78.6 and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - useful / 6.45e-3 / 1e12) < 1e-3 and abs(r["frac"] - r["achieved"] / 78.6) < 1e-4 and 0 < r["frac"] <= 1
    assert r["traffic"] == int((2 * 2.2e6 + 4.8e6) * 1024) and abs(r["valu_lane_util"] - lane_util) < 1e-4
    assert r["hbm_algorithmic"]["frac"] > 1          # the 8d line: kept as a side record, not bounded by 1
    none = bench.roofline_record(None, "no counters", 6.45, None, 33177600)
    assert none["frac"] is None and none["traffic"] is None and none["bound"] == "fp64-valu"


def test_committed_profile_is_refused_for_another_kernel_build(tmp_path, monkeypatch):
    bench = load("bench", "bench.py")
    assert len(bench.kernel_build_id()) == 16 and bench.kernel_build_id() == bench.kernel_build_id()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "hexray_amd" / "csrc")
    (tmp_path / "hexray_amd" / "csrc" / "k.hip").write_text("kernel source")
    counters = {c: 1.0 for _, cs in bench.PMC_PASSES for c in cs}
    (tmp_path / "profiles" / "pmc_live_meshes16.json").write_text(json.dumps({"kernel_build": "0123456789abcdef", "counters": counters}))
    got, why = bench.committed_pmc("meshes16")
    assert got is None and "not used" in why
    (tmp_path / "profiles" / "pmc_live_meshes16.json").write_text(json.dumps({"kernel_build": bench.kernel_build_id(), "counters": counters}))
    got, why = bench.committed_pmc("meshes16")
    assert got == counters and "kernel build" in why
