"""The evidence chain's own tools on synthetic profiler output (no GPU): tools/pmc_traffic.py and tools/pmc_valu.py must attribute
counters to the kernel the bench line names -- round 3's tool quoted the relay kernel's six tuning launches for the headline -- and
bench.py must refuse a figure that belongs to another kernel or to other kernel sources."""
import csv
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
FAST_A = "void vr::raymarch_fast_kernel<unsigned short, 1, 0, 0, false, true, true, true, 0, false, 8, true, true, true>(vr::FrameParams)"
FAST_B = "void vr::raymarch_fast_kernel<unsigned short, 1, 0, 0, false, true, true, true, 0, false, 8, true, true, false>(vr::FrameParams)"
RELAY = "void vr::raymarch_relay_kernel<unsigned short, 1, 0, true, true, true, true, true, 0, 0>(vr::FrameParams)"
TSLAB = "void vr::raymarch_tslab_kernel<unsigned short, 0, 0, true, 0, 8, 80, false, 4>(vr::FrameParams)"


def write_pass(out, counter, rows):
    d = out / f"pmc_{counter}"
    d.mkdir(parents=True)
    with open(d / "run_counter_collection.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for name, n, value in rows:
            for _ in range(n):
                w.writerow([name, counter, value])


def run(tool, *args, profiles):
    env_script = (f"import sys, runpy; sys.argv = {[str(ROOT / 'tools' / tool), *map(str, args)]!r}; "
                  f"import pathlib; import importlib; sys.path.insert(0, {str(ROOT / 'tools')!r}); "
                  f"runpy.run_path({str(ROOT / 'tools' / tool)!r}, run_name='__main__')")
    return subprocess.run([sys.executable, "-c", env_script], capture_output=True, text=True, cwd=profiles)


@pytest.fixture
def sandbox(tmp_path, monkeypatch):
    """the tools write ROOT/profiles/*.json: run them against copies so the committed evidence is untouched"""
    keep = {n: (ROOT / "profiles" / n).read_text() if (ROOT / "profiles" / n).exists() else None for n in ("traffic.json", "valu.json")}
    yield tmp_path
    for n, text in keep.items():
        if text is None:
            (ROOT / "profiles" / n).unlink(missing_ok=True)
        else:
            (ROOT / "profiles" / n).write_text(text)


def test_traffic_is_attributed_to_the_kernel_the_bench_line_names(sandbox):
    out = sandbox / "prof"
    out.mkdir()
    (out / "FETCH_SIZE.log").write_text('noise\n{"metric": "x", "config": {"kernel": "raymarch_fast_kernel"}}\n')
    # the settled instance (135 launches), the other loop's tuning launches, the relay kernel's six LAST in the file
    write_pass(out, "FETCH_SIZE", [(FAST_A, 135, 1000.0), (FAST_B, 22, 1100.0), (RELAY, 6, 2000.0)])
    write_pass(out, "WRITE_SIZE", [(FAST_A, 135, 10.0), (FAST_B, 22, 10.0), (RELAY, 6, 10.0)])
    p = run("pmc_traffic.py", out, "unit_test_key", profiles=sandbox)
    assert p.returncode == 0, p.stderr
    e = json.loads((ROOT / "profiles" / "traffic.json").read_text())["unit_test_key"]
    assert e["kernel"] == "raymarch_fast_kernel" and e["instance"] == FAST_A and e["launches"] == 135
    assert e["bytes"] == int(2 * 1000.0 * 1024 + 10.0 * 1024)
    # the JSON line names the LAST launch's kernel: when that family never settles (a third of the launches) the figure belongs to
    # the instance with more than half of the launches and says so (bench.py then refuses it for a line that names another kernel)
    out2 = sandbox / "prof2"
    out2.mkdir()
    (out2 / "a.log").write_text('{"config": {"kernel": "raymarch_relay_kernel"}}\n')
    write_pass(out2, "FETCH_SIZE", [(FAST_A, 135, 1000.0), (RELAY, 6, 2000.0)])
    write_pass(out2, "WRITE_SIZE", [(FAST_A, 135, 10.0), (RELAY, 6, 10.0)])
    p = run("pmc_traffic.py", out2, "unit_test_key2", profiles=sandbox)
    assert p.returncode == 0, p.stderr
    e = json.loads((ROOT / "profiles" / "traffic.json").read_text())["unit_test_key2"]
    assert e["kernel"] == "raymarch_fast_kernel" and e["instance"] == FAST_A and e["bytes"] == int(2 * 1000.0 * 1024 + 10.0 * 1024)
    # ... and a run without a settled kernel at all gives no figure
    out4 = sandbox / "prof4"
    out4.mkdir()
    (out4 / "a.log").write_text('{"config": {"kernel": "raymarch_relay_kernel"}}\n')
    write_pass(out4, "FETCH_SIZE", [(FAST_A, 60, 1000.0), (FAST_B, 60, 1000.0), (RELAY, 20, 2000.0)])
    write_pass(out4, "WRITE_SIZE", [(FAST_A, 60, 10.0), (FAST_B, 60, 10.0), (RELAY, 20, 10.0)])
    p = run("pmc_traffic.py", out4, "unit_test_key4", profiles=sandbox)
    assert p.returncode != 0 and "not the settled kernel" in (p.stderr + p.stdout)
    assert "unit_test_key4" not in json.loads((ROOT / "profiles" / "traffic.json").read_text())
    # the bench line's alias for the staged trilinear kernel
    out3 = sandbox / "prof3"
    out3.mkdir()
    (out3 / "a.log").write_text('{"config": {"kernel": "raymarch_tslab_kernel"}}\n')
    for c, v in (("FETCH_SIZE", 1700.0), ("WRITE_SIZE", 30.0), ("SQ_INSTS_VALU", 8.0e8), ("GRBM_GUI_ACTIVE", 2.4e7)):
        write_pass(out3, c, [(TSLAB, 100, v), ("void vr::raymarch_tri_kernel<unsigned short, 1, 0, 0, true, 0, true>(vr::FrameParams)", 8, v * 3)])
    assert run("pmc_traffic.py", out3, "unit_test_key3", profiles=sandbox).returncode == 0
    assert run("pmc_valu.py", out3, "unit_test_key3", profiles=sandbox).returncode == 0
    v = json.loads((ROOT / "profiles" / "valu.json").read_text())["unit_test_key3"]
    assert v["kernel"] == "raymarch_tslab_kernel" and v["valu_wave_insts"] == 8.0e8 and v["shader_cycles"] == 2.4e7 / 8


def test_bench_refuses_a_figure_of_another_kernel_or_of_other_sources(sandbox):
    sys.path.insert(0, str(ROOT))
    import bench

    h = bench.kernel_source_hash()
    t = ROOT / "profiles" / "traffic.json"
    d = json.loads(t.read_text()) if t.exists() else {}
    d["unit_a"] = {"bytes": 123, "kernel": "raymarch_fast_kernel", "kernel_source_hash": h}
    d["unit_b"] = {"bytes": 456, "kernel": "raymarch_fast_kernel", "kernel_source_hash": "0" * 16}
    t.write_text(json.dumps(d))
    assert bench.traffic_entry("unit_a", "raymarch_fast_kernel") == 123
    assert bench.traffic_entry("unit_a", "raymarch_relay_kernel") is None        # counters of another kernel than the one launched
    assert bench.traffic_entry("unit_b", "raymarch_fast_kernel") is None         # measured on other kernel sources
    assert bench.traffic_entry("missing", "raymarch_fast_kernel") is None and bench.traffic_entry(None, "x") is None
    v = ROOT / "profiles" / "valu.json"
    dv = json.loads(v.read_text()) if v.exists() else {}
    dv["unit_a"] = {"kernel": "raymarch_fast_kernel", "instance": FAST_A, "valu_wave_insts": 2.0e8, "shader_cycles": 1.0e6, "kernel_source_hash": h}
    v.write_text(json.dumps(dv))
    r = bench.valu_roofline("unit_a", "raymarch_fast_kernel", 0.46)
    cpi = json.loads((ROOT / "profiles" / "valu_cpi.json").read_text())["families"]["raymarch_fast_kernel"]["cpi"]
    assert r["bound"] == "valu" and abs(r["frac"] - 2.0e8 * cpi / 1024 / 1.0e6) < 1e-3
    assert bench.valu_roofline("unit_a", "raymarch_relay_kernel", 0.46) is None
