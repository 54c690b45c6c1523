"""Regression goldens of the oracle (tests/golden/oracle_goldens.json, minted by
tests/golden/make_oracle_goldens.py): BASELINE configs 0-2 at reduced probes.

CPU: the oracle still gives the frozen answers.  GPU: the HIP path, through the C ABI, gives the
same bits for the same rows.  These are NOT reference vectors -- those are tests/golden/ref_gl/ (the reference's
shader executed by a real GL, tests/test_ref_gl_goldens.py).  These freeze the oracle's answers where the reference
has no semantics to execute (TRILINEAR) and its per-pixel fetch counts, so oracle and kernels cannot drift
together or apart between rounds.  Re-minted in round 5 when the oracle adopted the executed reference's
arithmetic (normalize / dot order); the cfg0 frame hashes now equal the reference's own.
"""
import hashlib
import importlib.util
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).parent / "golden" / "oracle_goldens.json"
CASES = json.loads(GOLD.read_text())


def _maker():
    spec = importlib.util.spec_from_file_location("make_oracle_goldens", GOLD.with_name("make_oracle_goldens.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_frozen_answers(oracle, name):
    m = _maker()
    c = CASES[name]["case"]
    c["vol"] = tuple(tuple(v) if isinstance(v, list) else v for v in c["vol"])
    got = m.run_case(c)
    want = CASES[name]["expect"]
    assert got["samples"] == want["samples"] and got["pixels_with_alpha"] == want["pixels_with_alpha"]
    assert got["sha256_rows"] == want["sha256_rows"]
    for k, v in want["probe_rgba"].items():
        assert np.array_equal(np.float32(got["probe_rgba"][k]), np.float32(v)), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_reproduces_frozen_answers(vra, oracle, name):
    m = _maker()
    c = CASES[name]["case"]
    want = CASES[name]["expect"]
    vol = m.make_volume(tuple(tuple(v) if isinstance(v, list) else v for v in c["vol"]))
    W, H = c["img"]
    R = vra.renderer
    for layout in (R.LAYOUT_LINEAR, R.LAYOUT_BRICKED):
        with vra.RendererCore(0) as r:
            r.setup((W, H)); r.loadShader("VolumeRenderer.cs"); r.setQuirks(0)
            r.setLayout(layout)
            r.setVolume(vol)
            r.setWindow(*c["window"]); r.setAlpha(c["alpha"]); r.setMIP(bool(c.get("mip", 0)))
            r.setFilter(R.FILTER_TRILINEAR if c.get("filter", 0) else R.FILTER_NEAREST)
            r.render()
            frame = r.readPixels()
            _, spp = r.countSamples(per_pixel=True)
        rows = list(range(c["rows"] // 2, H, c["rows"]))
        h = hashlib.sha256()
        for y in rows:
            h.update(np.ascontiguousarray(frame[y]).tobytes())
        assert int(spp[rows].astype(np.int64).sum()) == want["samples"], (name, layout)
        assert h.hexdigest() == want["sha256_rows"], (name, layout)
