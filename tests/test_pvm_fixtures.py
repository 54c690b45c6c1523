"""PVM/DDS decoder against golden fixtures minted with the REFERENCE's own
codec (tests/golden/make_pvm_fixtures.py, run in the authoring container).  CPU only."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
MANIFEST = json.loads((GOLDEN / "pvm_manifest.json").read_text())


@pytest.mark.parametrize("case", MANIFEST, ids=[c["file"] for c in MANIFEST])
def test_decoder_matches_reference_codec(vra, case):
    out = vra.read_pvm_volume(GOLDEN / case["file"])
    assert out is not None
    data, dims, scale = out
    assert dims == (case["width"], case["height"], case["depth"], case["components"])
    assert scale == pytest.approx(tuple(case["scale"]))
    assert data.size == case["payload_bytes"]
    assert hashlib.sha256(data.tobytes()).hexdigest() == case["sha256"]
    assert vra.checksum(data) == case["ref_checksum"]          # the reference's checksum() value
    if "payload" in case:
        assert np.array_equal(data, np.load(GOLDEN / case["payload"]))


def test_v3e_header_and_block_interleave_is_exercised():
    big = [c for c in MANIFEST if c["payload_bytes"] > (1 << 24)]
    assert big and (GOLDEN / big[0]["file"]).read_bytes()[:8] == b"DDS v3e\n"
    small = [c for c in MANIFEST if c["payload_bytes"] < (1 << 24)]
    assert all((GOLDEN / c["file"]).read_bytes()[:8] == b"DDS v3d\n" for c in small)


def test_uncompressed_pvm_and_error_paths(vra, tmp_path):
    vox = np.arange(24, dtype=np.uint8)
    p = tmp_path / "plain.pvm"
    p.write_bytes(b"PVM\n# a comment line\n4 3 2\n1\n" + vox.tobytes())
    data, dims, scale = vra.read_pvm_volume(p)
    assert dims == (4, 3, 2, 1) and np.array_equal(data, vox) and scale == (1.0, 1.0, 1.0)
    p.write_bytes(b"PVM2\n4 3 2\n1 0.5 2\n1\n" + vox.tobytes())
    data, dims, scale = vra.read_pvm_volume(p)
    assert scale == (1.0, 0.5, 2.0)
    p.write_bytes(b"PVM3\n4 3 2\n1 1 1\n1\n" + vox.tobytes() + b"desc\0court\0\0comment\0")
    assert vra.read_pvm_volume(p)[1] == (4, 3, 2, 1)
    for bad in (b"PVX\n4 3 2\n1\n" + vox.tobytes(),            # wrong magic
                b"PVM\n4 3 2\n1\n" + vox.tobytes()[:-1],        # short payload
                b"PVM\n4 3 2\n1\n" + vox.tobytes() + b"x",      # trailing garbage (size check of ddsbase.cpp:838)
                b"PVM\n0 3 2\n1\n",                             # non-positive dimension
                b"PVM2\n4 3 2\n1 0 1\n1\n" + vox.tobytes(),     # non-positive scale
                b"PVM"):
        p.write_bytes(bad)
        assert vra.read_pvm_volume(p) is None
    assert vra.read_pvm_volume(tmp_path / "nope.pvm") is None


def test_checksum_known_values(vra):
    assert vra.checksum(np.zeros(10, dtype=np.uint8)) == 0
    # cipher_1 = 271*1 + 1 = 272, sum = 272; cipher_2 = 271*272 + 2 = 73714, sum += 147428
    assert vra.checksum(np.array([1, 2], dtype=np.uint8)) == 272 + 2 * 73714
