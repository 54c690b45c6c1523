"""profiles/launch_choices.bin (vr_export_choices; loaded by bench.py into every renderer) is only trusted by the library build that
measured it: its header carries a hash of every source under volume-renderer_amd/csrc (the Makefile's VR_BUILD_ID).  This test
recomputes that hash from the tree: a kernel edit without a fresh evidence pass (tools/round_evidence.sh) makes the blob stale -- it
would be ignored at run time (bench.py then explores during its untimed settle phase), and this test says so before that happens."""
import hashlib
import struct
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
BLOB = ROOT / "profiles" / "launch_choices.bin"
CSRC = ROOT / "volume-renderer_amd" / "csrc"


def source_build_id() -> int:
    files = sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")) + list(CSRC.glob("*.h")), key=lambda p: str(p.relative_to(CSRC.parent)))
    h = hashlib.sha256()
    for f in files:
        h.update(f.read_bytes())
    return int(h.hexdigest()[:16], 16)


def test_the_committed_choices_blob_is_well_formed():
    if not BLOB.exists():
        pytest.skip("no committed blob: bench.py starts cold and settles during its untimed set-up")
    b = BLOB.read_bytes()
    assert b[:8] == b"VRCHOICE" and struct.unpack_from("<I", b, 8)[0] == 1
    n = struct.unpack_from("<I", b, 12)[0]
    assert len(b) == 88 + 56 * n and n >= 1
    for i in range(n):
        key, ncand, settled, heur = struct.unpack_from("<Qiii", b, 88 + 56 * i)
        cand = struct.unpack_from("<8i", b, 88 + 56 * i + 20)
        assert 2 <= ncand <= 8 and settled in cand[:ncand] and heur in cand[:ncand], i
    assert b[24:88].split(b"\0")[0].startswith(b"cc9.5 cu256")          # measured on an MI355X (gfx950, 256 CUs)


def test_the_committed_choices_blob_was_measured_by_these_sources():
    if not BLOB.exists():
        pytest.skip("no committed blob")
    blob_id = struct.unpack_from("<Q", BLOB.read_bytes(), 16)[0]
    assert blob_id == source_build_id(), ("profiles/launch_choices.bin was measured by another build of the library: re-run tools/round_evidence.sh "
                                          "(or delete the file) after changing anything under volume-renderer_amd/csrc")
