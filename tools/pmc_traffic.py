#!/usr/bin/env python3
"""Turn a tools/profile.sh summary into profiles/traffic.json (HBM bytes per launch of the
dominant kernel) -- the `roofline.traffic` field bench.py prints for the same command.

HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024:
  * FETCH_SIZE / WRITE_SIZE are reported in KiB and collected in SEPARATE --pmc passes;
  * gfx950 correction (MI355X_MICROARCH.md "HBM"): this rocprofv3 tallies the L2's 128-B
    fabric read requests at 64 B, i.e. FETCH_SIZE is exactly half the bytes read.  Checked
    in our own access pattern: stats_kernel reads the linear 2 GiB volume once and reports
    1,048,608.8 KiB (profiles/r01_v1_linear_summary.txt), while TCC_EA0_RDREQ x 128 B gives
    the full 2 GiB.  WRITE_SIZE needed no correction (gen_volume_kernel: 2,097,152 KiB for
    2 GiB written).
usage: tools/pmc_traffic.py <summary.txt> <key>
"""
import json
import re
import sys
from pathlib import Path

summary, key = sys.argv[1], sys.argv[2]
vals = {}
for line in Path(summary).read_text().splitlines():
    if "raymarch_" not in line:
        continue
    m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([0-9.]+)", line)
    if m and int(m.group(2)) > 1:      # the timed launches, not the single instrumented one
        vals[m.group(1)] = float(m.group(3))
traffic = int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)
out = Path(__file__).resolve().parent.parent / "profiles" / "traffic.json"
d = json.loads(out.read_text()) if out.exists() else {}
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import kernel_source_hash   # the figure is only valid for the kernels it was measured on

d[key] = {"bytes": traffic, "kernel_source_hash": kernel_source_hash(), "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"]}
out.write_text(json.dumps(d, indent=1) + "\n")
print(key, traffic, vals)
