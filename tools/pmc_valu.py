#!/usr/bin/env python3
"""VALU-issue figure of the kernel a bench line names, from the PMC passes of a tools/profile.sh / tools/extras_traffic.sh
output directory -> profiles/valu.json (bench.py: `roofline_valu`).

    wave-instructions = SQ_INSTS_VALU (per launch, summed over the chip)
    shader cycles     = GRBM_GUI_ACTIVE / 8   (rocprofv3 sums the counter over the eight XCDs: profiles/r03_headline_summary.txt,
                        8.71 M for a 0.46-ms launch = 1.09 M cycles = 2.36 GHz)
    fraction          = wave-instructions x cpi / (1024 SIMDs x shader cycles), cpi per kernel family from the ISA of its hot
                        loops and the measured per-instruction costs (tools/valu_cpi.py -> profiles/valu_cpi.json)

Kernel selection as in tools/pmc_traffic.py: grouped by the full kernel name, the family the run's own JSON line names, the
most-launched instance, at least a third of the run's ray-march launches (else the instance with more than half of them).
usage: tools/pmc_valu.py <profile output dir> <key> [kernel family]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic import kernel_from_logs, per_kernel, settled


def main():
    out, key = sys.argv[1], sys.argv[2]
    named = sys.argv[3] if len(sys.argv) > 3 else kernel_from_logs(out)
    if not named:
        sys.exit(f"{key}: no kernel family given and no bench JSON line in {out}/*.log")
    vals, picked, family = {}, None, named
    for counter in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
        fam, name, v, _ = settled(per_kernel(out, counter), named, key, counter)
        if picked is not None and fam != family:
            sys.exit(f"{key}: the passes settled on different kernel families ({family}, {fam})")
        family = fam
        picked = name if picked is None else picked
        vals[counter] = (sum(v) / len(v), len(v))
    # counter-only figures (no cost model), when the pass that holds them exists (tools/profile.sh collects them for the
    # headline): SQ_ACTIVE_INST_VALU -- "cycles each wave spends executing VALU instructions", quad-cycle units per the
    # rocprofv3 counter list --, SQ_WAIT_INST_ANY and SQ_WAVE_CYCLES (same units)
    extra = {}
    for counter in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum",
                    "SQ_INSTS_VMEM_RD"):
        try:
            acc = per_kernel(out, counter)
            if picked in acc and acc[picked]:
                extra[counter] = round(sum(acc[picked]) / len(acc[picked]), 1)
        except SystemExit:
            pass
        except Exception:
            pass
    root = Path(__file__).resolve().parent.parent
    dst = root / "profiles" / "valu.json"
    d = json.loads(dst.read_text()) if dst.exists() else {}
    sys.path.insert(0, str(root))
    from bench import kernel_source_hash

    d[key] = {"kernel": family, "instance": picked, "launches": vals["SQ_INSTS_VALU"][1], "valu_wave_insts": round(vals["SQ_INSTS_VALU"][0], 1),
              "shader_cycles": round(vals["GRBM_GUI_ACTIVE"][0] / 8.0, 1), "kernel_source_hash": kernel_source_hash()}
    if extra:
        d[key]["counters"] = extra
    dst.write_text(json.dumps(d, indent=1) + "\n")
    print(key, family, d[key])


if __name__ == "__main__":
    main()
