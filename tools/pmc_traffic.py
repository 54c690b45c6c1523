#!/usr/bin/env python3
"""Turn the PMC passes of a tools/profile.sh / tools/extras_traffic.sh output directory into profiles/traffic.json
(HBM bytes per launch of the kernel the bench line names) -- the `roofline.traffic` field bench.py prints for the
same command.

HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024:
  * FETCH_SIZE / WRITE_SIZE are reported in KiB and collected in SEPARATE --pmc passes;
  * gfx950 correction (MI355X_MICROARCH.md "HBM"): this rocprofv3 tallies the L2's 128-B
    fabric read requests at 64 B, i.e. FETCH_SIZE is exactly half the bytes read.  Checked
    in our own access pattern: stats_kernel reads the linear 2 GiB volume once and reports
    1,048,608.8 KiB (profiles/r01_v1_linear_summary.txt), while TCC_EA0_RDREQ x 128 B gives
    the full 2 GiB.  WRITE_SIZE needed no correction (gen_volume_kernel: 2,097,152 KiB for
    2 GiB written).

Which kernel: the measured work model launches several ray-march kernels in one run (the candidates it tries
before it settles), so the counters are grouped by the FULL kernel name (every template argument), the family is
the one the run's own JSON line names (`config.kernel`, read from the pass logs; or the third argument), the
instance is the one with the most launches, and it must account for at least a third of the run's ray-march launches
(under the profiler the work model's measurements fail more often and are repeated; a configuration with six candidates
explores for ~25 launches) -- otherwise no entry is written (round 3 quoted the last `raymarch_` line of the text summary: the relay kernel's
six tuning launches instead of the settled fast kernel's 135).

usage: tools/pmc_traffic.py <profile output dir> <key> [kernel family]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict
from pathlib import Path



def kernel_from_logs(out):
    for f in sorted(glob.glob(os.path.join(out, "*.log"))):
        for line in Path(f).read_text(errors="replace").splitlines():
            if line.startswith("{") and '"config"' in line:
                try:
                    return json.loads(line)["config"]["kernel"]
                except (ValueError, KeyError):
                    pass
    return None


def per_kernel(out, counter):
    """{full kernel name: [values]} of one counter, from the pass that collected it"""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc




def settled(acc, family, key, counter):
    """(family, instance name, its values, ray-march launches of the run) of the settled kernel of one PMC pass: the most-launched
    instance of `family` when it has at least a third of the run's ray-march launches; otherwise -- the JSON line names the LAST
    launch's kernel, which under the profiler can be one of the work model's re-measurements -- the instance with more than half
    of the launches, whatever its family; otherwise none (exit)"""
    symbol = family            # bench.py's `config.kernel` IS the kernel's symbol (vr_last_kernel_name)
    march = {k: v for k, v in acc.items() if "raymarch_" in k}
    total = sum(len(v) for v in march.values())
    mine = {k: v for k, v in march.items() if re.search(r"\b" + re.escape(symbol) + r"\b", k)}
    name = max(mine, key=lambda k: len(mine[k])) if mine else None
    if name is not None and 3 * len(mine[name]) >= total:
        return family, name, mine[name], total
    top = max(march, key=lambda k: len(march[k])) if march else None
    if top is not None and 2 * len(march[top]) > total:
        sym = re.search(r"(raymarch_\w+)", top).group(1)
        return sym, top, march[top], total
    if name is None:
        sys.exit(f"{key}: no launches of {symbol} in the {counter} pass ({sorted(march)})")
    sys.exit(f"{key}: the most-launched {symbol} instance has {len(mine[name])} of {total} ray-march launches in the {counter} pass -- not the settled kernel")


def main():
    out, key = sys.argv[1], sys.argv[2]
    named = sys.argv[3] if len(sys.argv) > 3 else kernel_from_logs(out)
    if not named:
        sys.exit(f"{key}: no kernel family given and no bench JSON line in {out}/*.log")
    vals, picked, instances, family = {}, None, [], named
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        fam, name, v, total = settled(per_kernel(out, counter), named, key, counter)
        if picked is not None and fam != family:
            sys.exit(f"{key}: the passes settled on different kernel families ({family}, {fam})")
        family = fam
        # (two passes may settle on different instances of one family when two candidates tie -- the fast kernel's plain and
        # pipelined loops on a full frame: both are recorded)
        instances.append(name)
        picked = name if picked is None else picked
        vals[counter] = (sum(v) / len(v), len(v), total)
    traffic = int(2 * vals["FETCH_SIZE"][0] * 1024 + vals["WRITE_SIZE"][0] * 1024)
    root = Path(__file__).resolve().parent.parent
    dst = root / "profiles" / "traffic.json"
    d = json.loads(dst.read_text()) if dst.exists() else {}
    sys.path.insert(0, str(root))
    from bench import kernel_source_hash   # the figure is only valid for the kernels it was measured on

    d[key] = {"bytes": traffic, "kernel": family, "instance": picked, "instance_of_write_pass": instances[-1] if instances[-1] != picked else None, "launches": vals["FETCH_SIZE"][1], "raymarch_launches_in_run": vals["FETCH_SIZE"][2],
              "kernel_source_hash": kernel_source_hash(), "fetch_size_kib": round(vals["FETCH_SIZE"][0], 1), "write_size_kib": round(vals["WRITE_SIZE"][0], 1)}
    dst.write_text(json.dumps(d, indent=1) + "\n")
    print(key, family, traffic, {k: v[:2] for k, v in vals.items()})


if __name__ == "__main__":
    main()
