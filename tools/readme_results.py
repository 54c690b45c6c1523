#!/usr/bin/env python3
"""README.md's results table from the round's committed bench line: every number in the README between the
`<!-- results:begin -->` / `<!-- results:end -->` markers comes from profiles/<round>_bench.json (python bench.py --extras) and nowhere else.
usage: tools/readme_results.py profiles/r06_bench.json [--write]"""
import json, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1])
d = json.loads(src.read_text())
ex = d.get("extras", {})


def row(name, label, e):
    t = e.get("traffic")
    return (f"| {label} | {e['kernel_ms']:.4g} | {e['msamples_per_s'] / 1e3:.0f} | {e['mpixels_per_s']:.0f} | {e['roofline_frac']:.3f} | "
            f"{(t / 1e9):.2f} GB |" if t else
            f"| {label} | {e['kernel_ms']:.4g} | {e['msamples_per_s'] / 1e3:.0f} | {e['mpixels_per_s']:.0f} | {e['roofline_frac']:.3f} | — |") + f" {e['kernel'].replace('raymarch_', '').replace('_kernel', '')} |"


head = {"kernel_ms": d["kernel_ms"], "msamples_per_s": d["value"], "mpixels_per_s": d["mpixels_per_s"], "roofline_frac": d["roofline"]["frac"],
        "traffic": d["roofline"].get("traffic"), "kernel": d["config"]["kernel"]}
LABELS = [
    ("cfg3 1024³ u16 @ 1080p, NEAREST, default camera (**headline**)", head),
    ("cfg3 without the 12-bit packed copy", ex.get("headline_without_pack12")), ("cfg3 shallow (α = 1, early ray termination)", ex.get("shallow_alpha1_ert")),
    ("cfg3 off-axis pose (zenith 60°, azimuth 45°)", ex.get("offaxis_deep")), ("cfg3 MIP", ex.get("mip_deep")),
    ("cfg3 view top / bottom", None), ("cfg3 TRILINEAR", ex.get("trilinear_deep")), ("cfg3 TRILINEAR off-axis", ex.get("trilinear_offaxis_deep")),
    ("cfg3 TRILINEAR MIP", ex.get("trilinear_mip_deep")), ("cfg1 shape 256³ u8 @ 720p, α = 1", ex.get("cfg1_shape")), ("cfg1 shape TRILINEAR", ex.get("cfg1_shape_trilinear")),
    ("cfg2 shape 512×512×452 u16 @ 1080p, ERT + window", ex.get("cfg2_shape_ert_window")), ("cfg2 shape TRILINEAR", ex.get("cfg2_shape_ert_window_trilinear")),
    ("cfg4 2048³ u8 @ 4K, grey", ex.get("cfg4_grey")), ("cfg4 MIP", ex.get("cfg4_mip")), ("cfg4 TF + empty-space skipping", ex.get("cfg4_tf_skip")),
    ("cfg4 TRILINEAR grey", ex.get("cfg4_grey_trilinear")), ("cfg4 TRILINEAR + TF + skipping (north-star's config 4)", ex.get("cfg4_tf_skip_trilinear")),
]
lines = [f"Source: `{src.relative_to(ROOT) if src.is_absolute() else src}` (`python bench.py --extras`, one MI355X, kernel-only HIP-event ms at sustained clocks; every row's frame "
         "checked bit for bit against the CPU oracle on sampled rows; traffic = PMC HBM bytes per launch, `profiles/traffic.json`).", "",
         "| workload | kernel ms | Gsamples/s | Mpixels/s | HBM roofline fraction (algorithmic bytes) | HBM traffic | kernel |", "|---|---|---|---|---|---|---|"]
for label, e in LABELS:
    if label.startswith("cfg3 view top"):
        a, b = ex.get("view_top_deep"), ex.get("view_bottom_deep")
        if a and b:
            lines.append(f"| {label} | {a['kernel_ms']:.4g} / {b['kernel_ms']:.4g} | {a['msamples_per_s'] / 1e3:.0f} / {b['msamples_per_s'] / 1e3:.0f} | {a['mpixels_per_s']:.0f} / {b['mpixels_per_s']:.0f} | "
                         f"{a['roofline_frac']:.3f} / {b['roofline_frac']:.3f} | " + (f"{a['traffic'] / 1e9:.2f} / {b['traffic'] / 1e9:.2f} GB" if a.get('traffic') and b.get('traffic') else "—")
                         + f" | {a['kernel'].replace('raymarch_', '').replace('_kernel', '')} |")
        continue
    if e:
        lines.append(row("", label, e))
cb = d.get("cpu_baseline") or {}
if cb:
    lines += ["", f"CPU baseline of the same run (scalar oracle, `kind: port`): {cb['value']} Msamples/s on 1 core"
              + (f", {cb['all_cores']['value']} on {cb['all_cores']['cores']} threads" if cb.get("all_cores") else "")
              + f"; GPU frame bit-exact on the sampled rows: {cb.get('parity_bit_exact_on_sample')}.  `ms_per_step` {d['ms_per_step']} "
              f"(kernel {d['kernel_ms']}), roofline.frac {d['roofline']['frac']}, launch choices imported {d['config'].get('launch_choices_imported')}, "
              f"trial frames in the timed region {d['config'].get('trial_frames_in_timed_region')}."]
text = "\n".join(lines)
if "--write" in sys.argv:
    p = ROOT / "README.md"
    s = p.read_text()
    s = re.sub(r"<!-- results:begin -->.*?<!-- results:end -->", "<!-- results:begin -->\n" + text + "\n<!-- results:end -->", s, flags=re.S)
    p.write_text(s)
else:
    print(text)
