#!/usr/bin/env python3
"""Average VALU issue cost (cycles per wave-instruction per SIMD) of each ray-march kernel family's HOT LOOPS, from the
family's ISA and the per-instruction costs measured on the MI355X (tools/ubench/valu_rates2.hip,
profiles/r03_valu_rates2.txt).  Runs in the build container (hipcc cross-compiles the probe instances of tools/isa/);
writes profiles/valu_cpi.json, which bench.py multiplies with the measured SQ_INSTS_VALU of a launch
(profiles/valu.json, tools/pmc_valu.py) for the `roofline_valu` figure:

    VALU-issue fraction = SQ_INSTS_VALU x cpi / (1024 SIMDs x shader cycles of the launch)

Cost classes (cycles per wave-instruction per SIMD, sustained, six wavefronts per SIMD):
    2.3   v_add/sub/mul_f32, v_add/sub_u32, v_and/or/xor_b32, v_ashrrev_i32, v_lshrrev_b32, v_mov_b32
    2.8   v_fma_f32 / v_fmac_f32 / v_mac_f32
    4.3   everything else: v_max/min/med3, v_fract, every conversion, v_lshlrev, v_bfe, 24-bit multiplies, every
          three-operand integer op, v_cndmask, every v_cmp, cross-lane ops
    4.2   an instruction of the first two classes with an SGPR operand (fp32 and integer alike)
Output modifiers (clamp, mul:2 ...) and inline constants cost nothing (profiles/r04_valu_rates2.txt).
A hot loop = a loop at the kernel's deepest nesting level that contains the tap loads (LDS reads for the staged kernel,
buffer / global loads for the others)."""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_ashrrev_i32", "v_lshrrev_b32", "v_mov_b32", "v_add_i32", "v_sub_i32", "v_not_b32", "v_nop"}
FMA = {"v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_fmaak_f32", "v_fmamk_f32"}
PROBES = {
    "raymarch_tslab_kernel": ("probe_tslab.hip", ["-DPROBE_NW=8", "-DPROBE_LDSKB=80", "-DPROBE_PERM=false"], r"ds_read_u(16|8)"),
    "raymarch_tslab_kernel_half": ("probe_tslab.hip", ["-DPROBE_NW=8", "-DPROBE_LDSKB=80", "-DPROBE_PERM=true"], r"ds_read_u(16|8)"),
    "raymarch_fast_kernel": ("probe_headline.hip", [], r"buffer_load_(ushort|ubyte|sbyte|short)"),
    "raymarch_fast_kernel_big": ("probe_cfg4.hip", [], r"global_load_(ubyte|ushort)|buffer_load_(ushort|ubyte)"),
    "raymarch_tri_kernel": ("probe_tri.hip", [], r"buffer_load_(dword|ushort|ubyte)"),
    "raymarch_relay_kernel": ("probe_relay.hip", [], r"buffer_load_(ushort|ubyte)"),
}


def cost(mnem, operands):
    base = mnem.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base in FAST or base in FMA:
        # an SGPR source on a fast-rate instruction -- fp32 add / mul / fma and integer add alike (round 4: v_add_u32 with an SGPR
        # operand 4.14 cycles) -- issues at the slow rate (inline constants do not)
        srcs = operands.split(",")[1:]
        if any(re.match(r"\s*-?\|?s(\d+|\[)", s) for s in srcs):
            return 4.2, "sgpr"
        return (2.3, "fast") if base in FAST else (2.8, "fma")
    return 4.3, "slow"


def hot_loops(asm, kernel_substr, tap_re):
    """[(label, [instruction lines])] of the deepest loops of the kernel that contain tap loads"""
    lines = asm.splitlines()
    # the kernel's text: from its label to its .end_amdhsa_kernel / s_endpgm tail
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kernel_substr + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section") or lines[i].strip().startswith(".end_amdhsa_kernel"))
    blocks, cur, depth = [], None, {}
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks.append([cur, 0, []])
            d = re.search(r"Depth=(\d+)", l)
            blocks[-1][1] = int(d.group(1)) if d else 0
            continue
        d = re.search(r"; (?:=>)?\s*(?:This )?(?:Inner )?Loop Header: Depth=(\d+)|in Loop: Header=\S+ Depth=(\d+)", l)
        if d and blocks:
            blocks[-1][1] = max(blocks[-1][1], int(d.group(1) or d.group(2)))
            continue
        s = l.strip()
        if blocks and s and not s.startswith(";") and not s.startswith("."):
            blocks[-1][2].append(s.split(";")[0].strip())
    # merge consecutive blocks of one nesting level into loops; keep the ones with tap loads; deepest level that has any
    for level in range(max(b[1] for b in blocks), 0, -1):
        loops, run = [], []
        for b in blocks + [["end", -1, []]]:
            if b[1] >= level:
                run.append(b)
            elif run:
                ins = [i for bb in run for i in bb[2]]
                if any(re.match(tap_re, i) for i in ins):
                    loops.append((run[0][0], ins))
                run = []
        if loops:
            return loops
    return []


def main():
    out = {}
    for family, (probe, defs, tap_re) in PROBES.items():
        asm_path = Path("/tmp") / f"valu_cpi_{family}.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only",
               "-S", "-o", str(asm_path), str(ROOT / "tools" / "isa" / probe)] + defs
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        kern = family.replace("_half", "").replace("_big", "")
        loops = hot_loops(asm_path.read_text(), kern, tap_re)
        tot, n, mix, salu, lds = 0.0, 0, {"fast": 0, "fma": 0, "slow": 0, "sgpr": 0}, 0, 0
        for _, ins in loops:
            for i in ins:
                parts = i.split(None, 1)
                mnem, ops = parts[0], parts[1] if len(parts) > 1 else ""
                if mnem.startswith("v_"):
                    c, k = cost(mnem, ops)
                    tot += c; n += 1; mix[k] += 1
                elif mnem.startswith("s_") and not mnem.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")):
                    salu += 1
                elif mnem.startswith("ds_"):
                    lds += 1
        out[family] = {"cpi": round(tot / max(n, 1), 3), "hot_loops": len(loops), "valu_per_loop_copy": round(n / max(len(loops), 1), 1),
                       "salu_per_loop_copy": round(salu / max(len(loops), 1), 1), "lds_per_loop_copy": round(lds / max(len(loops), 1), 1),
                       "mix": {k: round(v / max(n, 1), 3) for k, v in mix.items()}, "probe": f"tools/isa/{probe} {' '.join(defs)}".strip()}
        print(family, out[family], flush=True)
    dst = ROOT / "profiles" / "valu_cpi.json"
    sys.path.insert(0, str(ROOT))
    from bench import kernel_source_hash

    dst.write_text(json.dumps({"kernel_source_hash": kernel_source_hash(), "costs": "profiles/r03_valu_rates2.txt", "families": out}, indent=1) + "\n")


if __name__ == "__main__":
    main()
