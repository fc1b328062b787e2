#!/usr/bin/env python3
"""Per-kernel resource table of the built product library: name, VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs,
scratch bytes per lane, LDS bytes, workgroup size and the occupancy these allow (workgroups per CU) -- read from
the gfx950 code object's metadata note (`llvm-readelf --notes`), i.e. from what actually ships in
pragma-dsp_amd/csrc/libpdsp_hip.so, not from a compiler log.

    python tools/kernel_resources.py [--lib PATH] [--csv profiles/r03_kernel_resources.csv]

Also importable: kernels(lib) -> list of dicts (tests/test_kernel_resources_cpu.py fails when a kernel on a
BASELINE-config or default drop-in path uses scratch)."""
from __future__ import annotations

import argparse
import csv
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "pragma-dsp_amd", "csrc", "libpdsp_hip.so")
FIELDS = ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
          "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size")


def demangle(names):
    filt = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = "c++filt"
    try:
        out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        res = out.splitlines()
        if len(res) == len(names):
            return res
    except Exception:  # noqa: BLE001
        pass
    return list(names)


def short_name(demangled: str) -> str:
    """`void pdsp::k<...>(args)` -> `k<...>` with the pdsp:: qualifiers dropped."""
    s = demangled
    if s.startswith("void "):
        s = s[5:]
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):  # the argument list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return s[:cut].replace("pdsp::", "")


def waves_per_simd(vgprs: int) -> int:
    alloc = max(8, -(-vgprs // 8) * 8)  # 8-register granule, 512 per SIMD lane (MI355X_MICROARCH.md, Register files)
    return min(8, 512 // alloc)


def kernels(lib: str = DEFAULT_LIB):
    """One record per kernel NAME.  The library is linked from several translation units, each with a code object of
    its own in .hip_fatbin (one offload bundle per unit, concatenated); a template kernel that two units instantiate
    appears in both -- the same source compiled with the same flags -- and is listed once, with the LARGER of any
    resource figure should they ever differ (`units` says how many code objects carry it)."""
    MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
    notes = ""
    with tempfile.TemporaryDirectory(prefix="pdsp_kres_") as td:
        fat = os.path.join(td, "fat.bin")
        # (an explicit output file: with the input alone llvm-objcopy rewrites the library IN PLACE)
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib,
                        os.path.join(td, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, a in enumerate(starts):
            one, co = os.path.join(td, f"bundle{i}.bin"), os.path.join(td, f"dev{i}.co")
            with open(one, "wb") as f:
                f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={one}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True,
                                    check=True).stdout + "\n"
    recs, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" and line.lstrip().startswith("- "):  # first key of a kernel record
            cur = {}
            recs.append(cur)
        if cur is None:
            continue
        if key in FIELDS:
            cur[key] = int(val)
        elif key == "name":
            cur["mangled"] = val
    recs = [r for r in recs if "mangled" in r]
    merged = {}
    for r in recs:
        m = merged.setdefault(r["mangled"], dict(r, units=0))
        m["units"] += 1
        for k in FIELDS:
            m[k] = max(m.get(k, 0), r.get(k, 0))
    recs = list(merged.values())
    for r, d in zip(recs, demangle([r["mangled"] for r in recs])):
        r["kernel"] = short_name(d)
        wg = r.get("max_flat_workgroup_size", 256)
        waves_wg = max(1, wg // 64)
        by_regs = (waves_per_simd(r["vgpr_count"] + r.get("agpr_count", 0)) * 4) // waves_wg
        lds = r["group_segment_fixed_size"]
        by_lds = (160 * 1024) // lds if lds else 99
        r["workgroups_per_cu"] = max(0, min(by_regs, by_lds, 32 // waves_wg))
    recs.sort(key=lambda r: r["kernel"])
    return recs


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--spills-only", action="store_true")
    args = ap.parse_args()
    recs = kernels(args.lib)
    cols = ["kernel", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
            "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size", "workgroups_per_cu"]
    if args.csv:
        with open(args.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "vgprs", "agprs", "sgprs", "vgpr_spills", "sgpr_spills", "scratch_bytes_per_lane",
                        "lds_bytes", "workgroup_size", "workgroups_per_cu"])
            for r in recs:
                w.writerow([r.get(c, 0) for c in cols])
    bad = [r for r in recs if r["private_segment_fixed_size"] or r["vgpr_spill_count"]]
    print(f"{len(recs)} kernels, {os.path.getsize(args.lib)} bytes; {len(bad)} with scratch", file=sys.stderr)
    for r in (bad if args.spills_only else recs):
        print(f'{r["vgpr_count"]:4d} v {r["sgpr_count"]:4d} s  spill {r["vgpr_spill_count"]:3d}  scratch {r["private_segment_fixed_size"]:4d} B  '
              f'lds {r["group_segment_fixed_size"]:6d}  wg/cu {r["workgroups_per_cu"]}  {r["kernel"]}')
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
