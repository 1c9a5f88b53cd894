"""Register / scratch / LDS figures of every kernel in a built library, read from the gfx950 code
object's metadata notes (what the hardware is actually given), not from profiler columns:

    python scripts/kernel_resources.py [lib.so] [--json out.json] [--filter substr]

rocprofv3's ``VGPR_Count`` column is the granulated allocation of ONE of the register files and its
``LDS_Block_Size`` the static segment only; the numbers that decide occupancy and spilling are the
notes' ``.vgpr_count``, ``.agpr_count``, ``.vgpr_spill_count``, ``.sgpr_spill_count``,
``.private_segment_fixed_size`` (scratch bytes per lane) and ``.group_segment_fixed_size`` (static LDS;
the dynamic part is requested at launch: ``guide_wave2_lds`` etc.).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object_notes(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        co = os.path.join(tmp, "gfx950.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "x")],
                       check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        return subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def kernels(lib):
    notes = code_object_notes(lib)
    recs, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key == "agpr_count" or (key == "args" and cur is None):
            pass
        if ln.lstrip().startswith("- .") and key in ("agpr_count", "args"):
            cur = {}
            recs.append(cur)
        if cur is not None and key in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count",
                                       "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
                                       "max_flat_workgroup_size", "wavefront_size"):
            cur[key] = val if key == "name" else int(val)
    recs = [r for r in recs if "name" in r]
    for r, d in zip(recs, demangle([r["name"] for r in recs])):
        r["kernel"] = re.sub(r"^void ", "", d).split("(")[0]
    return recs


def main():
    args = [a for a in sys.argv[1:]]
    out_json = flt = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = args[0] if args else os.path.join(root, "crispr-bean_amd", "lib", "libbean_hip.so")
    recs = kernels(lib)
    if flt:
        recs = [r for r in recs if flt in r["kernel"]]
    recs.sort(key=lambda r: r["kernel"])
    print(f"{'kernel':78s} vgpr agpr sgpr vspill sspill scratchB ldsB")
    for r in recs:
        print(f"{r['kernel'][:78]:78s} {r.get('vgpr_count', 0):4d} {r.get('agpr_count', 0):4d} {r.get('sgpr_count', 0):4d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_spill_count', 0):6d} "
              f"{r.get('private_segment_fixed_size', 0):8d} {r.get('group_segment_fixed_size', 0):5d}")
    if out_json:
        json.dump({r["kernel"]: {k: v for k, v in r.items() if k not in ("kernel", "name")} for r in recs},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
