"""Register / scratch / LDS budget of every kernel of libdgcn.so, from the compiler's own remarks.

    python profiles/kernel_resources.py [--only gen_aggr_egemm] [--out profiles/r04_kernel_resources.md]

Compiles each csrc/*.hip with the flags of deep_gcns_torch_amd/build.py plus
-Rpass-analysis=kernel-resource-usage (device code only, no object kept) and tabulates the remarks per kernel
template.  Needs hipcc only: runs in the CPU container.
"""
from __future__ import annotations

import argparse
import re
import subprocess
import sys
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from deep_gcns_torch_amd import build as _build   # noqa: E402

FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "VGPRs Spill": "spill", "SGPRs Spill": "sspill",
          "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds",
          "TotalSGPRs": "sgpr"}


def remarks(src: Path):
    cmd = [_build.hipcc_path(), f"--offload-arch={_build.ARCH}", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", f"-I{_build.INCLUDE}", f"-I{_build.CSRC}", "-c", str(src),
           "-o", "/dev/null"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:[^:]+: )?Function Name: (.*?)(?: \[-Rpass-analysis.*)?$", line)
        if m:
            cur = {"mangled": m.group(1).strip(), "src": src.stem}
            kernels.append(cur)
            continue
        m = re.search(r"remark:\s+(?:[^:]+:\d+:\d+:\s+)?\s*([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return kernels


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def short(name):
    return name.replace("dgcn::(anonymous namespace)::", "").replace("dgcn::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--title", default="current build")
    ap.add_argument("--raw", default=None, help="also dump one line per instantiation (tsv) here")
    a = ap.parse_args()
    srcs = sorted(_build.CSRC.glob("*.hip"))
    if a.only:
        srcs = [s for s in srcs if s.stem == a.only]
    with ThreadPoolExecutor(max_workers=8) as ex:
        ks = [k for lst in ex.map(remarks, srcs) for k in lst]
    for k, d in zip(ks, demangle([k["mangled"] for k in ks])):
        k["name"] = d
    groups = defaultdict(list)
    for k in ks:
        base = re.sub(r"^void ", "", k["name"]).replace("dgcn::(anonymous namespace)::", "").replace("dgcn::", "")
        base = base.split("<")[0].split("(")[0]
        if "rocprim" in k["name"] or "hipcub" in k["name"]:
            base = "rocPRIM (scan / radix sort inside graph_build)"
        groups[(k["src"], base)].append(k)
    out = [f"# Register / scratch / LDS budget of every kernel in libdgcn.so ({a.title})", "",
           "`python profiles/kernel_resources.py`: `hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage` over",
           f"`deep_gcns_torch_amd/csrc/*.hip`: {len(ks)} kernels (template instantiations counted separately).  Spill = VGPRs "
           "spilled to scratch.", "",
           "| source | kernel | instantiations | VGPRs (min - max) | occupancy (waves/SIMD, min - max) | spilling instantiations |",
           "|---|---|---|---|---|---|"]
    spills = []
    for (src, base), lst in sorted(groups.items()):
        v = [k.get("vgpr", 0) for k in lst]
        o = [k.get("occ", 0) for k in lst]
        sp = [k for k in lst if k.get("spill", 0) > 0]
        if "rocPRIM" not in base:
            spills += sp
        out.append(f"| {src} | `{base}` | {len(lst)} | {min(v)} - {max(v)} | {min(o)} - {max(o)} | {len(sp)} |")
    out += ["", "Instantiations with VGPR spills:" if spills else "No instantiation of this library's own kernels spills a VGPR.", ""]
    for k in spills:
        out.append(f"* `{short(k['name'])}`: {k.get('vgpr')} VGPRs, {k.get('spill')} spilled ({k.get('scratch')} bytes of scratch per lane)")
    scr = [k for k in ks if k.get("scratch", 0) > 0 and k.get("spill", 0) == 0 and "rocprim" not in k["name"]]
    if scr:
        out += ["", "Scratch without register spills (indexed private arrays):", ""]
        out += [f"* `{short(k['name'])}`: {k.get('scratch')} bytes per lane" for k in scr]
    if a.raw:
        Path(a.raw).write_text("".join(f"{k['src']}\t{short(k['name'])}\t{k.get('vgpr')}\t{k.get('agpr')}\t{k.get('spill')}\t"
                                       f"{k.get('scratch')}\t{k.get('occ')}\t{k.get('lds')}\n" for k in ks))
    text = "\n".join(out) + "\n"
    if a.out:
        Path(a.out).write_text(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
