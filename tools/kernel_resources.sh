#!/bin/bash
# Prints registers, spills and scratch of every kernel in a built object (offline, no GPU needed):
#   tools/kernel_resources.sh raytracer_amd/csrc/build/rt_sweep_grouped.o
# The object is a host ELF with the gfx950 code object in its .hip_fatbin section.
set -e
obj=${1:-raytracer_amd/csrc/build/rt_sweep_grouped.o}
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
llvm=/opt/rocm/lib/llvm/bin
$llvm/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" "$tmp/fatbin"
tgt=$($llvm/clang-offload-bundler --list --type=o --input="$tmp/fatbin" | grep gfx950)
$llvm/clang-offload-bundler --unbundle --type=o --input="$tmp/fatbin" --targets="$tgt" --output="$tmp/k.co"
$llvm/llvm-readelf --notes "$tmp/k.co" | python3 -c '
import re, sys, subprocess
txt = sys.stdin.read()
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    agpr = blk.split()[0]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"^void \(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    rows.append((name, g("vgpr_count"), agpr, g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
import os, json
if os.environ.get("KERNEL_RESOURCES_JSON"):
    print(json.dumps([dict(zip(("name", "vgpr", "agpr", "vgpr_spill", "sgpr", "sgpr_spill", "scratch", "lds"), r)) for r in rows]))
else:
    print("%-62s %5s %5s %7s %5s %7s %8s %6s" % ("kernel", "vgpr", "agpr", "vspill", "sgpr", "sspill", "scratchB", "ldsB"))
    for r in rows:
        print("%-62s %5s %5s %7s %5s %7s %8s %6s" % r)
'
