#!/bin/bash
# Registers / spills / LDS of every kernel of one translation unit (device code only):
#   tools/kernel_resources.sh zns_conv_fft.hip [pattern]
set -e
SRC=$(dirname "$0")/../zeronotesamba_amd/csrc/$1
TMP=${TMPDIR:-/tmp}/zns_kres_$$
mkdir -p "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c "$SRC" -o "$TMP/k.bundle"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/k.bundle" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/k.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/k.co" > "$TMP/k.notes"
python3 - "$TMP/k.notes" "${2:-.}" <<'PY'
import re, subprocess, sys
s = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
print("vgpr agpr vspill sspill lds scratch  kernel")
for e in re.split(r'\n\s+- \.agpr_count', s)[1:]:
    e = ".agpr_count" + e
    name = re.search(r'\.name:\s+(\S+)', e).group(1)
    g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, e).group(1))
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace("zns::", "")
    if pat.search(d):
        print("%4d %4d %6d %6d %6d %6d  %s" % (g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
                                              g('group_segment_fixed_size'), g('private_segment_fixed_size'), d[:170]))
PY
rm -rf "$TMP"
