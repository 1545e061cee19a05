#!/bin/bash
# Register / LDS / scratch use of the kernels in the built gfx950 code object:  bash tools/kernel_regs.sh [name filter]
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section=.hip_fatbin=$T/fat.bin "$(dirname "$0")/../bsarec_amd/libbsarec_hip.so" $T/stripped.so
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$L/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - "$T/notes.txt" "${1:-}" <<'P'
import re, subprocess, shutil, sys
t = open(sys.argv[1]).read()
filt = shutil.which('c++filt')
for k in re.split(r'\n\s*- \.agpr_count:', t)[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    g = lambda f: re.search(r'\.' + f + r':\s+(\d+)', k).group(1)
    dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
    dn = re.sub(r'\(.*', '', dn).replace('void ', '')
    if sys.argv[2] and sys.argv[2] not in dn:
        continue
    print(f"vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5}  {dn[:130]}")
P
rm -rf $T
