#!/bin/bash
# Development aid: compile conv2d_h8.hip with extra flags ($@) into build_ab/ and list registers / spills of the conv_h8_kernel instantiations
# whose mangled name contains $H8_FILTER (default: the 8-wave 128-channel configuration).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
( cd semanticlidarunc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I../../include -I. -c ${H8_SRC:-conv2d_h8.hip} -o ../../build_ab/${H8_OBJ:-conv2d_h8.o} -Rpass-analysis=kernel-resource-usage 2> ../../build_ab/res.txt ) || { tail -30 build_ab/res.txt; exit 1; }
python3 - <<'PY'
import os, re
t = open('build_ab/res.txt').read()
flt = os.environ.get('H8_FILTER', 'ELi2ELi2ELi4ELi2ELb0ELb0')
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print(name[27:100], 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'SGPRspill', g('SGPRs Spill'), 'VGPRspill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
PY
