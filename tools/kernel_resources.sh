#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel in liboa_icp.so's device code (the compiler's own report).
# usage: tools/kernel_resources.sh [name-filter-regex]
cd "$(dirname "$0")/../object_alignment_amd/csrc" || exit 1
# (every translation unit of the default library: the host unit's plain kernels + the kernel families, csrc/oa_families.hpp)
for unit in oa_icp.hip $(ls oa_fam_*.hip | grep -v oa_fam_exp); do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
    --cuda-device-only -c $unit -o /tmp/oa_dev.o -Rpass-analysis=kernel-resource-usage 2>&1
done | python3 -c "
import sys, re, subprocess
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
cur = None; d = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m: cur = m.group(1); d[cur] = {}
    for k in ['VGPRs', 'VGPR Spill', 'ScratchSize', 'Occupancy', 'LDS Size', 'SGPRs']:
        m = re.search(re.escape(k) + r'[^:]*: (\d+)', ln)
        if m and cur and k not in d[cur]: d[cur][k] = m.group(1)
for k, v in d.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.split('(')[0].strip()
    if flt is None or flt.search(name):
        print('%-50s %s' % (name[:50], ' '.join('%s=%s' % (a.replace(' ', '_'), b) for a, b in v.items())))
" "$@"
