#!/usr/bin/env python3
"""List VGPR/SGPR/scratch/LDS of every kernel in an AMDGPU assembly file (hipcc -S --cuda-device-only)."""
import re, sys
txt = open(sys.argv[1]).read()
for blk in txt.split('  - .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    short = re.sub(r'HIP_vector_typeI(.)Lj2EE', r'\1', name)
    print(f"{short[:70]:70s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
