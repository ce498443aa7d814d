#!/usr/bin/env python3
"""Developer aid: per-kernel register use and instruction histogram from `make -C swarmkit_amd/csrc asm`."""
import re
import sys
from collections import Counter

d = '/tmp/swp_asm/'
txt = open(d + 'resource_usage.txt').read()
pat = sys.argv[1] if len(sys.argv) > 1 else 'resolve1'
for b in re.split(r'Function Name: ', txt)[1:]:
    name = b.split()[0]
    if pat not in name and 'r1_' not in name:
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(name[:56].ljust(56), 'VGPR', g('VGPRs'), 'SGPR', g('TotalSGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'),
          'spillV', g('VGPRs Spill'))
s = open(d + 'swp_engine-hip-amdgcn-amd-amdhsa-gfx950.s').read()
sym = sys.argv[2] if len(sys.argv) > 2 else '_ZN6swpdev10k_resolve1ILi3ELi8EEEvNS_11ResolveArgsE'
i = s.index(sym + ':')
j = s.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in s[i:] else len(s)
lines = s[i:j].split('\n')
c = Counter()
for l in lines:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
print(sum(c.values()), 'instrs in', sym)
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print('  ', k, v)
print('vmcnt(0):', sum(1 for l in lines if 'vmcnt(0)' in l), ' any waitcnt:', sum(1 for l in lines if 's_waitcnt' in l))
