#!/bin/bash
# The unrolled matcher (wv::match_seq64, csrc/swp_wave.hpp) enters its 64 bodies through a computed jump: every body must be
# exactly WV_MB_BYTES long. Compiles a one-kernel probe for gfx950 (no GPU needed), disassembles it and checks the stride.
set -e
D=$(mktemp -d); R=$(cd "$(dirname "$0")/.." && pwd)
cat > $D/t.hip <<EOT
#include "$R/swarmkit_amd/csrc/swp_wave.hpp"
__global__ void k(unsigned* io) {
    unsigned l = threadIdx.x, bits = io[l], w = io[64 + l], bits2 = io[400 + l], w2 = io[464 + l], pickb = 0;
    unsigned at = wv::match_seq64(bits, w, bits2, w2, pickb, l, io[128]);
    io[192 + l] = bits; io[256 + l] = pickb; io[528 + l] = w; io[592 + l] = bits2; if (l == 0) io[320] = at;
}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -I$R/swarmkit_amd/csrc $D/t.hip -o $D/t.s 2>/dev/null
/opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $D/t.s -o $D/t.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d $D/t.elf | python3 -c "
import re, sys
want = int(re.search(r'#define WV_MB_BYTES (\d+)', open('$R/swarmkit_amd/csrc/swp_wave.hpp').read()).group(1))
addr = [int(m.group(1), 16) for l in sys.stdin if 'v_readlane_b32' in l and (m := re.search(r'// ([0-9A-F]+):', l))]
first = addr[0:128:2]   # two readlanes per body (the 64 step stubs behind them have one each)
strides = {b - a for a, b in zip(first, first[1:])}
assert len(first) == 64 and strides == {want}, (len(first), strides, want)
print('match_seq64: 64 bodies of', want, 'bytes')
"
rm -rf $D
