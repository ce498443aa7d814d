#!/usr/bin/env python3
"""Static checks over the device code of the PRODUCT's objects (swarmkit_amd/lib/obj/*.o as `make` left them; no GPU, no recompile):

  1. every instance of the hand-scheduled matcher (wv::match_seq64, csrc/swp_wave.hpp) in every kernel: the computed jump enters body
     `from` at 80 bytes each — the 64 bodies must lie at exactly that stride IN THE KERNELS THAT SHIP (tools/check_matcher_asm.sh
     checks a probe kernel only: other registers, possibly other encodings);
  2. the kernels' resource numbers from the code object's metadata: a 1024-thread kernel (4 waves per SIMD: 128 registers a lane) must
     stay at or below 120 VGPRs unless it is on the list below with its reason, must not spill VGPRs, and only the kernels listed may
     have a scratch segment. NOTES_r05 1 / DESIGN 9: the fused commit kernel once died under its debug timers at 126-127 VGPRs; round 6
     rebuilt that configuration twice (HEAD and commit 6f5858a with the fold as ONE batch of 64 registers: 112 and 127 VGPRs) and ran
     it under SWP_DBG=16 without a fault — the register count was not the cause — but the edge stays fenced off: a kernel that grows
     into it fails the build's check instead of being found by a debug run.

usage: python tools/check_kernels.py [--verbose]      exit code 1 on a violation"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "swarmkit_amd", "lib", "obj")
# kernels allowed a scratch segment (bytes per lane), with the reason
SCRATCH_OK = {
    "k_r6_commit_v": "vol_choose's per-mount arrays (the only block-resolver instance with cluster mounts)",
    "k_r7_commit_v": "the same, sharded",
    "k_groups2": "three inlined instances of one group's machine: SGPR spills to VGPR lanes, 192 B of timers and frames",
    "k_explain": "per-task Explain of tasks with generic reservations / mounts: small per-thread arrays",
    "k_vol_choose": "chooseTaskVolumes for one pair: per-mount arrays",
    "k_r6_volrows": "per-mount arrays",
    "k_resolve5": "the round resolver (SWP_RESOLVER=5 only; behind the block resolver at every size since round 4): its listers' per-task arrays",
}
# 1024-thread kernels allowed above 120 VGPRs
VGPR_EDGE_OK = {
    "k_groups2": "one workgroup per launch: the machine wave's three instances; no VGPR spills beyond the listed 4 is checked below",
    "k_resolve5": "the round resolver (SWP_RESOLVER=5 only)",
}
VGPR_SPILL_OK = {"k_groups2": 12,    # (orderedNodes' second hand-out: swap-based moves of 16-byte records — rare paths; checked in the disassembly)
                 "k_resolve5": 40}


def demangle_short(name):
    m = re.match(r"_ZN6swpdev(\d+)", name)
    if m:
        n = int(m.group(1))
        start = m.end()
        return name[start:start + n]
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    return name


def device_elf(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + ".fat")
    elf = os.path.join(tmp, os.path.basename(obj) + ".elf")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return None
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}", "--unbundle"], check=True)
    return elf


def kernels_meta(elf):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
    ks, cur = [], {}
    for line in out.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            ks.append(cur)
            cur = {}
        if k == "agpr_count":
            cur = {"agpr_count": v}
        elif k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "group_segment_fixed_size"):
            cur[k] = v
    if cur.get("name"):
        ks.append(cur)
    return ks


def matcher_instances(elf):
    """-> [(kernel, [body addresses])]: the address of the first v_readlane of each of the 64 bodies behind every s_setpc_b64 s[94:95]"""
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", elf], capture_output=True, text=True, check=True).stdout
    inst, kernel, pending, bodies = [], None, False, []
    lines = dis.splitlines()
    i = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            kernel = m.group(1)
        if "s_setpc_b64 s[94:95]" in l:
            # the bodies follow: body = v_readlane, v_readlane, s_sub, s_and, s_cbranch, ... (13 instructions, two readlanes at its head)
            addrs = []
            j = i + 1
            while j < len(lines) and len(addrs) < 64:
                lj = lines[j]
                if "v_readlane_b32" in lj and j + 1 < len(lines) and "v_readlane_b32" in lines[j + 1]:
                    a = re.search(r"//\s*([0-9A-Fa-f]+):", lj)
                    addrs.append(int(a.group(1), 16))
                    j += 2
                    continue
                if re.match(r"^[0-9a-f]+ <", lj):
                    break
                j += 1
            inst.append((kernel, addrs))
            i = j
            continue
        i += 1
    return inst


def main():
    verbose = "--verbose" in sys.argv
    want = int(re.search(r"#define WV_MB_BYTES (\d+)", open(os.path.join(ROOT, "swarmkit_amd", "csrc", "swp_wave.hpp")).read()).group(1))
    bad = []
    n_inst = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(OBJ)):
            if not f.endswith(".o"):
                continue
            elf = device_elf(os.path.join(OBJ, f), tmp)
            if elf is None:
                continue   # (a host-only object: swp_sched.o)
            for k in kernels_meta(elf):
                name = demangle_short(k["name"])
                vg, sc, wg = int(k.get("vgpr_count", 0)), int(k.get("private_segment_fixed_size", 0)), int(k.get("max_flat_workgroup_size", 0))
                vs = int(k.get("vgpr_spill_count", 0))
                if verbose:
                    print(f"{f:18s} {name:28s} wg {wg:5d} vgpr {vg:4d} sgpr {k.get('sgpr_count'):>4s} scratch {sc:4d} vgpr-spill {vs} sgpr-spill {k.get('sgpr_spill_count')}")
                if sc and name not in SCRATCH_OK:
                    bad.append(f"{name}: a scratch segment of {sc} bytes per lane (not on the list of kernels that may have one)")
                if vs > VGPR_SPILL_OK.get(name, 0):
                    bad.append(f"{name}: {vs} VGPRs spilled")
                if wg >= 1024 and vg > 120 and name not in VGPR_EDGE_OK:
                    bad.append(f"{name}: {vg} VGPRs in a 1024-thread kernel (128 is the end of the file: keep it at or below 120)")
            for kernel, addrs in matcher_instances(elf):
                n_inst += 1
                strides = {b - a for a, b in zip(addrs, addrs[1:])}
                if len(addrs) != 64 or strides != {want}:
                    bad.append(f"{demangle_short(kernel)}: an instance of match_seq64 with {len(addrs)} bodies at strides {sorted(strides)} (the computed jump assumes {want})")
    print(f"check_kernels: {n_inst} instances of match_seq64 with 64 bodies of {want} bytes each" if not bad else "check_kernels: FAILED")
    for b in bad:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
