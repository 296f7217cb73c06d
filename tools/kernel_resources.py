#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch use from the -save-temps ISA of the HIP build (amdhsa.kernels metadata).
Usage: python tools/kernel_resources.py [/tmp/vrtbuild/vrt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s] [name filter]"""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/vrtbuild/vrt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
flt = sys.argv[2] if len(sys.argv) > 2 else "march"
text = open(path).read()
meta = text[text.index("amdhsa.kernels:"):]
for block in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    f = {k: v for k, v in re.findall(r"\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\S+)", block)}
    name = f.get("name", "?")
    if flt not in name:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    v = int(f.get("vgpr_count", 0))
    print(f"{name[:70]:70s} vgpr {v:4d} (waves/SIMD {min(8, 512 // max(v, 1))})  sgpr {f.get('sgpr_count'):>4s}  lds {f.get('group_segment_fixed_size'):>6s}  "
          f"scratch {f.get('private_segment_fixed_size'):>4s}  spills s{f.get('sgpr_spill_count', '0')} v{f.get('vgpr_spill_count', '0')}")
