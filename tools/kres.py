#!/usr/bin/env python
"""Per-kernel resource usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, AGPRs, spills, scratch, LDS, occupancy.
    python tools/kres.py kernels_conv_h2.hip [name filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = [d for d in os.listdir(root) if d.startswith("one-stop") and d.endswith("_amd")][0] + "/csrc"
f = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/tmp/_kres.o"] + sys.argv[3:]
out = subprocess.run(cmd, cwd=os.path.join(root, csrc), capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        nm = t.split(":", 1)[1].strip()
        try: nm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", nm], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", nm).split("(")[0]
        cur = {"name": nm}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    if filt in r["name"]:
        print(f"{r['name'][:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPR Spill', r.get('VGPRs Spill','?')):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} LDS {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
