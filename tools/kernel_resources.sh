#!/bin/bash
# Per-kernel register / scratch / occupancy table of one HIP source (compiler remarks).
# usage: tools/kernel_resources.sh hip/fft_native.hip [filter-regex]
cd "$(dirname "$0")/../21cmfast_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result \
  -Wno-array-bounds -I../../include -Ihip -Ihost -Rpass-analysis=kernel-resource-usage $EXTRA \
  -c "$1" -o /tmp/kres_$$.o 2> /tmp/kres_$$.log
python3 - "$2" /tmp/kres_$$.log <<'PY'
import re, sys, subprocess
pat = re.compile(sys.argv[1] or ".")
rows, cur = [], None
for line in open(sys.argv[2]):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception: pass
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':58s} VGPR AGPR spillV scratch occ  LDS")
for r in rows:
    if not pat.search(r["name"]): continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>5s}")
PY
rm -f /tmp/kres_$$.o /tmp/kres_$$.log
