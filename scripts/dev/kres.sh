#!/bin/bash
# kres.sh <file.hip> [extra flags] — per-kernel SGPR / VGPR / AGPR / scratch / occupancy / LDS of one translation unit (hipcc remarks);
# leaves the gfx950 assembly in /tmp/kres/<name>-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
src=$(readlink -f "$1"); shift
mkdir -p /tmp/kres && cd /tmp/kres
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-logical-op-parentheses \
  -Rpass-analysis=kernel-resource-usage -save-temps "$@" -c "$src" -o /tmp/kres/out.o 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    if "error" in line or "warning:" in line: print(line.rstrip())
    m=re.search(r"remark: .*?: +(.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur={"name":t.split(": ",1)[1]}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
for r in rows:
    try: nm=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    except Exception: nm=r["name"]
    nm=re.sub(r"\(anonymous namespace\)::","",nm); nm=re.sub(r"\(.*","",nm)
    print("%-70s sgpr %-4s vgpr %-4s agpr %-4s scratch %-5s occ %-2s lds %s"%(nm[:70],r.get("TotalSGPRs"),r.get("VGPRs"),r.get("AGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
