"""Register / occupancy table of every kernel in two AMDGPU .s listings of the same file (e.g. built with and without -mllvm -amdgpu-mfma-vgpr-form=1):
    python scripts/isa_regs.py a.s b.s"""
import re,sys
def table(path):
    t={}
    txt=open(path).read()
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        t[m.group(2)]=(int(m.group(3)),int(m.group(1)),int(m.group(4)))
    return t
a=table(sys.argv[1]); b=table(sys.argv[2])
def occ(v): return min(8, 512//max(v,1)) if v else 8
for k in a:
    va,aa,sa=a[k]; vb,ab,sb=b.get(k,(0,0,0))
    flag = "  <-- better" if occ(vb)>occ(va) else ("  <-- WORSE" if occ(vb)<occ(va) or sb>sa else "")
    print(f"{k[:90]:90s} total {va:3d} (acc {aa:3d}, spill {sa}) occ {occ(va)} | vgpr-form total {vb:3d} (acc {ab:3d}, spill {sb}) occ {occ(vb)}{flag}")
