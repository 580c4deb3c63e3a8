#!/usr/bin/env python3
"""Condensed memory / wait event sequence of each kernel in an AMDGPU .s listing (hipcc -S --cuda-device-only): a quick way to spot an
s_waitcnt vmcnt that landed between the loads of one gather (the register allocator reusing a register of a load in flight) or a vmcnt(0)
where a counted wait was intended.  usage: isa_events.py file.s [substring of the kernel symbol]"""
import re, sys
path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
name, ev, n = None, [], 0
def flush():
    if name and want in name and ev:
        print(f"== {name}")
        out, last, cnt = [], None, 0
        for e in ev + [None]:
            if e == last: cnt += 1; continue
            if last is not None: out.append(last if cnt == 1 else f"{last}x{cnt}")
            last, cnt = e, 1
        print(" ".join(out))
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: flush(); name, ev = m.group(1), []; continue
    t = line.strip()
    if t.startswith("s_endpgm"): flush(); name = None; continue
    if name is None: continue
    op = t.split(" ")[0].split("\t")[0]
    if op.startswith(("buffer_load", "global_load")): ev.append("L" + ("lds" if "lds" in t else ""))
    elif op.startswith(("buffer_store", "global_store")): ev.append("S")
    elif op == "s_waitcnt" and "vmcnt" in t: ev.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
    elif op == "s_barrier": ev.append("|B|")
    elif op.startswith("v_mfma"): ev.append("m")
    elif op.startswith("s_cbranch"): ev.append("/")
    elif re.match(r"^\.LBB", t): ev.append(":")
