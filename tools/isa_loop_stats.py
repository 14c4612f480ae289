"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing.
Usage: python tools/isa_loop_stats.py file.s <substring of the mangled kernel name> [min mfma per block]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 4
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and ":" in l[:l.find(";")] if True)
end = start
while "s_endpgm" not in lines[end]:
    end += 1
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
    else:
        cur.append(l.strip())
blocks.append((name, cur))
for name, b in blocks:
    n = sum(1 for l in b if l.startswith("v_mfma"))
    if n >= min_mfma:
        ops = {}
        for l in b:
            if not l or l.startswith(";") or l.startswith("."):
                continue
            op = l.split()[0]
            ops[op] = ops.get(op, 0) + 1
        print(key[:60], name, "instructions", sum(ops.values()), "mfma", n)
        print("   ", sorted(ops.items(), key=lambda kv: -kv[1])[:30])
