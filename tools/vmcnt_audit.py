"""Audit the counted s_waitcnt vmcnt(N) of a kernel's hot loop in hipcc's ISA (-S output): for every wait, which vector-memory
operation is the YOUNGEST one that has to be complete, and how many instructions ago it was issued.  Vector memory operations
retire in order, so a wait that reaches an operation issued a few dozen instructions earlier (or a store of the same loop
pass) is a stall for a full memory round trip -- typically a false register dependency hipcc's waitcnt pass sees (e.g. a
packed instruction whose unused half names a register an in-flight load will write).
usage: python3 tools/vmcnt_audit.py <file.s> <kernel name substring> [min_age]"""
import re
import sys


def main(path, name, min_age=150):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    size, a, b = max(loops)
    print(f"{name}: hot loop = lines {a}..{b} of the kernel ({size} lines)")
    loop = [l for l in body[a:b] if l.strip() and not l.strip().startswith(";") and not l.startswith(".")]
    seq = loop + loop      # two passes: the second one is steady state
    ops = []               # (position, text) of vector memory operations in issue order
    for pos, l in enumerate(seq):
        t = l.strip()
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", t):
            ops.append((pos, t))
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m and pos >= len(loop):
            n = int(m.group(1))
            if len(ops) > n:
                ypos, ytxt = ops[len(ops) - n - 1]
                age = pos - ypos
                flag = "  <-- STALL RISK" if age < min_age else ""
                print(f"  line {pos - len(loop):4d}  vmcnt({n:2d}) needs `{ytxt[:60]}` issued {age} instructions earlier{flag}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 150)
