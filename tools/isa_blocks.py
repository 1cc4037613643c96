"""Development: per basic block of one kernel in a hipcc .s file, the instruction mix (MFMA / VALU / SALU / LDS / VMEM / waits).
    python tools/isa_blocks.py file.s 'conv_l16_fwd_kernelILi3ELi3ELi8ELi2ELb0ELb0E' [--min-mfma 4]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[4]) if len(sys.argv) > 4 else 4
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith("_Z"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    if re.match(r"^\.?L?BB\d+_\d+:", s):
        blocks.append((name, cur)); cur, name = [], s.split(":")[0]
        continue
    cur.append(s.split(";")[0].strip())
blocks.append((name, cur))


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


tot = {}
for name, ins in blocks:
    c = {}
    for i in ins:
        k = kind(i.split()[0])
        c[k] = c.get(k, 0) + 1
    if c.get("mfma", 0) >= min_mfma:
        print("%-12s %4d instr: " % (name, len(ins)) + "  ".join("%s %d" % kv for kv in sorted(c.items())))
        vops = {}
        for i in ins:
            if kind(i.split()[0]) in ("valu", "salu"):
                vops[i.split()[0]] = vops.get(i.split()[0], 0) + 1
        print("             " + " ".join("%s:%d" % kv for kv in sorted(vops.items(), key=lambda kv: -kv[1])[:14]))
print("blocks", len(blocks), "total instr", sum(len(b[1]) for b in blocks))
