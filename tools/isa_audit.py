"""Instruction audit of a kernel from the compiler's gfx950 assembly ("first commit an instruction audit").
  hipcc <the Makefile's flags> -S --offload-device-only k_godunov.hip -o /tmp/k_godunov.s
  python tools/isa_audit.py /tmp/k_godunov.s '_ZN5iamrx7k_god_zILi14ELi14ELi256ELi2ELb0ELb0'
Prints, for the whole kernel and for its hottest loop (the largest span closed by a backward branch = the per-plane march), the number of
instructions by class: fp64 arithmetic (fma / add / mul / min-max / other), compares, selects (v_cndmask), moves, integer / bit / address
VALU, LDS reads and writes, global loads and stores, scalar ALU, waits and barriers.  VALU classes are what SQ_INSTS_VALU counts."""
import re, sys, collections

CLASSES = [
    ("fp64 fma", r"^v_fma_f64|^v_fmac_f64"), ("fp64 add", r"^v_add_f64"), ("fp64 mul", r"^v_mul_f64"),
    ("fp64 min/max", r"^v_(min|max)_f64|^v_(min|max)imum"), ("fp64 other (rcp, div fixup, cvt, ldexp, trig...)", r"^v_\w+_f64"),
    ("compare", r"^v_cmp"), ("select (v_cndmask)", r"^v_cndmask"), ("move (v_mov / accvgpr / dpp)", r"^v_mov|^v_accvgpr|^v_readlane|^v_writelane|^v_readfirstlane|^v_swap"),
    ("int / bit / address VALU", r"^v_"),
    ("LDS read", r"^ds_read|^ds_load"), ("LDS write", r"^ds_write|^ds_store"), ("global load", r"^global_load|^buffer_load|^flat_load"),
    ("global store", r"^global_store|^buffer_store|^flat_store"), ("scratch", r"^scratch_"), ("wait", r"^s_waitcnt|^s_nop"), ("barrier", r"^s_barrier"),
    ("branch", r"^s_cbranch|^s_branch"), ("scalar ALU / other", r"^s_"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(key) and l.rstrip().endswith(":") is False and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.?\w+):", s)
        if m and not s.split()[0].startswith(("v_", "s_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
            labels[m.group(1)] = len(insts)
            continue
        insts.append(s.split(";")[0].strip())
    # backward branches -> loops
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    def count(seg):
        c = collections.Counter(classify(x.split()[0]) for x in seg)
        return c
    def report(title, seg):
        c = count(seg)
        valu = sum(v for k, v in c.items() if k.startswith(("fp64", "compare", "select", "move", "int")))
        print(f"{title}: {len(seg)} instructions, {valu} VALU")
        for name, _ in CLASSES + [("other", "")]:
            if c.get(name):
                print(f"    {c[name]:6d}  {name}")
    report("whole kernel", insts)
    if loops:
        a, b = max(loops, key=lambda ab: ab[1] - ab[0])
        report(f"hottest loop (instructions {a}..{b})", insts[a:b + 1])
        inner = [(x, y) for x, y in loops if x >= a and y <= b and (x, y) != (a, b)]
        for x, y in inner:
            print(f"    (inner loop {x}..{y}: {y - x + 1} instructions, counted once above)")
    m = re.search(r"; NumVgprs: (\d+)", "\n".join(lines[end:end + 80]))
    for l in lines[end:end + 80]:
        if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs", l):
            print("   ", l.strip("; ").strip())


main()
