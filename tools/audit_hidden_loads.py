"""CFG-aware audit of the -save-temps ISA of conv_split.hip / conv_halo.hip: between a hidden (inline-asm) global load
and the s_waitcnt vmcnt(N) that RETIRES it on every execution path, no instruction may read or write the load's
destination registers (hipcc does not know they are in flight: a register-allocator copy or reuse there would silently
corrupt the tile).  Vector-memory loads retire in issue order, so a wait vmcnt(N) retires exactly the loads that have
at least N younger vector-memory instructions behind them; the audit tracks, per pending register, the minimum number
of younger instructions over all paths, so a counted wait that leaves the hidden loads in flight does not clear them.

Usage: hipcc ... -c conv_split.hip -save-temps=obj ; python tools/audit_hidden_loads.py <file.s>"""
import re
import sys


def regs_of(line):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        regs.add(int(m.group(1)))
    return regs


def audit_function(name, lines):
    # split into basic blocks
    blocks, cur, label = {}, [], "entry"
    order = []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks[label] = cur
            order.append(label)
            label, cur = m.group(1), []
        else:
            cur.append(ln.strip())
    blocks[label] = cur
    order.append(label)
    succ = {}
    for i, lb in enumerate(order):
        ins = [l for l in blocks[lb] if l and l[0] not in ".;"]
        s = set()
        fall = True
        for l in ins:
            if l.startswith("s_cbranch"):
                s.add(l.split()[-1])
            elif l.startswith("s_branch"):
                s.add(l.split()[-1])
                fall = False
            elif l.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(order):
            s.add(order[i + 1])
        succ[lb] = s
    VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
            "flat_store", "scratch_")
    CAP = 64
    entry = {lb: {} for lb in order}   # register -> minimum number of younger vector-memory instructions
    bad = []
    changed = True
    rounds = 0
    while changed and rounds < 200:
        changed = False
        rounds += 1
        bad = []
        for lb in order:
            pending = dict(entry[lb])
            body = blocks[lb]
            for i, l in enumerate(body):
                if not l or l[0] in ".;":
                    continue
                if l.startswith("global_load_dwordx4") and i > 0 and body[i - 1].startswith(";;#ASMSTART"):
                    m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)
                    addr = regs_of(l.split(",", 1)[1])
                    if addr & set(pending):
                        bad.append((lb, l))
                    for r in pending:
                        pending[r] = min(CAP, pending[r] + 1)
                    for r in range(int(m.group(1)), int(m.group(2)) + 1):
                        pending[r] = 0
                    continue
                if l.startswith("s_waitcnt") and "vmcnt" in l:
                    n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                    pending = {r: k for r, k in pending.items() if k < n}
                    continue
                if regs_of(l) & set(pending):
                    bad.append((lb, l))
                if l.startswith(VMEM):
                    for r in pending:
                        pending[r] = min(CAP, pending[r] + 1)
            for t in succ[lb]:
                if t not in entry:
                    continue
                tgt = entry[t]
                for r, k in pending.items():
                    if r not in tgt or k < tgt[r]:
                        tgt[r] = k
                        changed = True
    return bad


def main(path):
    s = open(path).read()
    total = 0
    for m in re.finditer(r"^(_ZN2pp\w*conv_(?:halo_)?split(?:_ct|_tall)?_kernel\w+):", s, flags=re.M):
        a = m.end()
        b = s.index(".Lfunc_end", a)
        bad = audit_function(m.group(1), s[a:b].split("\n"))
        total += len(bad)
        print(m.group(1)[-40:], "violations:", len(bad))
        for lb, l in bad[:6]:
            print("   ", lb, l)
    print("total violations", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/conv_split-hip-amdgcn-amd-amdhsa-gfx950.s"))
