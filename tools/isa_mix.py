#!/usr/bin/env python3
"""isa_mix.py <file.s> <kernel-name-substring> [--between OPCODE]: static instruction mix of one kernel of a hipcc -S listing.
--between OPCODE restricts the count to the span from the first to the last occurrence of OPCODE (e.g. v_fmac_f32_dpp: the scatter
loop nest of p2g, which is straight-line code after unrolling)."""
import collections, sys

def main():
    path, sub = sys.argv[1], sys.argv[2]
    between = sys.argv[sys.argv.index("--between") + 1] if "--between" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if sub in l.split(":")[0] and l.split(";")[0].rstrip().endswith(":") and not l.startswith((" ", "\t", ".")))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    ins = []
    for l in lines[start + 1:end]:
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        ins.append(t.split()[0])
    if between:
        idx = [i for i, o in enumerate(ins) if o == between]
        ins = ins[idx[0]:idx[-1] + 1]
    c = collections.Counter(ins)
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    print(f"instructions {sum(c.values())}: VALU {grp('v_')} (DPP {sum(n for k, n in c.items() if 'dpp' in k)}, packed {grp('v_pk_')}), SALU {grp('s_')}, "
          f"LDS {grp('ds_')}, global {grp('global_') + grp('buffer_') + grp('flat_')}")
    for k, n in c.most_common(40): print(f"  {k:30s}{n}")

if __name__ == "__main__":
    main()
