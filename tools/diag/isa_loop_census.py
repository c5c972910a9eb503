"""Instruction census of the loops of one kernel in a hipcc -save-temps .s file:
    python tools/diag/isa_loop_census.py FILE.s MANGLED_NAME_FRAGMENT
Prints, for every backward branch (label .. branch), the counts of MFMA / other vector ALU / LDS / vector memory / scalar /
s_waitcnt instructions between them. The diet of the issue-bound kernels (lookup sampling loop, conv epilogues) is read from
this beside the SQ counters."""
import re
import sys
from collections import Counter


def census(lines):
    c = Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            c['vmem'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        else:
            c[op] += 1
    return dict(c)


def main():
    txt = open(sys.argv[1]).read()
    frag = sys.argv[2]
    m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(frag), txt, re.M)
    if not m:
        sys.exit("no kernel matching %s" % frag)
    start = m.end()
    end = txt.index('s_endpgm', start)
    body = txt[start:end].split('\n')
    print(m.group(1), "whole kernel:", census(body))
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a = labels[mm.group(1)]
            print("loop lines %d..%d:" % (a, i), census(body[a:i + 1]))


if __name__ == "__main__":
    main()
