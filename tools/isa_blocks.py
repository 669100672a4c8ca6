"""Instruction-class histogram per basic block of the kernels in a hipcc -S listing (perf triage helper).
usage: python tools/isa_blocks.py file.s kernel-name-substring [min_instructions]"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op.startswith('s_'): return 'salu'
    return op


def main():
    txt = open(sys.argv[1]).read()
    pat = sys.argv[2]
    minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):.*$', txt, re.M):
        name = m.group(1)
        body = txt[m.end():txt.index('s_endpgm', m.end())]
        parts = re.split(r'\n(\.LBB\d+_\d+):', body)
        print(name)
        for label, bb in zip(['entry'] + parts[1::2], [parts[0]] + parts[2::2]):
            ins = [l.split()[0] for l in bb.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
            if len(ins) >= minins:
                print('  ', label, len(ins), dict(collections.Counter(classify(o) for o in ins)))


main()
