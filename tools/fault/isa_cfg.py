"""Developer aid (round 4, the seed-24 fault): a control-flow graph over an llvm-objdump disassembly of ONE kernel, and a search for
definitions of a vector register on the paths between two instructions.
usage: python3 tools/fault/isa_cfg.py <kernel.s> <from_line> <to_line> <reg e.g. v31>"""
import re, sys
from collections import defaultdict, deque

def parse(path):
    ins = []   # (lineno, addr, text, target_addr or None)
    base = None
    for i, l in enumerate(open(path), 1):
        m = re.search(r'//\s*([0-9A-F]{8,16}):', l)
        if not m:
            continue
        addr = int(m.group(1), 16)
        text = l.split('//')[0].strip()
        t = None
        mt = re.search(r'<[^>]*\+0x([0-9a-f]+)>', l)
        if mt and re.match(r's_(c?branch)', text):
            t = int(mt.group(1), 16)
        ins.append([i, addr, text, t])
    return ins

def main():
    path, a, b, reg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    ins = parse(path)
    kbase = ins[0][1]
    by_addr = {x[1]: k for k, x in enumerate(ins)}
    by_line = {x[0]: k for k, x in enumerate(ins)}
    succ = defaultdict(list)
    for k, (ln, addr, text, t) in enumerate(ins):
        op = text.split()[0]
        if op == 's_endpgm':
            continue
        if op == 's_setpc_b64':
            continue   # long branches: target computed two instructions earlier (handled below)
        if t is not None:
            tk = by_addr.get(kbase + t)
            if tk is not None:
                succ[k].append(tk)
            if op == 's_branch':
                continue
        if k + 1 < len(ins):
            succ[k].append(k + 1)
    # long branches: s_getpc_b64 s[x:y]; s_add_u32 sx, sx, imm; s_addc_u32; s_setpc_b64  -> target = addr(getpc)+4 + imm
    for k, (ln, addr, text, t) in enumerate(ins):
        if text.startswith('s_getpc_b64') and k + 3 < len(ins) and ins[k + 3][2].startswith('s_setpc_b64'):
            m = re.search(r'0x([0-9a-f]+)$', ins[k + 1][2])
            if m:
                imm = int(m.group(1), 16)
                if imm >= 1 << 31:
                    imm -= 1 << 32
                tk = by_addr.get(addr + 4 + imm)
                if tk is not None:
                    succ[k + 3] = [tk]
    num = int(reg[1:])
    def defines(text):
        op = text.split()[0]
        if op.startswith(('scratch_store', 'global_store', 'flat_store', 'ds_write', 's_', 'v_cmp', 'v_readlane', 'v_readfirstlane', 'buffer_', 'global_atomic')) and not op.startswith('v_cmpx'):
            return False
        rest = text[len(op):].strip()
        first = rest.split(',')[0].strip()
        m = re.match(r'v\[(\d+):(\d+)\]$', first)
        if m:
            return int(m.group(1)) <= num <= int(m.group(2))
        m = re.match(r'v(\d+)$', first)
        return bool(m) and int(m.group(1)) == num
    src, dst = by_line[a], by_line[b]
    # forward reachability from src that can still reach dst
    pred = defaultdict(list)
    for k, ss in succ.items():
        for s in ss:
            pred[s].append(k)
    can_reach = set([dst]); dq = deque([dst])
    while dq:
        k = dq.popleft()
        for p in pred[k]:
            if p not in can_reach:
                can_reach.add(p); dq.append(p)
    seen = set([src]); dq = deque([src]); hits = []
    while dq:
        k = dq.popleft()
        if k != src and k in can_reach and defines(ins[k][2]):
            hits.append(k)
        if k == dst:
            continue
        for s in succ[k]:
            if s not in seen and s in can_reach:
                seen.add(s); dq.append(s)
    print("instructions on some path %d -> %d: %d" % (a, b, len(seen & can_reach)))
    for k in sorted(hits):
        print("  def of %s at line %d: %s" % (reg, ins[k][0], ins[k][2]))

main()
