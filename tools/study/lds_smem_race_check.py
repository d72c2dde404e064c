import re,sys
def regs(tok):
    # v12 or v[12:15]
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
def sregs(tok):
    m=re.match(r's\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r's(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
def analyze(path,kname):
    s=open(path).read()
    i=s.find(kname+':'); j=s.find('s_endpgm',i)
    lines=[l.strip() for l in s[i:j].splitlines()]
    pend_v=set(); pend_s=set(); findings=[]
    for n,l in enumerate(lines):
        if not l or l.startswith(';') or l.startswith('.'): continue
        mm=re.match(r'(\w+)\s*(.*)',l)
        if not mm: continue
        op,args=mm.group(1),mm.group(2)
        toks=[t.strip() for t in re.split(r',\s*(?![^\[]*\])',args.split(' op_sel')[0].split(' neg_')[0].split(' offset')[0])] if args else []
        if op.startswith('s_waitcnt'):
            if 'lgkmcnt(0)' in args: pend_v.clear(); pend_s.clear()
            continue
        if op.startswith('ds_read'):
            # reads address reg (toks[1]); writes toks[0]
            if toks and regs(toks[1]) & pend_v: findings.append((n,l,'addr pending'))
            pend_v |= regs(toks[0]); continue
        if op.startswith('s_load'):
            if len(toks)>1 and sregs(toks[1]) & pend_s: findings.append((n,l,'base pending'))
            pend_s |= sregs(toks[0]); continue
        # any other instruction: check all operands (incl. dst: WAW) against pending
        used_v=set(); used_s=set()
        for t in toks:
            used_v |= regs(t); used_s |= sregs(t)
        if used_v & pend_v: findings.append((n,l,'VGPR of an LDS read in flight: '+str(sorted(used_v&pend_v))))
        if used_s & pend_s: findings.append((n,l,'SGPR of a scalar load in flight: '+str(sorted(used_s&pend_s))))
    return findings
for path,k in [(sys.argv[1],sys.argv[2])]:
    f=analyze(path,k)
    print(k[:50],len(f),'findings')
    for x in f[:12]: print('  ',x)
