"""Summarise the inner loops of the kernels in a .s file: loads, waits, LDS ops, barriers and MFMA runs in order."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z[\w]+):', s, re.M):
    name = m.group(1)
    if pat and pat not in name: continue
    start = m.start(); end = s.find('.Lfunc_end', start)
    body = s[start:end].splitlines()
    out = []; mf = 0; va = 0
    for l in body:
        t = l.strip()
        if t.startswith('v_mfma'): mf += 1; continue
        key = ('s_waitcnt', 's_cbranch', 's_branch', 'global_load', 'flat_load', '.LBB', 'buffer_load', 's_barrier', 'ds_read', 'ds_write', 'global_store', 'ds_load', 'ds_store')
        if any(t.startswith(x) for x in key):
            if mf: out.append('   ... %d mfma' % mf); mf = 0
            out.append(t[:80])
    if mf: out.append('   ... %d mfma' % mf)
    # compress consecutive identical op classes
    comp = []; 
    for o in out:
        cls = o.split()[0] if not o.startswith('   ...') else o
        if comp and comp[-1][0] == cls and not cls.startswith('.LBB') and not cls.startswith('s_waitcnt'):
            comp[-1][1] += 1
        else:
            comp.append([cls, 1, o])
    vg = re.search(r'; NumVgprs: (\d+)', s[end:end+3000]); oc = re.search(r'; Occupancy: (\d+)', s[end:end+3000])
    print("=====", name[:90], "VGPRs", vg and vg.group(1), "occ", oc and oc.group(1))
    txt = []
    for cls, n, o in comp:
        txt.append((o if n == 1 else "%s x%d" % (cls, n)))
    full = "\n".join(txt)
    # only print around loop headers
    for lm in re.finditer(r'.*Inner Loop Header.*', full):
        i = lm.start()
        print(full[i:i+900]); print("   ------")
