"""Chain lengths of the v_mfma instructions of a kernel in a .s file: runs of consecutive MFMAs with the same destination
(= accumulator forwarding, the fast path of v_mfma_f64_16x16x4_f64: scratch/mfma_peak3.hip).
    python scratch/mfma_chains.py /tmp/dm_gemm.s zgemm_grouped_kernelILb0ELb0"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z[\w]+):', s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    start = m.start(); end = s.find('.Lfunc_end', start)
    seq = [l.strip().split()[1].rstrip(',') for l in s[start:end].splitlines() if l.strip().startswith('v_mfma')]
    runs, cur, n = [], None, 0
    for d in seq:
        if d == cur:
            n += 1
        else:
            if cur:
                runs.append(n)
            cur, n = d, 1
    if cur:
        runs.append(n)
    vg = re.search(r'; NumVgprs: (\d+)', s[end:end + 3000]); ag = re.search(r'; NumAgprs: (\d+)', s[end:end + 3000])
    oc = re.search(r'; Occupancy: (\d+)', s[end:end + 3000]); sp = re.search(r'; ScratchSize: (\d+)', s[end:end + 3000])
    print(name[:100], "mfma", len(seq), "VGPRs", vg and vg.group(1), "AGPRs", ag and ag.group(1), "occupancy", oc and oc.group(1),
          "scratch", sp and sp.group(1))
    print("   chain lengths:", runs[:96])
