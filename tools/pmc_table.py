"""Table from a tools/pmc_raw.sh text dump: python tools/pmc_table.py gpurun_out/x_pmc.txt  (issue_us = VALU instructions x 4 cycles over 1024 SIMDs at 2.1 GHz)"""
import re, sys
txt = open(sys.argv[1]).read()
ker = None; d = {}
for l in txt.split('\n'):
    if l and not l.startswith(' '):
        ker = l.strip(); d[ker] = {}
    else:
        m = re.match(r'\s+(\S+)\s+median\s+([\d.]+)', l)
        if m and ker: d[ker][m.group(1)] = float(m.group(2))
print('%-46s %8s %8s %8s %6s %6s %6s %6s %8s %8s %9s %9s' % ('kernel', 'us', 'issue_us', 'waves', 'occ', 'act', 'wait', 'winst', 'rdMB', 'wrMB', 'L1acc(M)', 'L2req(M)'))
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get('_us', 0)):
    if '_us' not in c or 'SQ_INSTS_VALU' not in c: continue
    us = c['_us']; issue = c['SQ_INSTS_VALU'] * 4 / 1024 / 2100.0
    wc = max(c.get('SQ_WAVE_CYCLES', 0), 1)
    print('%-46s %8.0f %8.0f %8.0f %6.2f %6.2f %6.2f %6.2f %8.0f %8.0f %9.1f %9.1f' % (k[:46], us, issue, c.get('SQ_WAVES', 0), wc * 4 / 2100.0 / 1024 / us,
          c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('FETCH_SIZE', 0) * 2 * 1024 / 1e6, c.get('WRITE_SIZE', 0) * 1024 / 1e6,
          c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / 1e6, c.get('TCP_TCC_READ_REQ_sum', 0) / 1e6))
