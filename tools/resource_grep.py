"""usage: python tools/resource_grep.py RES.txt PATTERN...  -- VGPRs / spills / scratch of the kernels whose mangled name contains a pattern
(RES.txt = stderr of hipcc -Rpass-analysis=kernel-resource-usage)"""
import re, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip().split(' ')[0]
    if any(p in name for p in sys.argv[2:]):
        g = lambda k: (re.search(k + r': (\d+)', b) or [None, None])[1]
        print("%-110s VGPR %s spill %s sgpr-spill %s scratch %s" % (name[:110], g('VGPRs'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'ScratchSize \[bytes/lane\]')))
