import re,sys
cur=None; case=None; last=None
for l in open(sys.argv[1]):
    l=l.rstrip()
    if l.startswith('==='): cur=l[4:]; print(cur); continue
    if l.startswith('----'): case=l; continue
    m=re.match(r'(q64 ksplit=\d+.*?)\s+([\d.]+) us\s+([\d.]+) TF', l)
    if m: last=[case[5:17], m.group(1)[:20], m.group(2), '']; continue
    m=re.search(r'vs shipped: max\|d\| = (\S+)(.*)', l)
    if m and last: last[3]=m.group(1)+m.group(2).strip()[-12:]; continue
    m=re.search(r'loop (\d+) \(([\d.]+) per tile, ([\d.]+) tiles\).*epilogue (\d+)', l)
    if m and last: print('   ', last[0], last[1], last[2], 'us core', m.group(1), 'per-tile', m.group(2), 'epi', m.group(4), ' err', last[3]); last=None
