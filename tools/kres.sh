#!/bin/bash
# kernel resource usage of one .hip file: name, VGPRs, spills, occupancy, LDS
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':m.group(1)};rows.append(cur);continue
    for k in ['VGPRs','AGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','VGPR Spill']:
        m=re.search(r'remark: .*?\s+'+k+r': (\d+)',l)
        if m and cur is not None: cur[k.split(' ')[0]]=m.group(1)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,n in zip(rows,names):
    n=re.sub(r'\(.*','',n)
    print('%-70s vgpr=%s agpr=%s sgpr=%s scratch=%s occ=%s lds=%s'%(n[:70],r.get('VGPRs'),r.get('AGPRs'),r.get('SGPRs'),r.get('ScratchSize'),r.get('Occupancy'),r.get('LDS')))
"
