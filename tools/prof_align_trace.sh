#!/bin/bash
# dev: kernel trace of the cfg-4 alignment loop, per-kernel durations by grid size (= by alignment level)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/align_trace
ITERS=20 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/align_trace -o a -- python tools/align8_bench.py > gpurun_out/align_trace.log 2>&1
python3 - <<'PY'
import csv,re,collections,glob
f=glob.glob('gpurun_out/align_trace/**/a_kernel_trace.csv',recursive=True)[0]
acc=collections.defaultdict(list); rows=[]
for r in csv.DictReader(open(f)):
    m=re.search(r'miso::(\w+)',r['Kernel_Name'])
    if not m: continue
    acc[(m.group(1), r['Grid_Size_X'], r['Grid_Size_Y'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),m.group(1),r['Grid_Size_X']))
for k,v in sorted(acc.items()):
    if len(v)>50: print('%-30s grid %9s,%4s calls %5d avg %8.1f us'%(k[0],k[1],k[2],len(v),sum(v)/len(v)/1e3))
rows.sort()
# gaps between consecutive kernels inside the steady loop of level 0 (pair_stage grid smallest)
gaps=collections.defaultdict(list)
for a,b in zip(rows[:-1],rows[1:]):
    gaps[(a[2],b[2])].append(b[0]-a[1])
for k,v in sorted(gaps.items(), key=lambda kv:-len(kv[1]))[:8]:
    print('gap %-28s -> %-28s n %5d avg %6.2f us'%(k[0],k[1],len(v),sum(v)/len(v)/1e3))
PY
