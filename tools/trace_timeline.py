import csv, collections, sys
f=sys.argv[1]; off=int(sys.argv[2]) if len(sys.argv)>2 else 600; n=int(sys.argv[3]) if len(sys.argv)>3 else 44
rows=list(csv.DictReader(open(f)))
def short(n):
    for k in ('k_resp_rows','k_resp_tile3','k_reduce_coarse_lds','k_scan_prep','k_reduce_fine'):
        if k in n: return k[2:]
    return n[:30]
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),r['Queue_Id'],r['Stream_Id']) for r in rows]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[4] not in ('1',)]
mid=ev[idx[0]+off: idx[0]+off+n]
t0=mid[0][0]
for s,e,nm,q,st in mid:
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:6.1f}  s{st} {'    '*(int(st)%2)}{nm}")
# steady per-step time in pipelined region: count resp_rows starts
rr=[e for e in ev[idx[0]:idx[-1]] if e[2]=='resp_rows' and e[4]!='1']
if len(rr)>100:
    a=rr[50][0]; b=rr[-50][0]
    print('per step us', (b-a)/1e3/(len(rr)-100))
