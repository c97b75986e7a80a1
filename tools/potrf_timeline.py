"""Per-outer-step timeline of the look-ahead Cholesky from a rocprofv3 kernel trace ("Where potrf's
time goes"):   rocprofv3 --output-format csv --kernel-trace -d OUT -o run -- python bench.py --steps 2 --warmup 1
                  python tools/potrf_timeline.py OUT/run_kernel_trace.csv
For every outer step: duration of part 1 / part 2 of the trailing update, span of the panel chain on the other stream
(with the k_diag128 durations), how long the chain outlasts part 2, and the step time."""
import csv, sys
def analyze(path, verbose=False):
    rows=list(csv.DictReader(open(path)))
    def short(n):
        for k in ('k_update_nt','k_diag128','k_trsm128','k_lauum','k_trtri_stage','k_kbuild','k_grad','k_inv128','k_trmv','k_panel_fused'):
            if k in n: return k
        return n[:20]
    ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),int(r['Queue_Id']),int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))) for r in rows]
    ev.sort()
    idx=[i for i,e in enumerate(ev) if e[2]=='k_kbuild']
    s=idx[-1]
    e_end=[i for i,e in enumerate(ev) if e[2]=='k_diag128' and i>s][-1]+3     # last diagonal block (+ its solve) of this evaluation
    seg=[e for e in ev[s+1:e_end] if e[2] in ('k_update_nt','k_diag128','k_trsm128','k_panel_fused')]
    print(path, "potrf span ms %.2f"%((seg[-1][1]-ev[s][1])/1e6))
    oth={}
    for e in ev[s+1:e_end]:
        if e[2] not in ('k_update_nt','k_diag128','k_trsm128'): oth[e[2]]=oth.get(e[2],0)+(e[1]-e[0])/1e3
    if oth: print(" other kernels running during potrf (us):", {k:round(v) for k,v in oth.items()})
    upd=[e for e in seg if e[2]=='k_update_nt' and e[4]>=400 or (e[2]=='k_update_nt' and e[3]==seg[0][3] and False)]
    # classify queues: chain queue = queue of k_diag128
    qd=[e[3] for e in seg if e[2]=='k_diag128'][0]
    q1=[e for e in seg if e[3]!=qd and e[2]=='k_update_nt']
    q2=[e for e in seg if e[3]==qd]
    tot={}
    for e in q2: tot[e[2]]=tot.get(e[2],0)+(e[1]-e[0])/1e3
    print(" chain-queue kernel time sums (us):", {k:round(v) for k,v in tot.items()}, " n_diag", sum(1 for e in q2 if e[2]=='k_diag128'))
    ndiag=sum(1 for e in q2 if e[2]=='k_diag128')
    if len(q1) >= ndiag//4*2 - 4:                      # older schedule: part 1 AND part 2 of every step on the main queue
        for p in range(0,len(q1)//2):
            p1=q1[2*p]; p2=q1[2*p+1] if 2*p+1<len(q1) else None
            nxt=q1[2*p+2][0] if 2*p+2<len(q1) else seg[-1][1]
            ch=[e for e in q2 if e[0]>=p1[1] and e[0]<nxt]
            if not ch: continue
            cs,ce=ch[0][0],ch[-1][1]
            d=[(e[1]-e[0])/1e3 for e in ch if e[2]=='k_diag128']; t=sum(e[1]-e[0] for e in ch if e[2]=='k_trsm128'); u=sum(e[1]-e[0] for e in ch if e[2]=='k_update_nt')
            if verbose or p%3==0: print(" p=%2d part1 %6.1f | part2 %7.1f | chain %7.1f (diag %s trsm %6.1f upd %6.1f) | chain-part2 end %7.1f | step %7.1f"%(p,(p1[1]-p1[0])/1e3,(p2[1]-p2[0])/1e3 if p2 else 0,(ce-cs)/1e3,' '.join('%5.0f'%x for x in d),t/1e3,u/1e3,(ce-(p2[1] if p2 else ce))/1e3,(nxt-p1[0])/1e3))
        return
    # current schedule (N >= 6144): part 1 rides on the chain queue in front of the next panel's first diag, part 2 alone on the
    # main queue.  A panel = the chain-queue kernels from one part 1 (the update right after a panel's last trsm) to the next.
    panels, cur, seen_diag = [], [], 0
    for e in q2:
        if e[2]=='k_update_nt' and seen_diag>=4 and cur and cur[-1][2]=='k_trsm128':
            panels.append(cur); cur=[]; seen_diag=0
        cur.append(e)
        if e[2]=='k_diag128': seen_diag+=1
    if cur: panels.append(cur)
    print(" panels on the chain queue: %d ; part-2 launches on the main queue: %d"%(len(panels),len(q1)))
    for p,ch in enumerate(panels):
        p1=ch[0] if ch[0][2]=='k_update_nt' else None
        body=ch[1:] if p1 else ch
        d=[(e[1]-e[0])/1e3 for e in body if e[2]=='k_diag128']; t=sum(e[1]-e[0] for e in body if e[2]=='k_trsm128'); u=sum(e[1]-e[0] for e in body if e[2]=='k_update_nt')
        cs,ce=ch[0][0],ch[-1][1]
        p2=[e for e in q1 if e[0]>=cs-5000 and e[0]<ce]
        p2d=sum(e[1]-e[0] for e in p2)/1e3
        if verbose or p%3==0: print(" panel %2d part1 %7.1f | chain %7.1f (diag %s trsm %6.1f upd %6.1f) | part 2 launched meanwhile %7.1f | panel span %7.1f"%(p,(p1[1]-p1[0])/1e3 if p1 else 0,(body[-1][1]-body[0][0])/1e3,' '.join('%5.0f'%x for x in d),t/1e3,u/1e3,p2d,(ce-cs)/1e3))
for p in sys.argv[1:]: analyze(p)
