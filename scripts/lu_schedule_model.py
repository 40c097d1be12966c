"""Developer tool (CPU): the discrete-event model behind DESIGN.md 3.5's "55.7 ms for this schedule, 48.6 ms for a deadline-ordered one".
One chain engine (the main stream: tau per 64-column base panel, tau_tail once fewer than 7000 rows remain) and one GEMM engine of rate R
(the update streams together); super-panel J's update of block m costs 2 W_J W_m (n - S1_J) + W_J^2 W_m flops and needs chain J and the
update (J - 1, m); chain m needs the update (m - 1, m).  `sim`: the GEMM engine picks among the available updates by policy (eager =
oldest super-panel first, what getrf_super submits; deadline = block needed soonest first); `sim_static`: a look-ahead window of D blocks;
`sim_budget`: deferral under a flop budget of beta chain times per boundary (the variant measured in docs/EXPERIMENTS.md R5 9).
Rates are inputs, not measurements: 150 / 125 us per base panel and 60 TFLOP/s by default.  Usage: python scripts/lu_schedule_model.py"""
import heapq, sys
def sim(widths, n=16384, tau=150e-6, tau_tail=125e-6, R=60e12, policy="eager", chunk=1e-3, verbose=False):
    # super-panels
    S0=[];S1=[];s=0
    for w in widths:
        S0.append(s); s=min(n,s+w); S1.append(s)
        if s>=n: break
    M=len(S0)
    W=[S1[i]-S0[i] for i in range(M)]
    def flops(J,m):
        return 2.0*W[J]*W[m]*(n-S1[J]) + W[J]*W[J]*W[m]
    # state
    done_through=[ -1 ]*M   # block m has updates applied through super-panel done_through[m]
    chain_done=[None]*M
    t_gemm=0.0  # gemm engine free time
    t_chain=0.0
    # event-driven: we simulate in time order using simple loop: chain J starts when block J ready and chain J-1 done
    # gemm engine picks next task by policy among available (J chain done, U(J-1,m) done)
    pending=set((J,m) for J in range(M) for m in range(J+1,M))
    task_done={}
    time=0.0
    chain_start=[None]*M
    cur_chain=0
    gemm_busy_until=0.0
    gemm_cur=None
    tot_busy=0.0
    # discrete loop
    events=[]
    def block_ready(m):
        return m==0 or ((m-1,m) in task_done)
    def try_start_chain(now):
        nonlocal cur_chain
        if cur_chain<M and chain_start[cur_chain] is None and (cur_chain==0 or chain_done[cur_chain-1] is not None and chain_done[cur_chain-1]<=now) and block_ready(cur_chain) and (cur_chain==0 or task_done[(cur_chain-1,cur_chain)]<=now):
            chain_start[cur_chain]=now
            rem = n - S0[cur_chain]
            tt = tau if rem>7000 else tau_tail
            d=(W[cur_chain]/64)*tt
            heapq.heappush(events,(now+d,'chain',cur_chain))
    def avail(now):
        out=[]
        for (J,m) in pending:
            if chain_done[J] is not None and chain_done[J]<=now and (J==0 or ((J-1,m) in task_done and task_done[(J-1,m)]<=now)):
                out.append((J,m))
        return out
    def pick(av):
        if policy=="eager": return min(av,key=lambda x:(x[0],x[1]))
        if policy=="deadline": return min(av,key=lambda x:(x[1],x[0]))
        if policy.startswith("window"):
            D=int(policy[6:])
            # eager within window of D blocks, else deadline
            return min(av,key=lambda x:((0,x[0],x[1]) if x[1]<=x[0]+D else (1,x[1],x[0])))
    now=0.0
    try_start_chain(now)
    gemm_free=True
    while pending or events:
        # start gemm if free
        if gemm_free:
            av=avail(now)
            if av:
                J,m=pick(av)
                pending.discard((J,m))
                d=flops(J,m)/R
                tot_busy+=d
                heapq.heappush(events,(now+d,'gemm',(J,m)))
                gemm_free=False
        if not events: break
        t,kind,x=heapq.heappop(events)
        now=t
        if kind=='chain':
            chain_done[x]=now
            cur_chain=x+1
        else:
            task_done[x]=now
            gemm_free=True
        try_start_chain(now)
    end=max([v for v in chain_done if v is not None]+list(task_done.values()))
    if verbose:
        for J in range(M): print(J,W[J],"chain start %.1f done %.1f"%(chain_start[J]*1e3,chain_done[J]*1e3))
    return end*1e3, tot_busy*1e3
for widths in ([512,1024]+[2048]*8, [512]+[1024]*16, [512]*32, [256]*64,[128]*128):
    for pol in ("eager","deadline","window2","window3"):
        e,b=sim(widths,policy=pol)
        print(widths[:4],pol,"makespan %.1f ms (gemm busy %.1f)"%(e,b))

def sim_static(widths, D, n=16384, tau=150e-6, tau_tail=125e-6, R=60e12, Rdeep=60e12, verbose=False):
    S0=[];S1=[];s=0
    for w in widths:
        S0.append(s); s=min(n,s+w); S1.append(s)
        if s>=n: break
    M=len(S0); W=[S1[i]-S0[i] for i in range(M)]
    applied=[-1]*M   # block m has S_0..S_applied[m] applied (enqueued)
    ready_time=[None]*M  # time when block m becomes fully updated through m-1
    ready_time[0]=0.0
    gemm_free=0.0; busy=0.0
    t=0.0
    for J in range(M):
        # chain J starts when block J ready and previous chain done
        start=max(t, ready_time[J])
        rem=n-S0[J]
        d=(W[J]/64)*(tau if rem>7000 else tau_tail)
        t=start+d   # boundary J time
        if verbose: print("chain",J,"W",W[J],"start %.1f end %.1f"%(start*1e3,t*1e3))
        # urgent: block J+1 gets S_J (main/mid): treat as gemm engine task too, first
        q=[]
        if J+1<M: q.append((J+1,J))
        for m in range(J+2,min(M,J+2+D)): q.append((m,J))
        # flush everything at the end? blocks beyond window wait.
        for (m,b) in q:
            a=applied[m]+1
            fl=0.0
            for i in range(a,b+1):
                fl+=2.0*W[i]*W[m]*(n-S1[i]) + W[i]*W[i]*W[m]
            st=max(gemm_free,t)
            dur=fl/(Rdeep if b>a else R)
            gemm_free=st+dur; busy+=dur
            applied[m]=b
            if b==m-1: ready_time[m]=gemm_free
    return max(t,gemm_free)*1e3, busy*1e3
for widths in ([512,1024]+[2048]*8, [512]+[1024]*16, [512]*32,[256]*64):
    for D in (0,1,2,3,4,6,8,100):
        e,b=sim_static(widths,D)
        print(widths[:4],"static D",D,"makespan %.1f ms (gemm busy %.1f)"%(e,b))

def sim_budget(widths, beta, n=16384, tau=150e-6, tau_tail=125e-6, R=60e12, tau_est=150e-6, R_est=60e12, verbose=False, tail_rows=7000):
    S0=[];S1=[];s=0
    for w in widths:
        S0.append(s); s=min(n,s+w); S1.append(s)
        if s>=n: break
    M=len(S0); W=[S1[i]-S0[i] for i in range(M)]
    def fl(i,m): return 2.0*W[i]*W[m]*(n-S1[i]) + W[i]*W[i]*W[m]
    applied=[-1]*M
    ready=[None]*M; ready[0]=0.0
    gemm_free=0.0; busy=0.0; t=0.0; waits=0.0
    for J in range(M):
        start=max(t, ready[J]); waits+=start-t
        rem=n-S0[J]
        d=(W[J]/64)*(tau if rem>tail_rows else tau_tail)
        t=start+d
        q=[]
        if J+1<M:
            q.append((J+1,J,J))
        # mandatory: block J+2 through J
        if J+2<M: q.append((J+2,applied[J+2]+1,J))
        # budget for filler
        if J+1<M:
            est_chain=(W[J+1]/64)*tau_est
            budget=beta*est_chain*R_est - sum(fl(i,m) for (m,a,b) in q for i in range(a,b+1))
        else: budget=1e30
        m=J+3
        fill=[]
        while m<M and budget>0:
            a=applied[m]+1
            # take tasks i=a..J one at a time
            bnew=a-1
            for i in range(a,J+1):
                if budget<=0: break
                budget-=fl(i,m); bnew=i
            if bnew>=a: fill.append((m,a,bnew))
            if bnew<J: break
            m+=1
        if J+1>=M-0:
            pass
        for (m,a,b) in q+fill:
            f=sum(fl(i,m) for i in range(a,b+1))
            st=max(gemm_free,t); dur=f/R
            gemm_free=st+dur; busy+=dur
            applied[m]=b
            if b==m-1: ready[m]=gemm_free
        if verbose: print("J",J,"W",W[J],"chain %.1f-%.1f"%(start*1e3,t*1e3),"far free at %.1f"%(gemm_free*1e3),[ (m,a,b) for (m,a,b) in fill][:6])
    # final flush: anything not applied (cannot happen: blocks caught up when mandatory)
    return max(t,gemm_free)*1e3, busy*1e3, waits*1e3
for widths in ([512,1024]+[2048]*8, [512]+[1024]*16, [512]*32):
    for beta in (0.5,0.8,1.0,1.2,1.5,2.0,1e9):
        e,b,w=sim_budget(widths,beta)
        print(widths[:4],"budget beta",beta,"makespan %.1f ms (gemm busy %.1f, chain waits %.1f)"%(e,b,w))
print("robustness: actual tau 180us, R 50e12, estimates 150/60")
for widths in ([512,1024]+[2048]*8, [512]+[1024]*16):
    for beta in (0.8,1.0,1.5,1e9):
        e,b,w=sim_budget(widths,beta,tau=180e-6,R=50e12)
        print(widths[:4],"budget beta",beta,"makespan %.1f ms (gemm busy %.1f, chain waits %.1f)"%(e,b,w))
