"""Discrete-time model of the decode pipeline, for choosing how many micro-batches to keep in flight:
N stages, two HIP streams each (micro-batch i on stream i % 2, in order per stream), a visit costs W
ms while both streams of the stage are busy and W / boost when one runs alone (measured on one
MI355X: a lone 2048-row micro-batch runs ~1.57 x faster), a hop adds `hop` ms to the micro-batch's
way to the next stage. Prints the throughput as a share of what the stages could do.

With exactly 2 N micro-batches every one of them is always being computed and a hop's latency is a
bubble nobody fills: 0.967 at N = 8 with 0.35 ms hops (16.8 MB of hidden state over one xGMI link),
0.90 with 1 ms; two spare micro-batches bring every case to 1.00.

    python bench/pipeline_sim.py
"""
import sys
def sim(N, M, W=7.7, boost=1.57, hop=0.35, steps=40, dt=0.01):
    # state per mb: (stage it is at or heading to, arrival time at that stage, remaining work)
    stage_of = [0]*M; arrive=[0.0]*M; rem=[W]*M; done_steps=[0]*M
    # per-stage per-stream queues in mb order
    t=0.0; finished_first=None; tokens_t=[]
    order=[[[m for m in range(M) if m%2==s] for s in range(2)] for _ in range(N)]
    ptr=[[0,0] for _ in range(N)]   # index into order[stage][stream]: which mb is next
    visits=[[ [0]*M for _ in range(2)] for _ in range(N)]
    while min(done_steps) < steps:
        for st in range(N):
            running=[]
            for s in range(2):
                q=order[st][s]
                m=q[ptr[st][s]%len(q)] if q else None
                if m is not None and stage_of[m]==st and arrive[m]<=t:
                    running.append((s,m))
            rate = 1.0 if len(running)==2 else boost
            for s,m in running:
                rem[m]-=dt*rate
                if rem[m]<=0:
                    ptr[st][s]+=1
                    rem[m]=W
                    if st==N-1:
                        done_steps[m]+=1; stage_of[m]=0; arrive[m]=t+hop*0.3   # token feedback: small message
                        tokens_t.append(t)
                    else:
                        stage_of[m]=st+1; arrive[m]=t+hop
        t+=dt
    # throughput over the second half
    half=len(tokens_t)//2
    thr=(len(tokens_t)-half)/(tokens_t[-1]-tokens_t[half])   # mb-steps per ms
    return thr, t
for N in (2,4,8):
    ideal=None
    for M in (2*N, 2*N+1, 2*N+2, 3*N):
        for hop in (0.0, 0.35, 1.0):
            thr,_=sim(N,M,hop=hop)
            print(f"N={N} M={M} hop={hop}: {thr*7.7/2:.3f} of stage capacity")
