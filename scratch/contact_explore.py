"""Exploration: near-contact records and episode ends in the reference's metadata log."""
import numpy as np
REF='/root/reference/data/metadata_MovingObstaclesNoRules.csv'
rows=[np.array(l.split(),dtype=np.float64) for l in open(REF)]
n=len(rows)
speed=np.array([r[0] for r in rows]); head=np.array([r[1] for r in rows])
# episode starts: speed returns to 0 (first row of an episode has speed 0?)
starts=[0]+[i for i in range(1,n) if speed[i]==0.0 and speed[i-1]!=0.0]
print('rows',n,'episodes',len(starts))
ends=[s-1 for s in starts[1:]]+[n-1]
ep_of=np.zeros(n,int)
for k,s in enumerate(starts): ep_of[s:]=k
# near-contact records
print('--- records with dist < 5')
for i,r in enumerate(rows):
    rec=r[2:].reshape(-1,5)
    for j,q in enumerate(rec):
        if q[1]<5.0:
            st = (q[0]==0 and q[2]==r[0])
            k=ep_of[i]
            print(f'row {i} (1-based {i+1}) ep {k} row-in-ep {i-starts[k]} of {ends[k]-starts[k]+1}: {"static" if st else "moving"} dist {q[1]:.3f} bearing {np.degrees(q[3]):.1f} speed_os {r[0]:.3f} rows-to-end {ends[k]-i}')
print('--- episode ends: last row min dist, speed, length')
for k,(s,e) in enumerate(zip(starts,ends)):
    rec=rows[e][2:].reshape(-1,5)
    md=rec[:,1].min()
    # min dist over the episode
    mn=min(rows[i][2:].reshape(-1,5)[:,1].min() for i in range(s,e+1))
    print(k,s,e,'len',e-s+1,'last-row min dist %.2f'%md,'episode min dist %.2f'%mn,'last speed %.3f'%speed[e], 'last heading %.3f'%head[e])
