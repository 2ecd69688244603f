import numpy as np, sys, itertools
names=sys.argv[1:]
d={n:np.load(n) for n in names}
keys=[k for k in d[names[0]].files if k!='loss']
for a,b in itertools.combinations(names,2):
    worst=[]
    for k in keys:
        r=np.abs(d[a][k]-d[b][k]).max()/max(np.abs(d[b][k]).max(),1e-30)
        worst.append((r,k))
    worst.sort(reverse=True)
    print(a.split('/')[-1],b.split('/')[-1],' '.join(f"{k}:{r:.1e}" for r,k in worst[:3]))
