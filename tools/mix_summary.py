import json,collections
d=json.load(open('gpurun_out/mix.json'))
c=collections.Counter()
for cfg,bt,probs,flags in d['nt']:
    c[(cfg,bt,tuple(tuple(p) for p in probs))]+=1
for (cfg,bt,probs),n in sorted(c.items(), key=lambda x:(x[0][0],-x[1])):
    gf=sum(2*m*nn*k for m,nn,k in probs)/1e9
    print(cfg,bt,n,'x',f'{gf:7.2f} GF',probs[:4],'...' if len(probs)>4 else '')
