import json,sys
a=json.loads(open(sys.argv[1]).readlines()[-1]); b=json.loads(open(sys.argv[2]).readlines()[-1])
def flat(o,p=''):
    out={}
    for k,v in o.items():
        if isinstance(v,dict): out.update(flat(v,p+k+'.'))
        elif isinstance(v,(int,float)) and not isinstance(v,bool): out[p+k]=v
    return out
fa,fb=flat(a),flat(b)
for k in fa:
    if k in fb and ('ms_per_step' in k or 'avg_launch_ms' in k or 'us_per_point' in k or 'wall_s' in k and 'leg_wall' not in k) and fa[k]:
        print('%-60s %10.4f %10.4f  %+.1f%%'%(k,fa[k],fb[k],100*(fb[k]/fa[k]-1)))
