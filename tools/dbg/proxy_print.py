import json,sys
l=[x for x in open(sys.argv[1]).read().split('\n') if x.startswith('{')][-1]
d=json.loads(l); p=d['strong_scaling_proxy']
print(sys.argv[1], d['ms_per_step'], p.get('ms_per_step'), {k:round(v['ms_per_step'],2) for k,v in p.get('weak_scaling_B32_8rank_shape',{}).get('variants',{}).items()}, p.get('error'))
