import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d = json.loads(l)
print({k: d[k] for k in d if k not in ('kernels', 'config', 'roofline')})
print('roofline', d['roofline'])
tot = 0
for k in d['kernels']:
    print(f"{k['entry']:34s} {k['ms']:8.3f} ms   {k['alg_GBps']}")
    tot += k['ms']
print('sum of entry points', round(tot, 3), 'ms;  step', d['ms_per_step'])
