import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
print('ms/proof', a['ms_per_proof'], b['ms_per_proof'])
for k in sorted(a['kernels'], key=lambda k:-a['kernels'][k]['share']):
    ka=a['kernels'][k]; kb=b['kernels'].get(k,{})
    print('%-22s %8.4f %8.4f  share %.4f %.4f' % (k, ka['avg_launch_ms'], kb.get('avg_launch_ms',0), ka['share'], kb.get('share',0)))
s=a.get('roofline',{}).get('serialised'); t=b.get('roofline',{}).get('serialised')
if s and t:
    print('serialised ms/proof', s.get('ms_per_proof'), t.get('ms_per_proof'))
    for k in s.get('kernels',{}):
        print('  ser %-20s %s %s' % (k, s['kernels'][k], t['kernels'].get(k)))
