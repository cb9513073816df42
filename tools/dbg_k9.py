import sys, time, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("0g-halo2_amd","oracle","harness","tests"): sys.path.insert(0, os.path.join(ROOT,p))
import numpy as np
import orc, zg_halo2 as zg
from circuits import toy_circuit
k=int(sys.argv[1]) if len(sys.argv)>1 else 9
T=time.time()
def lap(msg):
    global T
    print("%-40s %.2f s" % (msg, time.time()-T), flush=True); T=time.time()
ctx=zg.Ctx(0)
cs, asg, ilen = toy_circuit(k, force_degree=6)
img = cs.to_c()
params = orc.params_new(k, 0xABCDEF)
vk_repr = orc.fr_from_int(0x1234567)
pk = orc.ProvingKey(img, asg.fixed_values(), asg.sigma_values(), params, vk_repr)
lap("setup oracle")
prover = zg.Prover(ctx, img, asg.fixed_values(), asg.sigma_values(), params.g_np(), params.g_lagrange_np(), vk_repr)
lap("prover create")
adv, inst = asg.advice_values(), asg.instance_values(ilen)
bad = adv.copy(); bad[2,3] = orc.fr_from_int(99)
want = orc.create_proof(pk, adv, inst, 11)[1]; lap("oracle proof")
wantbad = orc.create_proof(pk, bad, inst, 12)[1]; lap("oracle proof bad")
got = prover.prove(adv, inst, 11); lap("gpu single (latency)"); print(got==want)
prover.set_batch(6); lap("set_batch")
for overlap in (True, False):
    prover.set_overlap(overlap)
    g,_ = prover.prove_batch([adv,bad,adv,adv],[inst]*4,[11,12,13,14]); lap("batch overlap=%s"%overlap)
    g2 = prover.prove(bad, inst, 12); lap("single bad overlap=%s"%overlap)
    print(g[0]==want, g[1]==wantbad, g2==g[1])
    r = orc.verify_proof_pairing(pk, inst, g[1]); lap("pairing verify of bad proof -> %d"%r)
r = orc.verify_proof_pairing(pk, inst, want); lap("pairing verify good -> %d"%r)
