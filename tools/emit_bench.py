import sys; sys.path.insert(0,'.')
from phylign_amd import _lib as pm, workload as W
pm.init(0)
ix = pm.Index.synth(1, 4000, 3_000_000, seed=661)
fasta,_ = W.make_queries(4000, 150, seed=3)
q = pm.Queries(fasta)
for thr, nb in ((0.7,0),(0.0,0),(0.0,100),(0.2,0)):
    best=None
    for _ in range(3):
        r = pm.search([ix], q, thr, nb_best_hits=nb); st=r.stats; ms=st.ms_scan; n=st.n_hits; r.free()
        best = ms if best is None else min(best, ms)
    print(f"thr {thr} nb {nb}: hits {n}  scan {best:.3f} ms  -> {n/max(best,1e-9)/1e3:.1f} M hits/s")
