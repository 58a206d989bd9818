import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle
from comap_amd import engine
from conftest import make_case
for (T,N,S,seed) in [(9,70,20,61),(12,64,20,164)]:
    case=make_case(T,N,S,seed)
    eng=engine.Engine(case['parent'],case['blen'],case['lot'],case['Q'],case['pi'],case['rates'],case['probs'])
    om=oracle.Model(case['parent'],case['blen'],case['lot'],case['Q'],case['pi'],case['rates'],case['probs'])
    r=eng.map_sites(case['aln']); o=oracle.map_sites(om,case['aln'])
    rel=np.abs(r['counts'][:,:,0]-o['counts'][:,:,0])/np.abs(o['counts'][:,:,0])
    print('case',T,seed,'parent',list(case['parent']))
    print(' per-branch max rel:', ' '.join('%d:%.1e'%(b,rel[:,b].max()) for b in range(rel.shape[1])))
    print(' ratio gpu/oracle for bad branches:', {b: float(np.median(r['counts'][:,b,0]/o['counts'][:,b,0])) for b in range(rel.shape[1]) if rel[:,b].max()>1e-6})
    print(' logL ok', np.max(np.abs(r['logL']-o['logL'])))
