"""Diagnostic: which stage of prepare_clouds differs from the golden on this machine."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd.preprocess import prepare_clouds
from hotformerloc_amd import synthetic as syn
from oracle import preprocess_ref
from oracle.gen_golden_coords import CASES, raw_cloud
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'preprocess.npz'))
def cmp(a, b):
    if a.shape != b.shape: return 'shape %s vs %s' % (a.shape, b.shape)
    d = a != b
    return 'equal' if not d.any() else '%d of %d differ (cols %s), max |d| %.3e' % (d.sum(), d.size, np.unique(np.nonzero(d)[1]).tolist() if a.ndim == 2 else '-', np.abs(a.astype(np.float64) - b).max())
for name, (seed, n, kind, extent, offset, normalize, coords) in CASES.items():
    raw = raw_cloud(seed, n, kind, extent, offset)
    dev_cart = prepare_clouds([raw], coordinates='cartesian', normalize=normalize)[0].cpu().numpy()
    st = {}
    host = preprocess_ref.prepare_cloud(torch.from_numpy(raw), normalize, coords, st).numpy()
    print(name, '| oracle-on-this-host vs golden: out', cmp(host, g[name + '_out']), '| masked', cmp(st['masked'].numpy(), g[name + '_masked']))
    norm_masked = st['normalized'].numpy()
    norm_masked = norm_masked[(np.abs(norm_masked) <= 1).all(1)]
    print('   device cartesian (normalise + |x|<=1 mask) vs host torch:', cmp(dev_cart, norm_masked))
    if coords == 'cylindrical':
        dev_host = prepare_clouds([raw], coordinates=coords, normalize=normalize, cylindrical='host')[0].cpu().numpy()
        dev_dev = prepare_clouds([raw], coordinates=coords, normalize=normalize, cylindrical='device')[0].cpu().numpy()
        print('   host-mode vs golden:', cmp(dev_host, g[name + '_out']), '| vs oracle here:', cmp(dev_host, host))
        print('   device-mode vs golden:', cmp(dev_dev, g[name + '_out']), '| vs oracle here:', cmp(dev_dev, host))
        m = g[name + '_masked']
        print('   syn.cylindrical(golden masked) vs golden out:', cmp(syn.cylindrical(m), g[name + '_out']))
