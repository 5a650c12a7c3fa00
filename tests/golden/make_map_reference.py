"""Golden vectors for the eval consumer computed by the REFERENCE'S OWN code:
/root/reference/src/eval/cap_eval_utils.py (calc_pr_ovr_noref, voc_ap) and src/eval/utils.py (compute_map)
are pure numpy.  They are Python 2 sources (`print` statements, `xrange`) that import IPython for a
debugger hook, so the two files are READ from the reference tree, cut before the two report printers, the
remaining print statements parenthesised in memory, `xrange` / the IPython module are provided as shims, and the code is exec'd -- nothing of it is copied
into this repository.  Run in the build container (the GPU box has no reference tree):

    python tests/golden/make_map_reference.py        # writes tests/golden/map_reference.npz
"""
import os
import re
import sys
import types

import numpy as np

REF = '/root/reference/src/eval'


def load_reference():
    dbg = types.ModuleType('IPython.core.debugger')
    dbg.Tracer = lambda *a, **k: (lambda *a2, **k2: None)
    sys.modules.setdefault('IPython', types.ModuleType('IPython'))
    sys.modules.setdefault('IPython.core', types.ModuleType('IPython.core'))
    sys.modules['IPython.core.debugger'] = dbg
    src = open(os.path.join(REF, 'cap_eval_utils.py')).read()
    src = src[:src.index('def print_benchmark_latex')]     # the two report printers (multi-line py2 print statements) are not needed
    src = re.sub(r'^(\s*)print (.*?);?\s*$', r'\1print(\2)', src, flags=re.M)      # py2 print statements
    cap = types.ModuleType('eval.cap_eval_utils')
    cap.__dict__['xrange'] = range
    exec(compile(src, os.path.join(REF, 'cap_eval_utils.py'), 'exec'), cap.__dict__)
    pkg = types.ModuleType('eval')
    pkg.cap_eval_utils = cap
    sys.modules['eval'] = pkg
    sys.modules['eval.cap_eval_utils'] = cap
    usrc = open(os.path.join(REF, 'utils.py')).read()
    utils = types.ModuleType('eval.utils')
    exec(compile(usrc, os.path.join(REF, 'utils.py'), 'exec'), utils.__dict__)
    return cap, utils


def main():
    cap, utils = load_reference()
    rng = np.random.RandomState(20260926)
    out = {}
    cases = [(60, 7, 'plain'), (200, 12, 'ties'), (35, 9, 'missing'), (500, 393, 'mpii'), (97, 51, 'hmdb')]
    for ci, (n, k, kind) in enumerate(cases):
        logits = rng.randn(n, k).astype(np.float32)
        labels = rng.randint(0, k, size=n)
        if kind == 'ties':                      # repeated scores: the argsort()[::-1] tie order matters
            logits = np.round(logits * 2) / 2
        if kind == 'missing':                   # classes without a positive are skipped by compute_map
            labels = rng.randint(0, 4, size=n)
        stdout, sys.stdout = sys.stdout, open(os.devnull, 'w')     # compute_map prints the skipped classes
        try:
            m, aps = utils.compute_map(logits, labels)
        finally:
            sys.stdout.close()
            sys.stdout = stdout
        out['c%d_logits' % ci] = logits
        out['c%d_labels' % ci] = labels.astype(np.int64)
        out['c%d_map' % ci] = np.float64(m)
        out['c%d_aps' % ci] = np.asarray(aps, dtype=np.float64).reshape(-1)
        cid = int(labels[0])
        P, R, score, ap = cap.calc_pr_ovr_noref((labels == cid).astype('float32'), logits[:, cid])
        out['c%d_cid' % ci] = np.int64(cid)
        out['c%d_P' % ci], out['c%d_R' % ci], out['c%d_score' % ci] = P, R, score
        out['c%d_ap' % ci] = np.float64(np.asarray(ap).reshape(-1)[0])
    out['n_cases'] = np.int64(len(cases))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'map_reference.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, {k: (float(out[k]) if out[k].shape == () else out[k].shape) for k in out if k.endswith('_map')})


if __name__ == '__main__':
    main()
