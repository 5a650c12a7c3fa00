"""The reference's configuration table, produced by EXECUTING /root/reference/src/config.py (defaults) and its
cfg_from_file on every shipped experiments/*.yaml.  The module is Python 2 (dict.iteritems / has_key) and
imports easydict; both are provided as shims and the source is exec'd from the reference tree -- nothing of
it is copied here.  Run in the build container:

    python tests/golden/make_config_reference.py        # writes tests/golden/config_reference.json
"""
import copy
import glob
import json
import os
import sys
import types

REF = '/root/reference'


class EasyDict(dict):
    """attribute access + recursive conversion (what easydict.EasyDict does), with the py2 dict methods"""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def iteritems(self):
        return self.items()

    def has_key(self, k):
        return k in self


def flatten(d, pre=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten(v, pre + k + '.'))
        else:
            out[pre + k] = v if isinstance(v, (int, float, str, bool, list)) or v is None else repr(v)
    return out


def main():
    ed = types.ModuleType('easydict')
    ed.EasyDict = EasyDict
    sys.modules['easydict'] = ed
    import yaml
    if not hasattr(yaml, '_orig_load'):              # the reference calls yaml.load(f) without a Loader (PyYAML < 5)
        yaml._orig_load = yaml.load
        yaml.load = lambda f, Loader=None: yaml._orig_load(f, Loader=Loader or yaml.SafeLoader)
    src = open(os.path.join(REF, 'src', 'config.py')).read()
    mod = types.ModuleType('refconfig')
    mod.__file__ = os.path.join(REF, 'src', 'config.py')
    exec(compile(src, mod.__file__, 'exec'), mod.__dict__)
    def py2_str(d):                                   # py2: b'' IS str (the reference writes one default as b'')
        for k, v in list(d.items()):
            if isinstance(v, dict):
                py2_str(v)
            elif isinstance(v, bytes):
                dict.__setitem__(d, k, v.decode())
    py2_str(mod.cfg)
    skip = ('ROOT_DIR', 'DATA_DIR')                   # absolute paths of the reference checkout
    defaults = copy.deepcopy(mod.cfg)
    out = {'defaults': {k: v for k, v in flatten(defaults).items() if k not in skip}, 'experiments': {}}
    for y in sorted(glob.glob(os.path.join(REF, 'experiments', '*.yaml'))):
        for k in list(mod.cfg.keys()):                # restore the defaults in place
            del mod.cfg[k]
        for k, v in copy.deepcopy(defaults).items():
            mod.cfg[k] = v
        mod.cfg_from_file(y)
        out['experiments'][os.path.basename(y)] = {k: v for k, v in flatten(mod.cfg).items() if k not in skip}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config_reference.json')
    json.dump(out, open(dst, 'w'), indent=1, sort_keys=True)
    print('wrote', dst, len(out['defaults']), 'default keys,', len(out['experiments']), 'experiments')


if __name__ == '__main__':
    main()
