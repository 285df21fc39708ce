"""bench.py against a variant library: python tools/bench_lib.py <name | path | base> <bench.py arguments>
(name -> tools/bin/*/libplsx_<name>.so; `base` = the shipped pypyls_amd/libplsx.so).  A/B tool only."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1]
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
if which != 'base':
    path = which if os.path.exists(which) else (glob.glob(os.path.join(ROOT, 'tools', 'bin', '*', 'libplsx_%s.so' % which)) + [''])[0]
    if not path:
        raise SystemExit('no variant library ' + which)
    from pypyls_amd import _build
    _build.LIB = path
    _build.is_stale = lambda: False
import bench                                           # noqa: E402
bench.main()
