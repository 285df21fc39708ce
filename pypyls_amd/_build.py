"""Build libplsx.so (hipcc, gfx950 only) in-tree next to this file.

The library is eleven translation units (csrc/plsx_internal.h has the map), compiled in parallel into
csrc/build/*.o and linked; a unit is recompiled when it or a header it includes is newer than its object."""
import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJDIR = os.path.join(CSRC, 'build')
UNITS = ['plsx_smallql1', 'plsx_smallql2', 'plsx_gram', 'plsx_urot', 'plsx_xprod', 'plsx_small', 'plsx_compact',
         'plsx_simpls_api', 'plsx_split', 'plsx_core', 'plsx_comm']                                   # (longest compile first)
COMMON = ['plsx_internal.h', 'plsx_kernels.h', 'plsx_common.h', 'plsx_k_prep.h', 'plsx_k_xprod.h', 'plsx_k_gram.h',
          'plsx_k_small.h', 'plsx_k_urot.h', 'plsx_k_misc.h', 'plsx_k_finish.h', 'plsx_symeig.h']
EXTRA = {'plsx_core': ['plsx_resample.h'], 'plsx_smallql1': ['plsx_smallql.h'], 'plsx_smallql2': ['plsx_smallql.h'],
          'plsx_simpls_api': ['plsx_simpls.h'], 'plsx_split': ['plsx_splitfused.h']}
PUBLIC = os.path.join(os.path.dirname(HERE), 'include', 'plsx.h')
LIB = os.path.join(HERE, 'libplsx.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function']


def lib_path():
    return LIB


def _deps(unit):
    return [os.path.join(CSRC, unit + '.hip'), PUBLIC] + [os.path.join(CSRC, h) for h in COMMON + EXTRA.get(unit, [])]


def _obj(unit):
    return os.path.join(OBJDIR, unit + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def is_stale():
    return _stale(LIB, [d for u in UNITS for d in _deps(u)])


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libplsx.so')
    os.makedirs(OBJDIR, exist_ok=True)
    todo = [u for u in UNITS if force or _stale(_obj(u), _deps(u))]

    def compile_unit(unit):
        t0 = time.time()
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, unit + '.hip'), '-o', _obj(unit)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (unit, r.stdout[-8000:]))
        if verbose:
            print('  %-16s %5.1f s' % (unit, time.time() - t0))

    t0 = time.time()
    if todo:
        if verbose:
            print(hipcc, ' '.join(FLAGS), '-c  x', len(todo))
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_unit, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread'] + [_obj(u) for u in UNITS] + ['-ldl', '-o', LIB]
    subprocess.run(cmd, check=True)
    if verbose:
        print('  linked %s in %.1f s total' % (LIB, time.time() - t0))
    return LIB
