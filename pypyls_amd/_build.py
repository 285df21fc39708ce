"""Build libplsx.so (hipcc, gfx950 only) in-tree next to this file."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'plsx_api.hip')
DEPS = [SRC, os.path.join(HERE, 'csrc', 'plsx_kernels.h'), os.path.join(HERE, 'csrc', 'plsx_simpls.h'),
        os.path.join(HERE, 'csrc', 'plsx_resample.h'),
        os.path.join(HERE, 'csrc', 'plsx_symeig.h'),
        os.path.join(os.path.dirname(HERE), 'include', 'plsx.h')]
LIB = os.path.join(HERE, 'libplsx.so')


def lib_path():
    return LIB


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libplsx.so')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
           '-munsafe-fp-atomics', '-pthread', SRC, '-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
