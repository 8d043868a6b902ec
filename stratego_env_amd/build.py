"""Builds libstratego_mi355x.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The shared library lands in stratego_env_amd/_build/ (git-ignored, travels to the GPU box with the
repo snapshot).  hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SRC = os.path.join(_PKG, 'csrc', 'stratego_mi355x.hip')
INCLUDE = os.path.join(_ROOT, 'include')
OUT_DIR = os.path.join(_PKG, '_build')
LIB_PATH = os.path.join(OUT_DIR, 'libstratego_mi355x.so')


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    csrc = os.path.dirname(SRC)
    deps = [SRC, os.path.join(INCLUDE, 'stratego_mi355x.h')] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.h')]
    newest = max(os.path.getmtime(f) for f in deps)
    return os.path.getmtime(LIB_PATH) < newest


def build(force=False, verbose=False):
    """Compile the library if it is missing or older than its sources.  Returns the .so path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libstratego_mi355x.so")
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fvisibility=hidden',
           '-Wall', '-I', INCLUDE, SRC, '-o', LIB_PATH + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + '.tmp', LIB_PATH)
    return LIB_PATH


BUILTIN_GEOMETRIES = ((10, 10), (15, 15), (8, 8), (6, 6), (5, 5), (4, 4), (3, 4))    # SGX_BUILTIN_GEOMETRIES in the .hip
MAX_CELLS = 1024


def geometry_lib_path(rows, columns):
    return os.path.join(OUT_DIR, 'libstratego_mi355x_%dx%d.so' % (rows, columns))


def build_geometry(rows, columns, force=False, verbose=False):
    """The library for a board size that is not compiled into libstratego_mi355x.so: the same sources with that one size
    (-DSGX_EXTRA_R / -DSGX_EXTRA_C / -DSGX_ONLY_EXTRA, ~20 s of hipcc), cached in _build/.  This is how any (rows, columns) the
    reference's StrategoProceduralEnv accepts (penv:27-36; here rows * columns <= 1024) gets its own specialised kernels."""
    rows, columns = int(rows), int(columns)
    if rows < 3 or columns < 3:
        raise ValueError("Both rows and columns have to be at least 3 (you passed rows: {} columns: {}).".format(rows, columns))
    if rows * columns > MAX_CELLS:
        raise ValueError("boards of more than %d cells are not supported (%d x %d)" % (MAX_CELLS, rows, columns))
    path = geometry_lib_path(rows, columns)
    csrc = os.path.dirname(SRC)
    deps = [SRC, os.path.join(INCLUDE, 'stratego_mi355x.h')] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith('.h')]
    if not force and os.path.exists(path) and os.path.getmtime(path) >= max(os.path.getmtime(f) for f in deps):
        return path
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        if os.path.exists(path) and not force:
            # a deployment box without the ROCm compiler: the library that was shipped is the one to use (file times do not survive
            # every way of copying a tree); sizes that were not prebuilt cannot be played there
            import warnings
            warnings.warn("%s may be older than its sources and hipcc is not available: using it as it is" % path, RuntimeWarning)
            return path
        raise RuntimeError("hipcc not found: cannot build the %dx%d kernels; prebuild them where the ROCm compiler is installed with "
                           "`python -m stratego_env_amd.build %dx%d` and ship stratego_env_amd/_build/" % (rows, columns, rows, columns))
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = '%s.%d.tmp' % (path, os.getpid())
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fvisibility=hidden', '-Wall',
           '-DSGX_EXTRA_R=%d' % rows, '-DSGX_EXTRA_C=%d' % columns, '-DSGX_ONLY_EXTRA', '-I', INCLUDE, SRC, '-o', tmp]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, path)
    return path


if __name__ == '__main__':
    # python -m stratego_env_amd.build            rebuilds the main library (every board size of the reference's variants)
    # python -m stratego_env_amd.build 7x7 12x12  prebuilds the libraries of other board sizes (for boxes without hipcc)
    import sys
    sizes = [a for a in sys.argv[1:] if 'x' in a]
    if not sizes:
        print(build(force=True, verbose=True))
    for a in sizes:
        r, c = a.lower().split('x')
        print(build_geometry(int(r), int(c), force=True, verbose=True))
