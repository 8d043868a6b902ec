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


BUILD_ID_MARKER = b'SGX_BUILD_ID='


def source_files():
    """Everything the library is compiled from: the translation unit, the device headers next to it, the ABI header."""
    csrc = os.path.dirname(SRC)
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.h', '.hip'))) + [os.path.join(INCLUDE, 'stratego_mi355x.h')]


def source_hash():
    """First 16 hex digits of SHA-256 over the names and contents of source_files(): the build id compiled into the library
    (-DSGX_BUILD_ID, returned by sgx_build_id())."""
    import hashlib
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
        h.update(b'\0')
    return h.hexdigest()[:16]


def read_build_id(path):
    """The build id of a library file, read from the bytes after its SGX_BUILD_ID= marker (no dlopen); None if it has none."""
    try:
        with open(path, 'rb') as fh:
            data = fh.read()
    except OSError:
        return None
    at = data.find(BUILD_ID_MARKER)
    if at < 0:
        return None
    end = data.find(b'\0', at)
    return data[at + len(BUILD_ID_MARKER):end].decode('ascii', 'replace')


def is_current(path):
    """True if `path` was compiled from exactly the sources that are here now (content hash, not file times: file times do not
    survive every way of copying a tree, and a newer-looking stale binary used to pass)."""
    return os.path.exists(path) and read_build_id(path) == source_hash()


def needs_build():
    return not is_current(LIB_PATH)


def _compile(out_path, extra_flags=(), verbose=False):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    tmp = '%s.%d.tmp' % (out_path, os.getpid())
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fvisibility=hidden', '-Wall',
           '-DSGX_BUILD_ID="%s"' % source_hash()] + list(extra_flags) + ['-I', INCLUDE, SRC, '-o', tmp]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, out_path)
    return out_path


def build(force=False, verbose=False):
    """Compile the library if it is missing or was built from other sources.  Returns the .so path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libstratego_mi355x.so")
    os.makedirs(OUT_DIR, exist_ok=True)
    return _compile(LIB_PATH, verbose=verbose)


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
    if not force and is_current(path):
        return path
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the %dx%d kernels%s; prebuild them where the ROCm compiler is installed with "
                           "`python -m stratego_env_amd.build %dx%d` and ship stratego_env_amd/_build/"
                           % (rows, columns, " (the library that is here was built from other sources)" if os.path.exists(path) else "",
                              rows, columns))
    os.makedirs(OUT_DIR, exist_ok=True)
    return _compile(path, ['-DSGX_EXTRA_R=%d' % rows, '-DSGX_EXTRA_C=%d' % columns, '-DSGX_ONLY_EXTRA'], verbose)


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
