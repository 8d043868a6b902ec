"""vgpr / sgpr / scratch / LDS of the kernels inside a BUILT library (the notes of its gfx950 code object: seconds, no recompilation).

    python tools/kernel_notes.py <lib.so> [substring of the mangled kernel name]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def kernel_notes(so):
    with tempfile.TemporaryDirectory() as td:
        import shutil
        shutil.copy(so, os.path.join(td, 'lib.so'))
        subprocess.check_call([os.path.join(LLVM, 'llvm-objdump'), '--offloading', os.path.join(td, 'lib.so')], stdout=subprocess.DEVNULL, cwd=td)
        co = [f for f in os.listdir(td) if 'gfx950' in f][0]
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', os.path.join(td, co)], capture_output=True, text=True, check=True).stdout
    res = {}
    for blk in re.split(r'\n\s+- \.agpr_count:', notes)[1:]:
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        res[re.search(r'\.name:\s+(\S+)', blk).group(1)] = dict(vgpr=g('vgpr_count'), sgpr=g('sgpr_count'), scratch=g('private_segment_fixed_size'),
                                                                 lds=g('group_segment_fixed_size'))
    return res


if __name__ == '__main__':
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, r in sorted(kernel_notes(sys.argv[1]).items()):
        if pat in name:
            print("%-90s vgpr %3d sgpr %3d scratch %4d lds %6d" % (name[:90], r['vgpr'], r['sgpr'], r['scratch'], r['lds']))
