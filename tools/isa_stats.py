"""Register / LDS / scratch figures of every kernel in the library's gfx950 ISA (hipcc -S), optionally the instruction mix of one.

    python tools/isa_stats.py [substring of the demangled kernel name] [-D...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stratego_env_amd import build as hip_build  # noqa: E402


def main():
    defs = [a for a in sys.argv[1:] if a.startswith('-D')]
    pat = [a for a in sys.argv[1:] if not a.startswith('-D')]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-I',
                               hip_build.INCLUDE, hip_build.SRC, '-o', out] + defs, stderr=subprocess.DEVNULL)
        text = open(out).read()
    for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', text, re.S):
        blk = m.group(0)
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'\(anonymous namespace\)::', '', dem)
        if pat and not all(p in dem for p in pat):
            continue
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        print("%-70s vgpr %3d sgpr %3d scratch %4d lds %6d" % (dem[:70], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'),
                                                                g('group_segment_fixed_size')))
        if pat:
            body = text[text.index(name + ':'):]
            body = body[:body.index('s_endpgm')]
            ops = re.findall(r'^\s+([a-z_0-9]+)', body, re.M)
            mix = {}
            for o in ops:
                k = 'valu' if o.startswith('v_') else 'salu' if o.startswith('s_') else 'lds' if o.startswith('ds_') else \
                    'vmem' if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'
                mix[k] = mix.get(k, 0) + 1
            print("   static instruction mix:", mix)


if __name__ == '__main__':
    main()
