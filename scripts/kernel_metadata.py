#!/usr/bin/env python
"""Code-object metadata of every kernel of a built libnfx.so (no GPU needed): registers, spills, scratch, LDS, workgroup size.

    python scripts/kernel_metadata.py [path/to/libnfx.so] [--scratch]      # --scratch: only the kernels that use scratch memory

llvm-objdump --offloading unbundles the gfx950 code objects (one per translation unit), llvm-readelf --notes prints the
AMDGPU metadata.  Used by tests/test_cpu_kernel_metadata.py: round 6 found that brdf_compact_kernel<2, 0, 8> — the one kernel
that returns wrong rows with two waves per SIMD — is also the only kernel of lvis_v2.hip whose
register allocation spills to SCRATCH MEMORY, and the shipped two-waves-per-SIMD kernels are pinned to "no scratch"."""
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
FIELDS = ('private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size', 'vgpr_count', 'agpr_count',
          'vgpr_spill_count', 'sgpr_spill_count', 'sgpr_count')


def kernels(lib):
    """[{name (demangled), symbol, <FIELDS>}] of every kernel in `lib`."""
    tmp = tempfile.mkdtemp(prefix='nfx_md_')
    try:
        local = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows = []
        for obj in sorted(glob.glob(os.path.join(tmp, 'lib.so.*amdgcn*'))):
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', obj], check=True, capture_output=True, text=True).stdout
            for chunk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
                chunk = '.agpr_count' + chunk
                row = {'symbol': re.search(r'\.name:\s+(\S+)', chunk).group(1)}
                for f in FIELDS:
                    m = re.search(r'\.%s:\s+(\d+)' % f, chunk)
                    row[f] = int(m.group(1)) if m else None
                rows.append(row)
        names = subprocess.run(['c++filt'], input='\n'.join(r['symbol'] for r in rows), capture_output=True, text=True).stdout.split('\n')
        for r, n in zip(rows, names):
            r['name'] = n
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'nerfactor_amd', 'libnfx.so')
    rows = kernels(lib)
    if '--scratch' in sys.argv:
        rows = [r for r in rows if r['private_segment_fixed_size']]
    for r in rows:
        print(json.dumps({k: r[k] for k in ('name',) + FIELDS}))
