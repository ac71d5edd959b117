#!/usr/bin/env python
"""Static per-tile statistics of the default NeRF MLP kernel's ISA (no GPU needed): compiles nerf_mlp_v6.hip to
assembly with the flags of the product build, cuts the main loop of nerf_mlp_bf16_v6_kernel<0, 1> at its s_barriers
(one per tile) and counts, per segment, instructions, MFMAs, ds_reads, s_nop cycles, s_waitcnt, AccVGPR moves and
epilogue VALU.  `--dump K [K ...]` prints the abbreviated instruction stream of those segments.

Round-2 reading (profiles/HISTORY.md section 2c): the first tile of a layer — 2x the time of a steady tile by cycle stamps — has
the SAME instruction mix, the same waits and the same `s_nop 10` before the epilogue as a steady tile; what does differ
between tiles is 16 extra v_accvgpr moves in the second to fourth tile of most layers (half of the activations live in
AccVGPRs at 491 registers per lane).

    python scripts/isa_tile_stats.py [--dump 17 20]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def assembly():
    from nerfactor_amd import build
    src = os.path.join(build.CSRC, 'nerf_mlp_v6.hip')
    out = os.path.join(tempfile.mkdtemp(), 'v6.s')
    cmd = [build.HIPCC] + build.FLAGS + build.PER_FILE_FLAGS.get('nerf_mlp_v6.hip', []) + [
        '-x', 'hip', '--cuda-device-only', '-S', src, '-o', out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read().split('\n')


def segments(lines, kernel='nerf_mlp_bf16_v6_kernelILi0ELi1ELb0EE'):
    label = lambda l: l.startswith('_Z') and ':' in l          # `<mangled name>: ; @<mangled name>`
    start = next(i for i, l in enumerate(lines) if label(l) and kernel in l.split(':')[0])
    end = next(i for i in range(start + 1, len(lines)) if label(lines[i]) or '.end_amdhsa_kernel' in lines[i])
    segs, cur = [], []
    for l in lines[start + 1:end]:
        t = l.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        cur.append(t)
        if t.split()[0] == 's_barrier':
            segs.append(cur)
            cur = []
    return segs


def stats(seg):
    ops = [x.split()[0] for x in seg]
    return {
        'n': len(seg), 'mfma': sum(o.startswith('v_mfma') for o in ops), 'ds_read': sum(o.startswith('ds_read') for o in ops),
        'nop_cycles': sum(int(re.search(r's_nop (\d+)', x).group(1)) + 1 for x in seg if x.startswith('s_nop')),
        'waitcnt': sum(o == 's_waitcnt' for o in ops), 'accvgpr': sum('accvgpr' in o for o in ops),
        'cvt_pk': sum('cvt_pk_bf16' in o for o in ops), 'lds_dma': sum('global_load_lds' in o for o in ops),
        'valu': sum(o.startswith('v_') and not o.startswith('v_mfma') for o in ops)}


def short(x):
    op = x.split()[0]
    if op.startswith('v_mfma'):
        return 'MFMA ' + ' '.join(re.findall(r'(?:v|a)\[\d+:\d+\]', x))
    if op in ('s_waitcnt', 's_nop'):
        return x
    if op.startswith('ds_read'):
        return 'dsr ' + x.split()[1].rstrip(',')
    if 'global_load_lds' in op:
        return 'LDS-DMA'
    if 'accvgpr' in op or 'cvt_pk' in op or op == 'v_pk_max_i16':
        return op.replace('v_accvgpr_', 'acc_').replace('v_cvt_pk_bf16_f32', 'cvt').replace('v_pk_max_i16', 'relu') + \
            ' ' + x.split()[1].rstrip(',')
    return op


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dump', type=int, nargs='*', default=[])
    args = ap.parse_args()
    segs = segments(assembly())
    print('%d barrier-delimited segments (segment i = tile i - 1; segment 1 also holds the positional encoding)' % len(segs))
    for i, seg in enumerate(segs):
        print(i, stats(seg))
    for k in args.dump:
        print('==== segment %d' % k)
        print(' | '.join(short(x) for x in segs[k]))


if __name__ == '__main__':
    main()
