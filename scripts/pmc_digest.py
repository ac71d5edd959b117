"""Digest rocprofv3 --pmc CSVs (gpurun_out/pmc_v*/<pass>/p_counter_collection.csv) into one JSON:
per-kernel averages per dispatch, derived MFMA utilisation and HBM traffic.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM): `hbm_read_bytes_corrected` doubles it; WRITE_SIZE is used as is (it
matches the kernel's known output bytes exactly here: 16 B per sample point)."""
import collections
import csv
import json
import os
import sys


def digest(root):
    out = {}
    for pas in sorted(os.listdir(root)):
        path = os.path.join(root, pas, 'p_counter_collection.csv')
        if not os.path.exists(path):
            continue
        rows = list(csv.DictReader(open(path)))
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r['Kernel_Name'].split('(')[0]
            per[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        for k, v in per.items():
            if not k.startswith(('void nfx', 'nfx')):
                continue
            d = out.setdefault(k, {})
            d['dispatches'] = len(disp[k])
            for c, val in v.items():
                d[c] = val / len(disp[k])
    for k, d in out.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            d['mfma_util'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024)
        if 'SQ_WAVE_CYCLES' in d:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
                if c in d:
                    d[c + '_frac_of_wave_cycles'] = d[c] / d['SQ_WAVE_CYCLES']
        if 'FETCH_SIZE' in d:
            d['hbm_read_bytes_corrected'] = 2 * d['FETCH_SIZE'] * 1024
        if 'WRITE_SIZE' in d:
            d['hbm_write_bytes'] = d['WRITE_SIZE'] * 1024
    return out


if __name__ == '__main__':
    print(json.dumps(digest(sys.argv[1]), indent=1, sort_keys=True))
