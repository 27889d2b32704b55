#!/usr/bin/env python3
"""Turn the output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_bench.json         the bench line printed under the profiler
  profiles/<tag>_traffic.json       per bench-kernel-name HBM traffic per launch from the two PMC passes
                                    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; KB -> bytes)
  profiles/traffic.json             copy of the latest traffic table; bench.py reads it for roofline.traffic
usage: tools/profile_summary.py <tag> [workload-suffix]"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_name(k: str):
    """rocprof kernel symbol -> the name bench.py's roofline.by_kernel uses (None: not a conv/wgrad kernel)."""
    for sym, name in (('conv_img_kernel', 'conv_smap<bf16>'), ('conv_px1_kernel', 'conv_px1<bf16>'), ('conv_c8_kernel', 'conv_c8<bf16>'),
                      ('conv_head_kernel', 'conv_head<bf16>'), ('conv_skinny_kernel', 'conv_skinny<bf16>')):
        if sym in k:                                            # round 3: whole-image / resident-tile / image-layer kernels (bf16 only)
            return name
    m = re.search(r'conv_\w+?_kernelI(DF16b|f)Li(\d+)ELi(\d+)E', k)
    if m:
        return f'conv_fused<{"bf16" if m.group(1) == "DF16b" else "f32"},{m.group(2)},{m.group(3)}>'
    m = re.search(r'conv_pp_kernel(?:ILi|<)(\d+)(?:ELi|, )(\d+)', k)    # software-pipelined form: bf16 only; <BM, BN, WM, WN, R, LGW, GK, ..>
    if m:
        gk = bool(re.search(r'conv_pp_kernelILi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELb1', k) or
                  re.search(r'conv_pp_kernel<\d+, \d+, \d+, \d+, \d+, \d+, true', k))
        return f'conv_fused<bf16,{m.group(1)},{m.group(2)}{",gk" if gk else ""}>'
    m = re.search(r'conv_(gk|mc)_kernelILi(\d+)ELi(\d+)E', k)          # K-major forms: bf16 only, no dtype parameter
    if m:
        return f'conv_fused<bf16,{m.group(2)},{m.group(3)},{m.group(1)}>'
    if 'wgrad_reduce' in k:
        return None
    if 'wgrad_multi_kernel' in k:                               # several 3x3 layers of a pass in one launch (bf16 only)
        return 'wgrad_multi<bf16>'
    m = re.search(r'wgrad_ring_kernel(?:ILi|<)(\d+)', k)        # bf16-only LDS-DMA ring form
    if m:
        return f'wgrad<bf16,{m.group(1)}>'
    m = re.search(r'wgrad_(?:pc_)?kernelI(DF16b|f)Li(\d+)E', k)
    if m:
        return f'wgrad<{"bf16" if m.group(1) == "DF16b" else "f32"},{m.group(2)}>'
    if 'wgrad_pc_kernel<' in k or 'wgrad_kernel<' in k:
        # rocprofv3 mis-demangles some __bf16 instantiations; observed forms (checked against VGPR counts: 3x3 kernels
        # hold 72 accumulators): '<bool _Accum, int, ELi, ...>' = a 3x3 instantiation, '<bool _Accum, int, E, LGW[, NCH]>' = 1x1
        return 'wgrad<bf16,3>' if 'ELi' in k else 'wgrad<bf16,1>'
    return None


def pmc(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            n = bench_name(r['Kernel_Name'])
            if n:
                acc[n].append(float(r['Counter_Value']))
    return acc


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
    dst = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
    meta = {}
    for line in open(os.path.join(src, 'bench_trace.log')):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(line)
            b = json.loads(line)
            # what the table was measured on: bench.py attaches roofline.traffic only to a run of the same workload
            meta = {'workload': b['config'].get('workload_key'), 'batch': b['config'].get('batch_per_gpu'), 'dtype': b['dtype'],
                    'commit': os.popen(f'git -C {ROOT} rev-parse --short HEAD').read().strip(), 'tag': tag}
    fetch = pmc(os.path.join(src, 'fetch', 'f_counter_collection.csv'), 'FETCH_SIZE')
    write = pmc(os.path.join(src, 'write', 'w_counter_collection.csv'), 'WRITE_SIZE')
    table = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fk = sum(f) / max(1, len(f)); wk = sum(w) / max(1, len(w))
        table[k] = {'launches': len(f), 'fetch_size_kb': fk, 'write_size_kb': wk,
                    'bytes_per_launch': (2.0 * fk + wk) * 1024.0,
                    'note': 'FETCH_SIZE x2 (gfx950 half-count of wide reads) + WRITE_SIZE, averaged over the launches of one eager iteration'}
    table['_meta'] = meta
    json.dump(table, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)
    json.dump(table, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
    for k, v in table.items():
        if k == '_meta':
            continue
        print(f'{k:28s} n={v["launches"]:4d} fetch {v["fetch_size_kb"]:10.0f} KB  write {v["write_size_kb"]:10.0f} KB')


if __name__ == '__main__':
    main()
