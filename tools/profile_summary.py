#!/usr/bin/env python3
"""Turn the output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
  profiles/<tag>_bench.json         the bench line printed under the profiler
  profiles/<tag>_traffic.json       per bench-kernel-name HBM traffic per launch from the two PMC passes
                                    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; KB -> bytes)
  profiles/traffic.json             copy of the latest HEADLINE traffic table; bench.py reads it for roofline.traffic
  profiles/<tag>_steady_kernel_stats.csv   rocprofv3 stats of the steady-state graph replay (30 + 5 iterations, nothing else)
  profiles/<tag>_steady.json (+ profiles/steady.json for the headline)   per bench-kernel-name launches per iteration and
                                    average duration from that trace; bench.py quotes roofline.frac from it
Both tables carry _meta.kernel_hash (bench.kernel_sources_hash()): bench.py ignores a table measured on other kernel sources.
Prints the per-iteration table of the steady-state trace (launches, us, share; TFLOP/s and fraction of the 2.5 PFLOP/s bf16
peak for the kernel groups whose FLOPs the bench line's roofline.by_kernel carries).
usage: tools/profile_summary.py <tag>"""
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
    m = re.search(r'wgrad_c8_kernel<(\d)>|wgrad_c8_kernelILi(\d)E', k)    # image-layer weight gradients (round 4)
    if m:
        return f'wgrad<bf16,{m.group(1) or m.group(2)}>'
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


def steady_table(src, dst, tag, meta, bench):
    """profiles/<tag>_steady_kernel_stats.csv + steady.json from the steady-state trace; prints the per-iteration table."""
    path = os.path.join(src, 'steady', 's_kernel_stats.csv')
    if not os.path.exists(path):
        print('(no steady-state trace in this profile)')
        return
    shutil.copy(path, os.path.join(dst, f'{tag}_steady_kernel_stats.csv'))
    rows = list(csv.DictReader(open(path)))
    iters = None
    for line in open(os.path.join(src, 'bench_steady.log')):
        if line.startswith('{"metric"'):
            b = json.loads(line)
            iters = b['steps'] + b['warmup']
    if iters is None:
        print('(the steady-state bench printed no JSON line)')
        return
    # The replayed iterations are found in the kernel TRACE: the launch-name sequence at the end of the run is periodic (one
    # period = one iteration: the latent draw, the input copies, the graph's kernels); everything in front of the last
    # `use` periods (pool draws, capture warm-up, the first replays) is left out.  Launch counts are then exact integers.
    groups = collections.OrderedDict()
    tot_us = tot_launch = 0.0
    tpath = os.path.join(src, 'steady', 's_kernel_trace.csv')
    period = None
    if os.path.exists(tpath):
        tr = sorted(csv.DictReader(open(tpath)), key=lambda r: int(r['Start_Timestamp']))
        names = [r['Kernel_Name'] for r in tr]
        # (the run ends with a few launches that belong to no iteration -- the loss trace's copy to the host -- so the periodic
        #  stretch is looked for at every end offset up to 64 launches)
        end = len(names)
        for off in range(0, 64):
            e = len(names) - off
            for P in range(8, min(4000, e // 2)):
                if names[e - P:e] == names[e - 2 * P:e - P]:
                    period, end = P, e
                    break
            if period:
                break
    if period:
        # as many whole periods back from there as repeat exactly (a periodic state reset of the bench loop ends the run of them)
        use = 2
        while (use + 1) * period <= end and use < b['steps'] - 1 and \
                names[end - (use + 1) * period:end - use * period] == names[end - period:end]:
            use += 1
        tr = tr[:end]
        for r in tr[-use * period:]:
            n = bench_name(r['Kernel_Name']) or ('~ ' + re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*', '', r['Kernel_Name'])[:60])
            g = groups.setdefault(n, [0, 0.0])
            g[0] += 1; g[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            tot_us += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot_launch += 1
        iters = use
        print(f'(steady iterations from the kernel trace: period {period} launches, the last {use} iterations)')
    else:
        # no trace / no period found: aggregate counts over the whole run (non-integer launch counts per iteration)
        for r in rows:
            calls, total = int(r['Calls']), float(r['TotalDurationNs']) / 1e3
            n = bench_name(r['Name']) or ('~ ' + re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*', '', r['Name'])[:60])
            g = groups.setdefault(n, [0, 0.0])
            g[0] += calls; g[1] += total
            tot_us += total; tot_launch += calls
    by = (bench or {}).get('roofline', {}) or {}
    by = by.get('by_kernel', {})
    kernels = {}
    print(f'steady state: {iters} iterations, {tot_launch / iters:.0f} launches and {tot_us / iters / 1e3:.3f} ms of kernel time per iteration')
    print(f'{"kernel group":58s} {"launches":>8s} {"avg us":>8s} {"us/iter":>9s} {"share":>6s} {"TFLOP/s":>8s} {"frac":>6s}')
    mfma_us = 0.0
    for n, (calls, total) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        per = calls / iters
        line = f'{n:58s} {per:8.1f} {total / calls:8.1f} {total / iters:9.1f} {100 * total / tot_us:5.1f}%'
        if not n.startswith('~'):
            kernels[n] = {'launches_per_step': round(per, 3), 'avg_us': total / calls, 'us_per_step': total / iters}
            mfma_us += total / iters
            e = by.get(n)
            if e and abs(e['launches'] - per) < 0.51:
                flops = e['tflops'] * 1e12 * e['total_ms'] * 1e-3           # algorithmic FLOPs of the group per iteration (eager pass)
                tf = flops / (total / iters * 1e-6) / 1e12
                kernels[n].update(tflops=tf, frac=tf / 2500.0)
                line += f' {tf:8.0f} {tf / 2500.0:6.3f}'
        print(line)
    print(f'convolution / weight-gradient kernels {mfma_us / 1e3:.3f} ms, everything else {(tot_us / iters - mfma_us) / 1e3:.3f} ms per iteration')
    out = {'_meta': dict(meta, iterations=iters, launches_per_step=tot_launch / iters, kernel_ms_per_step=tot_us / iters / 1e3),
           'kernels': kernels}
    json.dump(out, open(os.path.join(dst, f'{tag}_steady.json'), 'w'), indent=1)
    if meta.get('workload') == 'cifar10':
        json.dump(out, open(os.path.join(dst, 'steady.json'), 'w'), indent=1)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
    dst = os.path.join(ROOT, 'profiles')
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
    meta, bline = {}, None
    for line in open(os.path.join(src, 'bench_trace.log')):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(line)
            b = bline = json.loads(line)
            # what the table was measured on: bench.py attaches roofline.traffic only to a run of the same workload and kernel sources
            meta = {'workload': b['config'].get('workload_key'), 'batch': b['config'].get('batch_per_gpu'), 'dtype': b['dtype'],
                    'commit': os.popen(f'git -C {ROOT} rev-parse --short HEAD').read().strip(), 'tag': tag,
                    'kernel_hash': bench_mod.kernel_sources_hash()}
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
    if meta.get('workload') == 'cifar10':
        json.dump(table, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
    for k, v in table.items():
        if k == '_meta':
            continue
        print(f'{k:28s} n={v["launches"]:4d} fetch {v["fetch_size_kb"]:10.0f} KB  write {v["write_size_kb"]:10.0f} KB')
    steady_table(src, dst, tag, meta, bline)


if __name__ == '__main__':
    main()
