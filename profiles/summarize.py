# -*- coding: utf-8 -*-
'''
Summarise rocprofv3 CSV output into the small files committed under profiles/.

  python profiles/summarize.py stats  <dir> <out.md> [a:b | timed[:K]]  kernel-trace summary
      (a:b = ms before the end of the trace; timed = between the marker
      launches bench.py puts around its timed steps, K = steps expected there)
  python profiles/summarize.py pmc    <fetch_dir> <write_dir> <out.json> [kernel] [grid]
  python profiles/summarize.py gaps   <dir> <out.md>     idle time between kernels
  python profiles/summarize.py sequence <dir> <out.txt> a:b   launch order of a window

The PMC summary follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE come from separate passes, are in KiB, and on gfx950
FETCH_SIZE reports half of the bytes of a coalesced streaming read, so
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch.  (The x2 is
calibrated for 16-B-per-lane loads; the SpMV issues 8-B and 4-B loads, so the
absolute figure is indicative -- Infinity-Cache hits are also counted.)
'''
from __future__ import print_function

import csv
import glob
import json
import os
import sys
from collections import defaultdict


def _marker_window(directory, steps=None):
    '''(t0, t1) in trace time: from the FIRST launch of profile_marker_kernel
    with 1 workgroup to the first one with 2 (bench.py brackets the timed steps
    of every window with them; the headline window is the first).  steps: the number of time steps that
    must lie in between (counted by pressure_rhs_kernel, launched once per
    step) -- a window that silently covers something else is an error.'''
    t = {1: None, 2: None}
    once_per_step = []
    for path in _find(directory, 'kernel_trace.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                name = r['Kernel_Name']
                if 'profile_marker_kernel' in name:
                    g = int(r['Grid_Size_X']) // max(int(r.get(
                        'Workgroup_Size_X', 64) or 64), 1)
                    if g in t:
                        # the headline window is the FIRST pair of a run
                        if t[g] is None:
                            t[g] = int(r['Start_Timestamp'])
                elif 'pressure_rhs_kernel' in name:
                    once_per_step.append(int(r['Start_Timestamp']))
    if t[1] is None or t[2] is None or t[2] <= t[1]:
        raise SystemExit('no marker pair in the trace under %s' % directory)
    inside = sum(1 for x in once_per_step if t[1] <= x <= t[2])
    if steps is not None and inside != int(steps):
        raise SystemExit('%d time steps between the markers, expected %s'
                         % (inside, steps))
    return t[1], t[2], inside


def _window(directory, window, t_end):
    '''(lo, hi) start-time bounds of a window: "a:b" = between a and b ms
    before the end of the trace; "timed" / "timed:K" = between bench.py's
    markers (K: the number of steps that must lie there).'''
    if window.startswith('timed'):
        steps = window.split(':')[1] if ':' in window else None
        lo, hi, _ = _marker_window(directory, steps)
        return lo, hi
    a, b = [float(v) * 1.0e6 for v in window.split(':')]
    return t_end - b, t_end - a


def _find(directory, suffix):
    hits = glob.glob(os.path.join(directory, '**', '*' + suffix), recursive=True)
    if not hits:
        raise SystemExit('no *%s under %s' % (suffix, directory))
    return hits


def stats(directory, out, window=None):
    '''window = "a:b" as in gaps(): only kernels that start between a and b
    milliseconds before the end of the trace.'''
    rows = defaultdict(lambda: [0, 0.0])
    t_end = 0
    if window:
        for path in _find(directory, 'kernel_trace.csv'):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    t_end = max(t_end, int(r['End_Timestamp']))
        lo, hi = _window(directory, window, t_end)
    for path in _find(directory, 'kernel_trace.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if window and not (hi >= int(r['Start_Timestamp']) >= lo):
                    continue
                if 'profile_marker_kernel' in r['Kernel_Name']:
                    continue
                name = r['Kernel_Name'].split('(')[0]
                if 'spmv_stream' in name and 'Grid_Size_X' in r:
                    # one row per matrix: the same kernel serves the pressure
                    # matrix, the P2 mass matrices and the Jacobian
                    name += ' [grid %sx%s]' % (
                        r['Grid_Size_X'], r.get('Grid_Size_Y', '1'))
                dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
                rows[name][0] += 1
                rows[name][1] += dur
    total = sum(v[1] for v in rows.values())
    lines = [
        '| kernel | calls | total ms | avg us | % |',
        '|---|---|---|---|---|',
        ]
    for name, (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        lines.append('| %s | %d | %.3f | %.2f | %.1f |' % (
            name, cnt, us * 1e-3, us / cnt, 100.0 * us / max(total, 1e-30)))
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:14]))


def gaps(directory, out, threshold_us=8.0, window=None):
    '''Where the GPU waits for the host: idle intervals between consecutive
    kernels of the trace, grouped by (kernel before -> kernel after).
    window = "a:b": only the kernels that start between a and b milliseconds
    before the END of the trace (the timed steps of bench.py sit there, behind
    the setup and the Stokes start).'''
    ev = []
    for path in _find(directory, 'kernel_trace.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                           r['Kernel_Name'].split('(')[0].replace('void ', '')))
    ev.sort()
    if window:
        lo, hi = _window(directory, window, ev[-1][1])
        ev = [e for e in ev if hi >= e[0] >= lo
              and 'profile_marker_kernel' not in e[2]]
    busy = sum(e - s for s, e, _ in ev) * 1e-3
    span = (ev[-1][1] - ev[0][0]) * 1e-3
    sites = defaultdict(lambda: [0, 0.0])
    small = 0.0
    end = ev[0][1]
    prev = ev[0][2]
    for s, e, name in ev[1:]:
        gap = (s - end) * 1e-3
        if gap > threshold_us:
            key = '%s -> %s' % (prev, name)
            sites[key][0] += 1
            sites[key][1] += gap
        elif gap > 0:
            small += gap
        if e > end:
            end, prev = e, name
    lines = [
        'kernels %d, span %.1f ms, busy %.1f ms, gaps <= %.0f us: %.1f ms' % (
            len(ev), span * 1e-3, busy * 1e-3, threshold_us, small * 1e-3),
        '', '| before -> after | count | total ms | avg us |', '|---|---|---|---|',
        ]
    for key, (cnt, us) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:40]:
        lines.append('| %s | %d | %.3f | %.1f |' % (key, cnt, us * 1e-3, us / cnt))
    # context of the last few long gaps: the kernels around them
    ctx = []
    end = ev[0][1]
    for i in range(1, len(ev)):
        s, e, name = ev[i]
        if 300.0 < (s - end) * 1e-3 < 5000.0:
            ctx.append((i, (s - end) * 1e-3))
        end = max(end, e)
    lines += ['', 'context of the last long gaps (300 us .. 5 ms):']
    for i, gap in ctx[-6:]:
        lines.append('gap %.0f us:' % gap)
        for j in range(max(0, i - 5), min(len(ev), i + 4)):
            lines.append('   %s%s  %.1f us' % (
                '>> ' if j == i else '   ', ev[j][2], (ev[j][1] - ev[j][0]) * 1e-3))
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:30]))


def _counter(directory, counter, kernel, grid=None):
    '''Counter values of the dispatches of `kernel` on the pressure matrix: the
    same kernel also serves the (smaller) multigrid levels and, a few times,
    the (larger) velocity operators -- keep the largest grid that was
    dispatched at least 20 times (the timed roofline launches alone are 50).'''
    by_grid = {}
    for path in _find(directory, 'counter_collection.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] == counter and \
                        kernel in r['Kernel_Name']:
                    g = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
                    by_grid.setdefault(g, []).append(float(r['Counter_Value']))
    if grid is not None:
        return by_grid.get(int(grid), [])
    grids = [g for g, v in by_grid.items() if len(v) >= 20]
    if not grids:
        return []
    return by_grid[max(grids)]


def pmc(fetch_dir, write_dir, out, kernel='flow::spmv_stream_kernel<false>',
        grid=None):
    # group by dispatch size: keep the dispatches of the pressure matrix
    # (the most frequent value set is the CG loop + the timed roofline launches;
    # `grid`: the dispatch size to keep, e.g. 1923072 for the pressure matrix
    # of the headline workload)
    f = _counter(fetch_dir, 'FETCH_SIZE', kernel, grid)
    w = _counter(write_dir, 'WRITE_SIZE', kernel, grid)
    if not f or not w:
        raise SystemExit('no counter rows for %s' % kernel)
    f.sort()
    w.sort()
    fmed = f[len(f) // 2]
    wmed = w[len(w) // 2]
    res = {
        'kernel': kernel,
        'dispatches_fetch_pass': len(f),
        'dispatches_write_pass': len(w),
        'FETCH_SIZE_KiB_median': fmed,
        'WRITE_SIZE_KiB_median': wmed,
        'hbm_bytes_per_launch': (2.0 * fmed + wmed) * 1024.0,
        'note': 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 '
                'correction of MI355X_MICROARCH.md; median over dispatches',
        }
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


def counters(out, *dirs):
    '''Per (kernel, dispatch size): the median over its dispatches of every
    counter found in the given rocprofv3 --pmc output directories (one pass
    each), as a markdown table; rows ordered by dispatch count.  Byte columns:
    FETCH_SIZE / WRITE_SIZE are KiB; `HBM MB` = (2 FETCH + WRITE) * 1024 / 1e6
    (gfx950 correction of MI355X_MICROARCH.md).'''
    vals = defaultdict(lambda: defaultdict(list))
    names = []
    for d in dirs:
        for path in _find(d, 'counter_collection.csv'):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    k = r['Kernel_Name'].split('(')[0].replace('void ', '') \
                        .replace('flow::', '')
                    g = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
                    c = r['Counter_Name']
                    if c not in names:
                        names.append(c)
                    vals[(k, g)][c].append(float(r['Counter_Value']))

    def med(v):
        v = sorted(v)
        return v[len(v) // 2]
    rows = []
    for (k, g), cs in vals.items():
        n = max(len(v) for v in cs.values())
        if n < 5:
            continue
        row = {c: med(v) for c, v in cs.items()}
        if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
            row['HBM MB'] = (2.0 * row['FETCH_SIZE'] + row['WRITE_SIZE']) \
                * 1024.0 / 1e6
        rows.append((n, k, g, row))
    rows.sort(key=lambda r: -r[0])
    cols = names + (['HBM MB'] if 'FETCH_SIZE' in names and 'WRITE_SIZE' in names
                    else [])
    lines = ['| kernel | grid | dispatches | ' + ' | '.join(cols) + ' |',
             '|---|---|---|' + '---|' * len(cols)]
    for n, k, g, row in rows:
        lines.append('| %s | %d | %d | %s |' % (
            k[:70], g, n, ' | '.join(
                ('%.4g' % row[c]) if c in row else '' for c in cols)))
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:40]))


def sequence(directory, out, window, min_gap_us=15.0):
    '''The kernels of a window "a:b" (ms before the end of the trace) in launch
    order, runs of equal names folded, every idle interval >= min_gap_us shown:
    where in a time step the GPU waits for the host.'''
    ev = []
    for path in _find(directory, 'kernel_trace.csv'):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                           r['Kernel_Name'].split('(')[0].replace('void ', '')
                           .replace('flow::', '')))
    ev.sort()
    lo, hi = _window(directory, window, ev[-1][1])
    ev = [e for e in ev if hi >= e[0] >= lo]
    lines = []
    run_name, run_count, run_us = None, 0, 0.0
    end = ev[0][0]

    def flush():
        if run_name is not None:
            lines.append('%5d x %-60s %9.1f us' % (run_count, run_name[:60], run_us))
    for s0, e0, name in ev:
        gap = (s0 - end) * 1e-3
        if gap >= min_gap_us:
            flush()
            run_name, run_count, run_us = None, 0, 0.0
            lines.append('        ---- idle %.0f us ----' % gap)
        if name != run_name:
            flush()
            run_name, run_count, run_us = name, 0, 0.0
        run_count += 1
        run_us += (e0 - s0) * 1e-3
        end = max(end, e0)
    flush()
    with open(out, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    if sys.argv[1] == 'counters':
        counters(sys.argv[2], *sys.argv[3:])
        sys.exit(0)
    if sys.argv[1] == 'gaps':
        gaps(sys.argv[2], sys.argv[3],
             window=sys.argv[4] if len(sys.argv) > 4 else None)
    elif sys.argv[1] == 'sequence':
        sequence(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3],
              window=sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(*sys.argv[2:7])
