"""Summarise a rocprofv3 *_kernel_stats.csv into per-iteration categories."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def cat(name):
    if name.startswith('Cijk'):
        return 'vendor gemm'
    if 'gemm16_kernel' in name:
        return 'rrl gemm (MFMA)'
    if 'multi_tensor' in name:
        return 'torch foreach/optimizer'
    if 'reduce_kernel' in name:
        return 'torch reduce'
    if 'elementwise' in name or 'CatArray' in name or 'index' in name or 'distribution' in name:
        return 'torch elementwise'
    if 'anonymous namespace' in name and 'at::' not in name:
        return 'rrl other: ' + name.split('(anonymous namespace)::')[1].split('(')[0][:40]
    return 'other: ' + name[:50]


# kernels of the planner roofline probe that bench.py runs after the timed loop: not part of the iteration
PROBE = ('plan_cost_kernel', 'plan_finish_kernel', 'pack_layer_kernel', 'pack_input_kernel', 'pack_vector_kernel',
         'pack_head_kernel')

agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if any(p in r['Name'] for p in PROBE):
        continue
    k = cat(r['Name'])
    agg[k][0] += int(r['Calls'])
    agg[k][1] += float(r['TotalDurationNs'])
tot_n = tot_t = 0
for k, (n, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print('%-52s launches/iter %7.1f   us/iter %8.1f   avg us %6.2f' % (k, n / steps, t / 1e3 / steps, t / 1e3 / n))
    tot_n += n
    tot_t += t
print('%-52s launches/iter %7.1f   us/iter %8.1f' % ('TOTAL', tot_n / steps, tot_t / 1e3 / steps))
