#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv of bench.py by category.  usage: prof_summary.py csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def cat(n):
    if 'msda' in n or 'tile_scan' in n: return 'native: msda'
    if 'gemm3' in n: return 'native: gemm3 (fp32 as 3xbf16)'
    if 'attn_' in n and 'anonymous' in n: return 'native: attention'
    if 'anonymous namespace)::' in n and any(k in n for k in ('match_cost', 'mask_loss', 'point_sample', 'select_unc')): return 'native: loss'
    if n.startswith('Cijk') and '_S_B_' in n: return 'gemm_fp32'
    if n.startswith('Cijk'): return 'gemm_bf16'
    if any(k in n for k in ('igemm', 'Conv', 'SubTensor', 'gridwise', 'transpose')) or 'miopen' in n.lower() or 'conv' in n.lower(): return 'conv/miopen'
    if 'elementwise' in n or 'FillFunctor' in n: return 'elementwise'
    if 'reduce_kernel' in n: return 'reduce'
    if 'upsample' in n: return 'upsample'
    if 'multi_tensor' in n: return 'optimizer/foreach'
    if 'copyBuffer' in n or 'fillBuffer' in n: return 'memcpy/memset'
    if any(k in n for k in ('layer_norm', 'RowwiseMoments', 'GroupNorm', 'group_norm', 'GammaBeta', 'cuCompute', 'ComputeInternal')): return 'norm'
    if 'softmax' in n.lower(): return 'softmax'
    if 'cat' in n.lower(): return 'cat'
    return 'other'
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"GPU busy {tot / 1e6 / steps:.2f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
cats = {}
for r in rows:
    d = cats.setdefault(cat(r['Name']), [0, 0]); d[0] += float(r['TotalDurationNs']) / 1e6 / steps; d[1] += int(r['Calls']) / steps
for c, (ms, calls) in sorted(cats.items(), key=lambda x: -x[1][0]):
    print(f"  {c:20s} {ms:8.2f} ms/step  {calls:8.0f} launches/step")
if len(sys.argv) > 3:
    for r in rows[:int(sys.argv[3])]:
        print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms calls={int(r['Calls'])/steps:7.1f} avg={float(r['AverageNs'])/1e3:8.1f}us {r['Name'][:110]}")
