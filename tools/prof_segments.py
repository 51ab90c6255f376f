#!/usr/bin/env python3
"""Per-phase launches / busy / span of one steady-state step from a rocprofv3 kernel trace of bench.py.
Phases are cut at marker kernels.  usage: prof_segments.py <kernel_trace.csv> <warmup_steps>"""
import csv, sys, collections
trace, warmup = sys.argv[1], int(sys.argv[2])
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, (s, e, n) in enumerate(rows) if ("FusedAdam" in n or "opt_adamw_kernel" in n)]
# last FusedAdam launch of each step = step end
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or rows[adam[i + 1]][0] - rows[adam[i]][1] > 5e6]
acc = collections.OrderedDict()
kinds = {}
ktime = {}
def short(n):
    n = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:90]
nsteps = 0
for si in range(warmup, len(ends) - 1):
    seg = rows[ends[si] + 1: ends[si + 1] + 1]
    nsteps += 1
    names = [n for _, _, n in seg]
    def first(pred, start=0):
        for i in range(start, len(seg)):
            if pred(seg[i][2]):
                return i
        return len(seg)
    i_pix = first(lambda n: "RowwiseMoments" in n or "gn_chunk_stats" in n)
    i_dec = first(lambda n: "attn_mask_kernel" in n, i_pix)
    i_crit = first(lambda n: "point_sample_kernel" in n or "match_cost" in n, i_dec)
    i_bwd = first(lambda n: "mask_loss_bwd" in n, i_crit)
    i_decb = first(lambda n: "attn_bwd" in n, i_bwd)
    i_encb = first(lambda n: "msda_bwd_push" in n, i_decb)
    last_enc = max([i for i, n in enumerate(names) if "gemm3_nt" in n] or [i_encb])
    i_opt = first(lambda n: ("multi_tensor" in n and "Norm" in n) or "opt_sqnorm_kernel" in n, last_enc)
    cuts = [("backbone fwd", 0, i_pix), ("pixel decoder fwd", i_pix, i_dec), ("decoder fwd", i_dec, i_crit),
            ("criterion fwd (+matching)", i_crit, i_bwd), ("criterion bwd", i_bwd, i_decb), ("decoder bwd", i_decb, i_encb),
            ("encoder bwd", i_encb, last_enc + 1), ("input proj + backbone bwd", last_enc + 1, i_opt), ("clip + AdamW", i_opt, len(seg))]
    for name, a, b in cuts:
        if b <= a:
            continue
        part = seg[a:b]
        busy = sum(e - s for s, e, _ in part) / 1e6
        nxt = seg[b][0] if b < len(seg) else part[-1][1]
        span = (nxt - part[0][0]) / 1e6
        d = acc.setdefault(name, [0, 0.0, 0.0])
        d[0] += len(part); d[1] += busy; d[2] += span
        kc = kinds.setdefault(name, collections.Counter())
        kt = ktime.setdefault(name, collections.Counter())
        for s_, e_, n_ in part:
            kc[short(n_)] += 1
            kt[short(n_)] += (e_ - s_) / 1e6
print(f"{'phase':28s} {'launches':>9s} {'busy ms':>9s} {'span ms':>9s} {'idle ms':>9s} {'us/launch':>10s}")
tl = tb = ts = 0
for k, (c, b, s) in acc.items():
    c, b, s = c / nsteps, b / nsteps, s / nsteps
    tl += c; tb += b; ts += s
    print(f"{k:28s} {c:9.0f} {b:9.2f} {s:9.2f} {s - b:9.2f} {s / c * 1e3:10.1f}")
print(f"{'total':28s} {tl:9.0f} {tb:9.2f} {ts:9.2f} {ts - tb:9.2f}")

if len(sys.argv) > 3:
    for k, kc in kinds.items():
        print(f"--- {k}: kernels by time (ms per step, launches per step)")
        for n, t in ktime[k].most_common(int(sys.argv[3])):
            print(f"   {t / nsteps:7.3f} ms  x{kc[n] / nsteps:6.1f}  {n}")
