#!/usr/bin/env python
"""profiles/rNN/window_timeline.md: the mode the headline runs in, seen by rocprofv3 --kernel-trace (NO counters: under --pmc dispatches serialize and the gate in front of
the Mixer bank runs into its time limit).  From the per-dispatch start / end of k_eq_three_spec_tiled and k_mixer in the default (second-stream) schedule and in the
one-stream schedule (MX_OVERLAP_AUTO=0): how much of each EqThree launch the Mixer bank of the step before overlaps, and what either kernel takes beside the other / alone.
usage: python tools/window_timeline.py <kernel_trace_default.csv> <kernel_trace_one_stream.csv> <strips> <ticks> > window_timeline.md"""
import csv
import statistics
import sys

f_auto, f_one, strips, ticks = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
frames = ticks * 800
EQ_B, MIX_B = 8 * strips * frames, (4 * strips + 16) * frames


def load(path):
    eq, mix, gate = [], [], []
    for r in csv.DictReader(open(path)):
        n, s, e = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "k_eq_three_spec_tiled" in n:
            eq.append((s, e))
        elif "k_mixer" in n:
            mix.append((s, e))
        elif "k_tail_gate" in n:
            gate.append((s, e))
    return sorted(eq), sorted(mix), sorted(gate)


def big(xs):   # the launches of the timed shape (drop warm-up oddities: keep those within 30 % of the median duration)
    if not xs:
        return xs
    m = statistics.median(e - s for s, e in xs)
    return [(s, e) for s, e in xs if 0.7 * m <= e - s <= 1.3 * m]


def ms(xs):
    return statistics.mean(e - s for s, e in xs) / 1e6 if xs else float("nan")


eq_a, mix_a, gate_a = load(f_auto)
eq_1, mix_1, _ = load(f_one)
eq_a, mix_a, eq_1, mix_1 = big(eq_a), big(mix_a), big(eq_1), big(mix_1)
ov, tot = 0, 0
per = []
for s, e in eq_a:
    o = sum(max(0, min(e, me) - max(s, ms_)) for ms_, me in mix_a)
    ov += o; tot += e - s
    per.append(o / (e - s))
step_a = (eq_a[-1][0] - eq_a[0][0]) / (len(eq_a) - 1) / 1e6 if len(eq_a) > 1 else float("nan")
step_1 = (eq_1[-1][0] - eq_1[0][0]) / (len(eq_1) - 1) / 1e6 if len(eq_1) > 1 else float("nan")
print(f"# The headline's window, from `rocprofv3 --kernel-trace` alone ({strips} strips x {ticks} ticks per step; tools/window_timeline.py)\n")
print("Two passes of `python bench.py --headline-only --no-headline-parity --steps 10 --warmup 2` under `rocprofv3 --kernel-trace` (no `--pmc`: counter passes serialize the")
print("dispatches and the gate in front of the bank then runs to its 300 us limit): the default schedule and `MX_OVERLAP_AUTO=0` (every launch group on one stream).\n")
print("| | default: Mixer bank of step k beside EqThree of step k + 1 | one stream |\n|---|---|---|")
print(f"| `k_eq_three_spec_tiled` launches kept | {len(eq_a)} | {len(eq_1)} |")
print(f"| EqThree launch, mean | {ms(eq_a):.3f} ms = hbm {EQ_B / (ms(eq_a) * 1e-3) / 8e12:.3f} | {ms(eq_1):.3f} ms = hbm {EQ_B / (ms(eq_1) * 1e-3) / 8e12:.3f} |")
print(f"| `k_mixer` launch, mean | {ms(mix_a):.3f} ms | {ms(mix_1):.3f} ms = hbm {MIX_B / (ms(mix_1) * 1e-3) / 8e12:.3f} |")
print(f"| start-to-start of consecutive EqThree launches (a step) | {step_a:.3f} ms | {step_1:.3f} ms |")
print(f"| share of an EqThree launch's time with a `k_mixer` running beside it | {ov / max(1, tot):.3f} (min {min(per):.2f}, max {max(per):.2f}) | 0 |")
if gate_a:
    print(f"| `k_tail_gate` (one wave spinning until the EqThree launch's last workgroup is placed), mean | {ms(gate_a) * 1e3:.1f} us over {len(gate_a)} | - |")
print(f"\nBoth kernels' bytes over the EqThree launch in the default schedule: {(EQ_B + MIX_B) / (ms(eq_a) * 1e-3) / 8e12:.3f} of 8 TB/s (the line's `roofline.window_frac`); "
      f"every byte of a step over the step: {(EQ_B + MIX_B) / (step_a * 1e-3) / 8e12:.3f} (`roofline.step_hbm_frac`).")
