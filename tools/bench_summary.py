"""Condense bench.py JSON lines: python tools/bench_summary.py file.log [...]"""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f):
        if not line.startswith('{"metric'):
            continue
        d = json.loads(line)
        r = d["roofline"]
        print(f"{f}: {d['value']:.2f} {d['unit']}  {d['ms_per_step']:.1f} ms/step  [{d['config'].get('conv_precision')}, hw {d['config'].get('frame_hw')}]  "
              f"dominant {r['kernel']} frac {r['frac']:.3f}")
        for k, v in sorted(r.get("all_kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print(f"    {k:62s} x{v['launches_per_step']:5.1f}  {v['avg_launch_ms']:8.3f} ms  {v['achieved_tflops']:7.1f} TF/s  {v['ms_per_step']:7.2f} ms/step")
        es = r.get("encoder_span")
        if es:
            print(f"    encoder span {es['ms_per_step_in_kernel']:.1f} ms  {es['achieved']:.1f} TF/s ({es['frac']:.3f})")
