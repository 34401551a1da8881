#!/usr/bin/env python3
"""Development: the few numbers of a bench.py line one looks at first.  usage: bench_summary.py line.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.4g frames/s, %.2f us/step; single stream %.2f us/step; regions %s" % (
    d["value"], d["ms_per_step"] * 1e3, d["single_stream"]["ms_per_step"] * 1e3, {k: round(v, 3) if isinstance(v, float) else v for k, v in d["timed_regions"].items() if k != "note"}))
r = d["roofline"]
print("roofline", r["kernel"], "frac %.3f" % r["frac"], r["all_kernels_avg_us"], "traffic", r.get("traffic"), "busy", (r.get("step_datapath_busy") or {}).get("value"))
for k in ("alt_fp32_fft_frontend", "alt_crnn_split_bf16", "fast_profile"):
    if d.get(k):
        print(k, "%.2f us/step" % (d[k]["ms_per_step"] * 1e3))
if d.get("wavenet"):
    print("wavenet bf16x3 %.2f us/step, fp32 %.2f" % (d["wavenet"]["bf16x3"]["ms_per_step"] * 1e3, d["wavenet"]["fp32_mfma_parity_mode"]["ms_per_step"] * 1e3),
          d["wavenet"]["bf16x3"]["roofline"]["all_kernels_avg_us"])
if d.get("streaming"):
    print("streaming", {k: {a: (round(b, 4) if isinstance(b, (int, float)) else b) for a, b in v.items()} for k, v in d["streaming"].items() if isinstance(v, dict)})
ev = d.get("eval_testset")
if ev:
    print("eval: %.4f s, frr %.6f, oracle %s" % (ev["seconds_host_pcm_in_to_curves_out"], ev["frr_at_0.5_fa_per_hour"], ev.get("oracle", {}).get("fa_counts_identical")))
    if ev.get("at_scale"):
        a = ev["at_scale"]
        print("at_scale:", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items() if k not in ("workload", "note")}))
if d.get("cpu_baseline"):
    print("cpu baseline %.4g (%d threads), one thread %.4g" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread"]["value"]))
