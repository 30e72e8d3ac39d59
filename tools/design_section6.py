#!/usr/bin/env python3
"""Rewrites the measurement tables of DESIGN.md section 6 from profiles/r02_* (run after tools/make_profiles.py)."""
import json
import re

L = {json.loads(l)["_run"]: json.loads(l) for l in open("profiles/r02_bench_lines.jsonl")}
tr = json.load(open("profiles/r02_trainer_throughput.json"))
c3 = L["cfg3"]
rows = [l for l in open("profiles/r02_kernel_stats.md") if l.startswith("| ")][2:]
tot = helpk = 0.0
nll_roc = 0.0
for l in rows:
    c = [x.strip() for x in l.strip("|\n").split("|")]
    tot += float(c[2])
    if re.search(r"splitk_reduce|colsum|act_bwd_colsum|small_up_bwd_reduce", c[0]):
        helpk += float(c[2])
    if c[0].startswith("recon_nll_kernel<1, true>"):
        nll_roc = float(c[3])
disp = re.search(r"# (\d+) dispatches", open("profiles/r02_step_timeline.txt").read()).group(1)


def f(x, n=0):
    return f"{x:,.{n}f}"


m3, r3, im = c3["roofline_mfma"], c3["roofline"], c3["roofline_image"]
e4 = L.get("cfg4_b64_eager")
t = f"""| configuration (`bench.py --config`) | samples/s, 1 GPU | ms/step | GEMM FLOP/step → TFLOP/s | CPU oracle (16 host threads) |
|---|---:|---:|---:|---:|
| **cfg3** (headline): MoPoE MnistSvhn K=10, B=512 | **{f(c3['value'])}** | **{c3['ms_per_step']:.3f}** | 170.5 GFLOP → {m3['step_achieved']:.1f} ({m3['step_frac']:.2f} of 157.3) | {f(c3['cpu_baseline']['value'])} |
| cfg3k1: the same with the reference's single sample | {f(L['cfg3k1']['value'])} | {L['cfg3k1']['ms_per_step']:.3f} | | {f(L['cfg3k1']['cpu_baseline']['value'])} |
| cfg2: MMVAE MnistSvhn K=1, B=256, Normal / IWAE | {f(L['cfg2']['value'])} | {L['cfg2']['ms_per_step']:.3f} | | {f(L['cfg2']['cpu_baseline']['value'])} |
| cfg2, Laplace(softmax) / DReG | {f(L['cfg2_laplace_dreg']['value'])} | {L['cfg2_laplace_dreg']['ms_per_step']:.3f} | | |
| cfg5: JMVAE 64x64 CUB-ResNet image + 40 attributes, L=64, B=128 | {f(L['cfg5']['value'])} | {L['cfg5']['ms_per_step']:.2f} | 1.29 TFLOP → {L['cfg5']['roofline_mfma']['step_achieved']:.1f} | {f(L['cfg5']['cpu_baseline']['value'], 1)} (B=8) |
| cfg4: MMVAE+ PolyMNIST 5 x ResNet, K=10, 32+32, Adam(amsgrad), B=32 (hipGraph) | {f(L['cfg4']['value'], 1)} | {L['cfg4']['ms_per_step']:.1f} | 5.32 TFLOP → {L['cfg4']['roofline_mfma']['step_achieved']:.1f} | {f(L['cfg4']['cpu_baseline']['value'], 1)} (B=8) |
| cfg4 at B=64, eager launches (B=128 eager fills the 288 GB: 276 GB used, 240-700 ms per step depending on the allocator; a B=256 step does not fit) | {f(e4['value'], 1) if e4 else '—'} | {f(e4['ms_per_step'], 0) if e4 else '—'} | {e4['roofline_mfma']['step_gemm_gflop'] / 1e3:.1f} TFLOP → {e4['roofline_mfma']['step_achieved']:.1f} | |
| `BaseTrainer.train()` cfg3 / cfg1 (MVTCAE MLP, B=64) | {f(tr[0]['samples_per_s_trainer'])} / {f(tr[1]['samples_per_s_trainer'])} | | = {100 * tr[0]['samples_per_s_trainer'] / c3['value']:.0f} % of the bare cfg3 step | |
"""
k = f"""| headline step, per kernel class (`profiles/r02_kernel_stats.md`, `r02_step_timeline.txt`) | |
|---|---|
| fused reconstruction-NLL kernel | **{r3['avg_launch_us']:.1f} µs → {f(r3['achieved'])} GB/s algorithmic = {r3['frac']:.2f} of the 8 TB/s HBM peak** (bench line, device timestamps; {r3['frac_minus_kernel_boundary']:.2f} after one kernel boundary; rocprofv3 average of the same command: {nll_roc:.1f} µs); traffic 166.9 MB vs 165.8 MB algorithmic (ratio 1.006) |
| six register-stationary convolution launches (21.47 GFLOP each) | {m3['us_per_step']:.0f} µs per step → **{m3['achieved']:.0f} TFLOP/s = {m3['frac']:.2f} of 416.7** ({m3['frac_vs_fp32_input_mfma']:.2f} x the fp32-input MFMA rate); alone: 98-113 µs each, matrix pipe 61-77 % busy at 1.63-1.77 GHz (`r02_pmc_mfma.md`) |
| image-layer kernels | fwd {im['image_layer_fwd']['avg_launch_us']:.0f} µs ({im['image_layer_fwd']['frac']:.2f} of HBM), bwd {im['image_layer_bwd']['avg_launch_us']:.0f} µs ({im['image_layer_bwd']['frac']:.2f}) |
| dispatches per step | {disp} (2 batch copies in front of the graph; the noise comes from the device-resident generator inside it) — 103 at the start of the round, 94 in round 1; helper launches (`splitk_reduce*`, `colsum*`, `act_bwd_colsum`) {100 * helpk / tot:.0f} % of kernel time (17 % in round 1) |
| MVK_FORCE_DIST=1 (RCCL all-reduce of the 6.2 MB flat buffer on one GPU) | {L['force_dist']['ms_per_step']:.3f} ms/step (+{100 * (L['force_dist']['ms_per_step'] / c3['ms_per_step'] - 1):.1f} %) |
"""
s = open("DESIGN.md").read()
a = s.index("| configuration (`bench.py --config`)")
b = s.index("\nHeadline step: history of the round")
s = s[:a] + t + s[b:]
a = s.index("| headline step, per kernel class")
b = s.index("\nJoint-likelihood evaluation")
s = s[:a] + k + s[b:]
open("DESIGN.md", "w").write(s)
print(t)
print(k)
