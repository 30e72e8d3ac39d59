#!/usr/bin/env python3
"""gpurun_out/<TAG>/ (written by tools/gpu_profile_r0N.sh on the GPU box) -> the tracked summaries under profiles/.
usage: python tools/make_profiles.py [TAG]"""
import json
import os
import re
import subprocess
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROUND = TAG[1:3].lstrip("0") or "0"
COMMIT = open(f"gpurun_out/{TAG}/commit.txt").read().strip() if os.path.exists(f"gpurun_out/{TAG}/commit.txt") else "unrecorded"
SRC, DST = f"gpurun_out/{TAG}", "profiles"
run = lambda *a: subprocess.run([sys.executable, *a], capture_output=True, text=True).stdout


def json_line(path):
    for ln in reversed(open(path).read().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


# 1. kernel trace of the default bench command
open(f"{DST}/{TAG}_kernel_stats.md", "w").write(run("tools/rocpd_summary.py", f"{SRC}/trace_results.db"))
open(f"{DST}/{TAG}_step_timeline.txt", "w").write(
    "# one hipGraph-replayed step of the timed region (q = hardware queue; rocprofv3 --kernel-trace of `python bench.py "
    "--steps 20 --warmup 5`)\n" + run("tools/step_timeline.py", f"{SRC}/trace_results.db", "0", "14"))
line = json_line(f"{SRC}/bench_under_rocprof.log")
open(f"{DST}/{TAG}_bench_under_rocprof.json", "w").write(json.dumps(line, indent=1) + "\n")
# 2. bench lines of every configuration
with open(f"{DST}/{TAG}_bench_lines.jsonl", "w") as f:
    for c in ("cfg3", "cfg3k1", "cfg2", "cfg2_laplace_dreg", "cfg5", "cfg5_bf16x3", "cfg4", "cfg4_bf16x3", "cfg4_b64_eager", "cfg4_b64", "force_dist"):
        p = f"{SRC}/bench_{c}.json"
        d = json_line(p) if os.path.exists(p) else None
        if d:
            d["_run"] = c
            f.write(json.dumps(d) + "\n")
    import glob
    for p in sorted(glob.glob(f"{SRC}/bench_ab_*.json") + glob.glob(f"{SRC}/bench_cfg4_branchmax*.json")):  # same-box A/B lines
        d = json_line(p)
        if d:
            d["_run"] = os.path.basename(p)[len("bench_"):-len(".json")]
            f.write(json.dumps(d) + "\n")
for c in ("cfg5", "cfg4", "cfg2"):  # kernel traces of the other configurations
    p = f"{SRC}/{c}_kernel_stats.md"
    if os.path.exists(p) and os.path.getsize(p) > 0:
        open(f"{DST}/{TAG}_{c}_kernel_stats.md", "w").write(open(p).read())
    p = f"{SRC}/{c}_step_groups.md"  # one step of the same trace by (kernel, grid): tools/step_groups.py
    if os.path.exists(p) and os.path.getsize(p) > 0:
        open(f"{DST}/{TAG}_{c}_step_groups.md", "w").write(open(p).read())
# 3. trainer throughput
tr = [json_line(f"{SRC}/trainer_{c}.json") for c in ("cfg3", "cfg1")]
open(f"{DST}/{TAG}_trainer_throughput.json", "w").write(json.dumps([t for t in tr if t], indent=1) + "\n")


# 4. SQ / LDS counters of the convolution kernels
def parse(path):
    rows, cols = {}, None
    for ln in open(path):
        if ln.startswith("n dur_us"):
            cols = [c.strip() for c in ln.split("|")[1:]]
            continue
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+\(\S+\)\s+((?:[\d.]+M\s+)+)(\S.*)$", ln)
        if m and cols:
            vals = [float(v.rstrip("M")) for v in m.group(3).split()]
            rows.setdefault(m.group(4).strip(), {}).update(dict(zip(cols, vals)), dur=float(m.group(2)))
    return rows


ctr = parse(f"{SRC}/pmc_conv.txt")
with open(f"{DST}/{TAG}_pmc_mfma.md", "w") as f:
    f.write(f"# Round {ROUND} - SQ / LDS counters of the register-stationary convolution kernels (rocprofv3 --pmc, kernel-trace only)\n\n"
            "Command (MI355X, tools/imgconv_pmc.sh = two PMC passes of `python tools/imgconv_probe.py prof new`: 5 launches of every\n"
            "kernel at the headline batch, n = 5120 images = K 10 x B 512):\n\n"
            "    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace ...\n"
            "    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace ...\n\n"
            "Per launch, counters in millions (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES\n"
            "cycles; 1024 waves per launch = 256 workgroups x 4 waves, one wave per SIMD):\n\n"
            "| kernel | us | INSTS_MFMA | MFMA_BUSY_CYCLES | WAVE_CYCLES | ACTIVE_INST_ANY | WAIT_INST_ANY | WAIT_ANY | INSTS_VALU | LDS_IDX_ACTIVE | LDS_BANK_CONFLICT |\n"
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    der = []
    for k, v in ctr.items():
        if "SQ_INSTS_MFMA" not in v or v["SQ_INSTS_MFMA"] == 0:
            continue
        f.write(f"| {k} | {v['dur']:.1f} | {v['SQ_INSTS_MFMA']:.3f} | {v['SQ_VALU_MFMA_BUSY_CYCLES']:.2f} | {v['SQ_WAVE_CYCLES']:.2f} | "
                f"{v['SQ_ACTIVE_INST_ANY']:.2f} | {v['SQ_WAIT_INST_ANY']:.2f} | {v['SQ_WAIT_ANY']:.2f} | {v['SQ_INSTS_VALU']:.2f} | "
                f"{v.get('SQ_LDS_IDX_ACTIVE', 0):.2f} | {v.get('SQ_LDS_BANK_CONFLICT', 0):.2f} |\n")
        cyc_wave = v["SQ_WAVE_CYCLES"] * 4e6 / 1024       # shader cycles per wave
        mfma_simd = v["SQ_VALU_MFMA_BUSY_CYCLES"] * 1e6 / 1024  # matrix-pipe busy cycles per SIMD
        der.append((k, v["dur"], cyc_wave, cyc_wave / v["dur"] / 1e3, mfma_simd / cyc_wave,
                    v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                    v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"],
                    21.47 / v["dur"] * 1e3))
    f.write("\nDerived (a launch is 21.47 GFLOP = 3.93 M v_mfma_f32_32x32x16_bf16 at 32 matrix-pipe cycles each = 122.9 k busy cycles per SIMD;\n"
            "the kernels whose last template argument is 2 are the scaled-fp16 form: 1.97 M v_mfma_f32_32x32x16_f16, 61.4 k busy cycles):\n\n"
            "| kernel | us | cycles per wave | clock GHz (cycles / duration) | matrix pipe busy | issuing | issue-stalled | parked (waitcnt / barrier) | VALU per MFMA | TFLOP/s (fp32-equivalent) |\n"
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    for r in der:
        f.write(f"| {r[0]} | {r[1]:.1f} | {r[2] / 1e3:.0f} k | {r[3]:.2f} | {100 * r[4]:.0f} % | {100 * r[5]:.0f} % | {100 * r[6]:.0f} % | "
                f"{100 * r[7]:.0f} % | {r[8]:.1f} | {r[9]:.0f} |\n")
    lo, hi = min(r[4] for r in der), max(r[4] for r in der)
    clo, chi = min(r[3] for r in der), max(r[3] for r in der)
    f.write(f"\nReading: the matrix pipe is busy {100 * lo:.0f}-{100 * hi:.0f} % of the wave's cycles; the clock under this load is "
            f"{clo:.2f}-{chi:.2f} GHz (not the 2.4 GHz of the\ndata sheet), so the practical ceiling of a 21.47-GFLOP launch is 122.9 k "
            f"cycles / {0.5 * (clo + chi):.2f} GHz = {122.9 / (0.5 * (clo + chi)):.0f} us, not 51 us.  `issue-stalled`\n"
            "is mostly the in-order wave waiting for the matrix pipe (natural when MFMA-bound); `parked` is LDS / global data not there yet\n"
            "or the barrier of the kernels whose waves split the taps.  Start of round 2 (before the scheduling pipeline, the AGPR-resident\n"
            "weights and the two-tile-latency loop): 100 / 135 / 109 / 116 / 141 / 128 us, matrix pipe 46-70 % busy.\n")

# 4a. the dense16 kernels (tools/dense16_pmc.sh: two PMC passes of tools/dense16_probe.py, M = 5120, H = 512, D = 784)
pd = f"{SRC}/pmc_dense16.txt"
if os.path.exists(pd):
    d16 = parse(pd)
    GF = 2.0 * 5120 * 784 * 512 / 1e9
    with open(f"{DST}/{TAG}_pmc_dense16.md", "w") as f:
        f.write(f"# Round {ROUND} - SQ / LDS counters of the dense16 kernels (csrc/dense16.hip: the MLP decoder on pre-split fp16 pair planes)\n\n"
                "Command (MI355X, tools/dense16_pmc.sh = two rocprofv3 --pmc passes, kernel-trace only, of `python tools/dense16_probe.py`:\n"
                "z [5120, 20] -> 512 -> 784, targets [512, 784]; every GEMM launch is 4.11 GFLOP of fp32 work = 3 fp16 MFMAs per product).\n"
                "Counters per launch in millions (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles).\n\n"
                "| kernel | us | TFLOP/s (fp32-equivalent) | frac of 833 | matrix pipe busy (of the launch) | issuing | parked | VALU per MFMA | LDS array busy (per CU) | LDS bank-conflict share |\n"
                "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, v in d16.items():
            if "SQ_INSTS_MFMA" not in v or v["SQ_INSTS_MFMA"] == 0:
                continue
            mfma_busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] * 1e6 / 1024  # busy cycles per SIMD
            clk = 2.0e3  # cycles per us at ~2.0 GHz (the probe's clock is not measured per kernel here)
            f.write(f"| {k} | {v['dur']:.1f} | {GF / v['dur'] * 1e3:.0f} | {GF / v['dur'] * 1e3 / 833.3:.3f} | "
                    f"{100 * mfma_busy / (v['dur'] * clk):.0f} % | "
                    f"{100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % | {100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % | "
                    f"{v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']:.1f} | {100 * v.get('SQ_LDS_IDX_ACTIVE', 0) * 1e6 / 256 / (v['dur'] * clk):.0f} % | "
                    f"{100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1e-9):.0f} % |\n")
        f.write("\n(matrix pipe busy / LDS busy: busy cycles per SIMD / per CU over the launch duration at 2.0 GHz; issuing / parked: share of the\n"
                "waves' resident cycles.  Nothing is saturated: the launches are bound by the issue stalls of their global loads and LDS stores\n"
                "in front of the MFMAs of an in-order wave, DESIGN.md section 9.)\n\nOther launches of the chain:\n\n| kernel | us |\n|---|---:|\n")
        for k, v in d16.items():
            if v.get("SQ_INSTS_MFMA", 0) == 0:
                f.write(f"| {k} | {v['dur']:.1f} |\n")
        pp = f"{SRC}/dense16_probe.txt"
        if os.path.exists(pp):
            f.write("\nIsolated launches and errors against float64, `python tools/dense16_probe.py` (HIP events around 20 launches incl. the Python call):\n\n```\n")
            f.write("".join(ln for ln in open(pp) if "rel" in ln or " us " in ln))
            f.write("```\n")
    print(open(f"{DST}/{TAG}_pmc_dense16.md").read())

# 4b. the register-stationary 3x3 kernels (tools/conv3_pmc.sh: one shape per kernel name)
C3_SHAPES = {"64, 64": (38.65, "64 -> 64 @64x64, n = 128"), "64, 128": (19.33, "64 -> 128 @32x32, n = 128"),
             "128, 128": (9.66, "128 -> 128 @16x16, n = 128"), "128, 256": (19.33, "128 -> 256 @16x16, n = 128"),
             "128, 64": (46.24, "128 -> 64 @14x14, n = 1600")}
C3_GFLOP = {}
for cc, (gf_, what_) in C3_SHAPES.items():  # NP = 3: bf16 pieces (6 MFMAs per product), NP = 2: scaled fp16 pairs (3)
    C3_GFLOP[f"mvk::c3rs_kernel<{cc}, true, false, 3>"] = (gf_, what_ + ", bf16x3")
    C3_GFLOP[f"mvk::c3rs_kernel<{cc}, true, false, 2>"] = (gf_, what_ + ", fp16x2")
C3_GFLOP["mvk::c3wg_kernel<3>"] = (38.65, "weight gradient 64 x 64 @64x64, n = 128, bf16x3")
C3_GFLOP["mvk::c3wg_kernel<2>"] = (38.65, "weight gradient 64 x 64 @64x64, n = 128, fp16x2")
p3 = f"{SRC}/pmc_conv3.txt"
if os.path.exists(p3):
    c3 = parse(p3)
    with open(f"{DST}/{TAG}_pmc_conv3.md", "w") as f:
        f.write(f"# Round {ROUND} - SQ / LDS counters of the register-stationary 3x3 convolution kernels (csrc/conv3rs.hip)\n\n"
                "Command (MI355X, tools/conv3_pmc.sh = two rocprofv3 --pmc passes, kernel-trace only, of `python tools/conv3_probe.py prof`:\n"
                "5 launches per kernel name, one layer shape each; backward-data form = mask + column sums in the epilogue).\n"
                "Counters per launch in millions (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles); 1024 waves per launch.\n\n"
                "| kernel | layer | us | GFLOP | TFLOP/s | cycles per wave | clock GHz | matrix pipe busy | issuing | issue-stalled | parked | VALU per MFMA | LDS bank-conflict share |\n"
                "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for k, v in c3.items():
            if "SQ_INSTS_MFMA" not in v or v["SQ_INSTS_MFMA"] == 0 or k not in C3_GFLOP:
                continue
            gf, what = C3_GFLOP[k]
            cyc = v["SQ_WAVE_CYCLES"] * 4e6 / 1024
            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] * 1e6 / 1024 / cyc
            f.write(f"| {k} | {what} | {v['dur']:.1f} | {gf:.1f} | {gf / v['dur'] * 1e3:.0f} | {cyc / 1e3:.0f} k | {cyc / v['dur'] / 1e3:.2f} | "
                    f"{100 * busy:.0f} % | {100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % | "
                    f"{100 * v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % | {100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % | "
                    f"{v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']:.1f} | "
                    f"{100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1e-9):.0f} % |\n")
        for c, title in (("f16", "cfg5 shapes, n = 128"), ("f16cfg4", "cfg4 shapes, n = 1600")):
            pp = f"{SRC}/conv3_probe_{c}.txt"
            if os.path.exists(pp):
                f.write(f"\nbf16-piece kernels vs the scaled-fp16 form, `python tools/conv3_probe.py {c}` ({title}; HIP events around 10 launches\n"
                        "incl. the Python call — launches under ~60 us are bound by the host here, the kernel-trace numbers above are the\n"
                        "device times; err = max |y - float64| / max |float64| on two images):\n\n")
                f.write("".join(ln for ln in open(pp) if ln.startswith("|") or ln.startswith("    weight")))
        for c in ("cfg5", "cfg4"):
            pp = f"{SRC}/conv3_probe_{c}.txt"
            if os.path.exists(pp):
                f.write(f"\nIsolated launches, `python tools/conv3_probe.py {c}` (HIP events around 10 launches incl. the Python call; tiled = the\n"
                        "implicit-GEMM engine, debug flag 0x400; the weight-gradient time includes the ordered finish of its slabs):\n\n")
                f.write("".join(ln for ln in open(pp) if ln.startswith("|") or ln.startswith("cfg")))
    print(open(f"{DST}/{TAG}_pmc_conv3.md").read())

# 5. HBM traffic
fs, ws = parse(f"{SRC}/pmc_FETCH_SIZE.txt"), parse(f"{SRC}/pmc_WRITE_SIZE.txt")
with open(f"{DST}/{TAG}_pmc_hbm.md", "w") as f:
    f.write(f"# Round {ROUND} - HBM traffic counters (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, kernel-trace only)\n\n"
            "Command: `rocprofv3 --pmc <C> --kernel-trace -d ... -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline` (MoPoE MnistSvhn K=10 B=512).\n"
            "FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section; calibrated\n"
            "on small_up_fwd_kernel, whose only input is the 167.8 MB tensor g3); WRITE_SIZE as reported.  Counter unit: KB.\n\n"
            "| kernel | us | read MB (2 x FETCH) | written MB | algorithmic MB | ratio |\n|---|---:|---:|---:|---:|---:|\n")
    alg = {"recon_nll_kernel<1, true>": 165.84, "small_up_fwd_kernel<3, 32, 1024, true>": 167.77 + 62.91,
           "small_up_fwd_bf_kernel<3, 512, true>": 167.77 + 6.29 + 62.91,
           "small_up_fwd_h_kernel<3, 512, true>": 167.77 + 6.29 + 62.91,
           "small_up_bwd_bf_kernel<3, true>": 62.91 + 167.77 * 2,
           "small_up_bwd_kernel<3, 32, 256, 256, 2, 2, 1, true>": 62.91 * 2 + 167.77 * 2}
    alg.update({"d16_nt_kernel<64, 0>": 10.49 + 1.61 + 1.61 + 16.06 + 0.14, "d16_nt_kernel<128, 0>": 10.49 + 1.61 + 1.61 + 16.06 + 0.14,
                "d16_nt_kernel<64, 1>": 16.06 + 1.61 + 5.24 + 10.49, "d16_tn_kernel": 16.06 + 10.49 + 14.45,
                "smallk_fwd_kernel<5, 16>": 0.41 + 41.94, "d16_first_kernel<5, 20>": 0.41 + 10.49})
    for np_ in (3, 2):  # bf16 pieces / scaled fp16 pairs: the same algorithmic bytes
        alg.update({f"mvk::imgconv_kernel<0, 8, 64, 32, false, true, {np_}>": 83.89 + 167.77,
                    f"mvk::imgconv_kernel<0, 4, 128, 64, false, true, {np_}>": 41.94 + 83.89,
                    f"mvk::imgconv_kernel<1, 8, 32, 64, true, true, {np_}>": 167.77 + 83.89 * 2,
                    f"mvk::imgconv_kernel<1, 4, 64, 128, true, true, {np_}>": 83.89 + 41.94 * 2,
                    f"mvk::imgwgrad_kernel<8, 32, 64, {np_}>": 167.77 + 83.89 + 33.55,
                    f"mvk::imgwgrad_kernel<4, 64, 128, {np_}>": 83.89 + 41.94 + 33.55})
    for k, a in alg.items():
        if k in fs and k in ws:
            rd, wr = 2 * fs[k]["FETCH_SIZE"] * 1e6 * 1024 / 1e6, ws[k]["WRITE_SIZE"] * 1e6 * 1024 / 1e6
            f.write(f"| {k} | {fs[k]['dur']:.1f} | {rd:.1f} | {wr:.1f} | {a:.1f} | {(rd + wr) / a:.3f} |\n")
    f.write("\nThe register-stationary kernels with ONE workgroup type (64<->32 channels) read every image once (83.9 MB in, 167.8 MB\n"
            "out for the forward).  The 128<->64 pair has 4 workgroup types per worker, each reading the same images: until round 4\n"
            "they sat on four XCDs (type = block % 4) and the input crossed the fabric 2.0-2.5 times (169.5 MB for a 41.9 MB tensor);\n"
            "with the types of a worker on ONE XCD (slot = block / 8 on XCD block % 8, type = slot % 4) the ratio is 1.00-1.04.\n"
            "d16_tn_kernel: slice index = XCD index (3.9x before).\n")
k = "recon_nll_kernel<1, true>"
if k in fs and k in ws:
    rd, wr = int(2 * fs[k]["FETCH_SIZE"] * 1e6 * 1024), int(ws[k]["WRITE_SIZE"] * 1e6 * 1024)
    json.dump({"kernel": k, "workload": "MoPoE MnistSvhn K=10 B=512 (both modalities in one launch), bench.py default config",
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), MI355X, round {ROUND} "
                         f"(tools/gpu_profile_{TAG}.sh); summary profiles/{TAG}_pmc_hbm.md",
               "commit": COMMIT,
               "FETCH_SIZE_KB_avg": fs[k]["FETCH_SIZE"] * 1e6, "WRITE_SIZE_KB_avg": ws[k]["WRITE_SIZE"] * 1e6,
               "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read (MI355X_MICROARCH.md): doubled; "
                             "WRITE_SIZE as reported",
               "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": 165838848}, open(f"{DST}/recon_nll_traffic.json", "w"), indent=1)
kf = next((kk for kk in fs if (kk.startswith("small_up_fwd_h_kernel") or kk.startswith("small_up_fwd_bf_kernel")) and kk in ws and "true" in kk), None)
if kf:  # the fused decoder tail: the launch that carries the large modality's reconstruction NLL
    rd, wr = int(2 * fs[kf]["FETCH_SIZE"] * 1e6 * 1024), int(ws[kf]["WRITE_SIZE"] * 1e6 * 1024)
    json.dump({"kernel": kf, "workload": "MoPoE MnistSvhn K=10 B=512, bench.py default config (fused decoder tail on)",
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), MI355X, round {ROUND} "
                         f"(tools/gpu_profile_{TAG}.sh)", "commit": COMMIT,
               "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read: doubled; WRITE_SIZE as reported",
               "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": 4 * 5120 * 256 * 32 + 4 * 5120 * 3072 + 4 * 512 * 3072},
              open(f"{DST}/fused_tail_traffic.json", "w"), indent=1)
# 6. rectifier counts of the assembled-configuration goldens, float64 distances of the IWAE / DReG gradients (jsonl written by the tests)
def jsonl(path):
    return [json.loads(ln) for ln in open(path) if ln.strip()] if os.path.exists(path) else []


fc = jsonl(f"{SRC}/flip_counts.jsonl")
if fc:
    what = json.load(open(f"{DST}/r05_flip_counts.json"))["what"] if os.path.exists(f"{DST}/r05_flip_counts.json") else ""
    json.dump({"what": what, "commit": COMMIT, "cases": fc}, open(f"{DST}/{TAG}_flip_counts.json", "w"), indent=1)
f64 = jsonl(f"{SRC}/iwae_float64.jsonl")
if f64:
    json.dump({"what": "IWAE / DReG gradients of the K >= 10 goldens (tests/test_gpu_golden.py::test_mmvae_golden / test_mmvaeplus_golden): "
                       "largest per-tensor rel-to-max distance of the fp32 CPU oracle and of the HIP path from the FLOAT64 evaluation of "
                       "the same oracle (tests/test_oracle_float64.py asserts the oracle's own distance on the CPU); the tests allow 5e-4",
               "commit": COMMIT, "cases": f64}, open(f"{DST}/{TAG}_iwae_float64.json", "w"), indent=1)
print(open(f"{DST}/{TAG}_pmc_mfma.md").read()[-2600:])
print(open(f"{DST}/{TAG}_pmc_hbm.md").read()[-1800:])
