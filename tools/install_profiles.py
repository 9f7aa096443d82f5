"""Copy the artefacts of tools/_measure.sh (gpurun_out/m_*) into profiles/ as the round's final evidence."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"

subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f"{G}/m_pmc_fetch", f"{G}/m_pmc_write", "gemm_stream|mlp_fused",
                       f"{P}/{tag}_pmc_linear_gemm_stream_traffic.json"], stdout=subprocess.DEVNULL)
# whole step: every kernel of the 4 profiled steps (3 timed + 1 warm-up), bytes per step
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f"{G}/m_pmc_fetch", f"{G}/m_pmc_write", ".",
                       f"{P}/{tag}_pmc_step_traffic.json", "4"], stdout=subprocess.DEVNULL)
for cfg in ("cfg2", "cfg4", "cfg5", "cfg4_train", "cfg5_train", "cfg3_use_checkpoint"):
    if os.path.exists(f"{G}/m_bench_{cfg}.json"):
        shutil.copy(f"{G}/m_bench_{cfg}.json", f"{P}/{tag}_bench_line_{cfg}.json")
    st = glob.glob(f"{G}/m_prof_{cfg}/*/*_kernel_stats.csv")
    if st:
        shutil.copy(st[0], f"{P}/{tag}_{cfg}_kernel_stats.csv")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), f"{G}/m_pmc_fetch", f"{G}/m_pmc_write",
                       "wgrad|attn|taps|gemm_kernel<1|imghead|smallconv", f"{P}/{tag}_pmc_other_kernels_traffic.json"], stdout=subprocess.DEVNULL)
stats = glob.glob(f"{G}/m_prof/*/*_kernel_stats.csv")[0]
shutil.copy(stats, f"{P}/{tag}_final_bench_default_kernel_stats.csv")
shutil.copy(f"{G}/m_bench.json", f"{P}/{tag}_final_bench_line.json")
shutil.copy(f"{G}/m_prof_bench.json", f"{P}/{tag}_final_bench_line_under_rocprof.json")
if os.path.exists(f"{G}/m_sq.txt"):
    hdr_sq = ("# SQ counters per kernel of the cfg3 train step (two rocprofv3 --pmc passes of 8 SQ counters + a kernel trace; tools/_measure.sh,\n"
              "# tools/sq_summary.py).  act / wait / stall = share of wave-cycles issuing / parked on s_waitcnt or a barrier / issue-stalled;\n"
              "# valu% / mfma% = pipe busy as a fraction of the launch; ldsconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n")
    open(f"{P}/{tag}_sq_counters.txt", "w").write(hdr_sq + open(f"{G}/m_sq.txt").read())
NSTEPS = 25
body = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), f"{G}/m_prof", str(NSTEPS)], text=True)
tot, n = 0.0, 0
for r in csv.DictReader(open(stats)):
    if "gemm_stream" in r["Name"] or "mlp_fused" in r["Name"]:
        tot += float(r["TotalDurationNs"])
        n += int(r["Calls"])
d = json.load(open(f"{G}/m_bench.json"))
fam = d["roofline"]["hbm_family"]
traffic = json.load(open(f"{P}/{tag}_pmc_linear_gemm_stream_traffic.json"))["avg_hbm_bytes_per_launch"]
hdr = f"""# rocprofv3 --kernel-trace --stats of the default bench command ({tag})

Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline` (25 train steps of cfg3,
bs 32, incl. 5 warm-up, `--steps 20`; the HIP-event probe brackets every {fam.get('sampled_every', 1)}th launch of the roofline family during the 20 timed steps).
Bench line of the same run: `{tag}_final_bench_line_under_rocprof.json`; un-profiled bench line: `{tag}_final_bench_line.json`
({d['ms_per_step']:.2f} ms/step = {d['value'] / 1e6:.1f} M HR px/s).

"""
tail = f"""

Template arguments: `gemm_stream_split_kernel<epilogue, K/64, rows per tile, epilogue waves>`, `gemm_stream_kernel<epilogue, K/64, rows per
tile, K split>`; `gemm_kernel<loader, epilogue, NT, narrow>` (loader 1 conv3x3 / 2 conv3x3 over pixel-shuffled input);
epilogue 0 bf16, 1 qkv, 2 proj+residual+LN2, 3 GELU, 4 residual+next LN, 5 dGELU, 6 LeakyReLU, 7 PixelShuffle, 8 image,
10 residual->bf16, 11 dLeakyReLU, 12 f32+bf16, 13 fused LayerNorm backward.

`mlp_fused_fwd_kernel` / `mlp_fused_bwd_kernel` (csrc/gemm_stream.hip) replace the fc1 + fc2 forward launches and the dGELU (`<5, ...>`) +
LayerNorm-backward (`<13, 6, ...>`) launches; `qkv_attn_bwd_kernel` (csrc/attn_bwd_fused.hip) replaces `attn_bwd_kernel` + the proj dgrad (`<0, 3, ...>`).

The roofline kernel family of bench.py is `gemm_stream*_kernel` + `mlp_fused_*_kernel` (csrc/gemm_stream.hip): {n / NSTEPS:.0f} launches/step, average duration in
this trace {tot / n / 1e3:.1f} us over {n} launches; bench.py's HIP-event probe in the un-profiled run: {fam['avg_launch_us']:.1f} us
(`roofline.avg_launch_us`, every {fam.get('sampled_every', 1)}th launch sampled; the event pair itself adds ~3 us to a bracketed launch).
PMC HBM traffic of the family: `{tag}_pmc_linear_gemm_stream_traffic.json` ({traffic / 1e6:.0f} MB/launch measured vs {fam['algorithmic_bytes_per_launch'] / 1e6:.0f} MB
algorithmic un-padded; the difference is the 180->192 channel and 30->32 head padding).
"""
open(f"{P}/{tag}_final_bench_default_kernel_stats.md", "w").write(hdr + body + tail)
r = fam
print(f"ms/step {d['ms_per_step']:.2f}  value {d['value'] / 1e6:.2f} M px/s  tflops {d['config']['step_tflops_per_gpu']:.1f}  roofline {r['achieved']:.0f} GB/s "
      f"frac {r['frac']:.3f}  probe {r['avg_launch_us']:.1f} us  rocprof {tot / n / 1e3:.1f} us  share {r['share_of_step']:.3f}  cpu {d['cpu_baseline']['value']:.0f}")
